"""GPU parity tests: the HIP path (libslode.so through the C ABI) against the CPU oracle on identical seeded inputs.
Run on an MI355X with ``pytest -m gpu``.  Tolerances (fp32 path, stated per test):
  * stage-time table: bit-exact;  * encoder loc/scale: 2e-5 relative;  * latent trajectories: 1e-5 * max(1, |x|);
  * -ELBO: 1e-5 relative (north-star bar: 1e-4);  * gradients: 5e-4 norm-wise relative per tensor vs the fp64 oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import slode_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    # name: (oracle spec factory, kwargs, B, T)
    "cvs_c1_rk4": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="rk4"), 48, 200),        # BASELINE config[1] shapes
    "cvs_c0_rk4": ("cvs", dict(z_iext=1, z_rtpr=1, z_eps=2, solver="rk4"), 32, 100),        # BASELINE config[0]
    "cvs_ref_midpoint": ("cvs", dict(solver="midpoint"), 37, 86),                           # reference default, ragged B
    "cvs_gauss_euler": ("cvs", dict(gauss=True, solver="euler"), 16, 86),
    "challenge_c4_rk4_gauss": ("challenge", dict(gauss=True, solver="rk4"), 12, 300),       # BASELINE config[4] shapes
    "challenge_ald_midpoint": ("challenge", dict(solver="midpoint"), 9, 142),
    "proc_c2_rk4": ("proc", dict(z_g=10, z_eps=10, solver="rk4"), 16, 100),                 # BASELINE config[2] shapes (fixed grid)
    "proc_gauss_midpoint": ("proc", dict(z_g=3, z_eps=2, gauss=True, solver="midpoint"), 7, 86),
    # a label head that reads 17 latent dims (a legal reference config, e.g. z_iext_dim = 17): every entry point takes it; the auxiliary
    # kernel switches to its wide instantiation (round-3 advisor finding: check_shape used to reject z_dim > 16 everywhere)
    "cvs_wide_head_z17": ("cvs", dict(z_iext=17, z_rtpr=3, z_eps=2, solver="midpoint"), 5, 64),
}


def _mk(case):
    from structured_latent_odes_amd import engine as E
    fam, kw, B, T = CASES[case]
    ospec = {"cvs": O.cvs_spec, "challenge": O.challenge_spec, "proc": O.proc_spec}[fam](**kw)
    espec = {"cvs": E.cvs_spec, "challenge": E.challenge_spec, "proc": E.proc_spec}[fam](**kw)
    S = 8 if fam == "proc" else 5
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(11)
    # move off the near-zero initialisation so every gradient path is exercised; keep std params positive-ish
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    dev = torch.device("cuda:0")
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)   # [B,C,T] view of contiguous [B,T,C]
    return dict(ospec=ospec, p=p, obs=obs, u=u, eps=eps, times=times, eng=eng, flat=flat, obs_d=obs_d,
                u_d=u.to(dev).contiguous(), eps_d=eps.to(dev).contiguous(), dev=dev, B=B, T=T, S=S)


@pytest.fixture(scope="module", params=list(CASES))
def ctx(request):
    return _mk(request.param)


def _close(a, b, tol=1e-5):
    """|a - b| <= tol * max(1, |b|) element-wise (trajectories are O(1); the perturbed test weights can push them higher)."""
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs() / b.abs().clamp_min(1.0)).max().item() < tol


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_library_loaded_from_tree():
    from structured_latent_odes_amd import _lib
    lib = _lib.load()
    assert lib.slode_version() >= 100
    assert os.path.dirname(_lib.LIB_PATH).endswith("structured_latent_odes_amd")


def test_stage_times_bit_exact(ctx):
    want = O.stage_times(ctx["times"], ctx["ospec"].solver)
    assert torch.equal(ctx["eng"]._stage_t.cpu(), want)


def test_encoder_forward(ctx):
    loc, scale, pooled, hid = ctx["eng"].encoder_fwd(ctx["flat"], ctx["obs_d"])
    wl, ws = O.encoder_conv(ctx["p"], ctx["obs"], ctx["ospec"].pool_size)
    assert _rel(loc, wl) < 2e-5 and _rel(scale, ws) < 2e-5
    # contiguous [B,C,T] input (the proc layout) gives the same result as the permuted view
    loc2, scale2, _, _ = ctx["eng"].encoder_fwd(ctx["flat"], ctx["obs_d"].contiguous())
    assert torch.equal(loc, loc2) and torch.equal(scale, scale2)


def test_ode_solve_forward(ctx):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(ctx["B"], ctx["ospec"].latent_dim, generator=g)
    x = ctx["eng"].ode_solve(ctx["flat"], z.to(ctx["dev"]))
    want = O.solve_ode(ctx["p"], z, ctx["times"], ctx["ospec"].solver)
    assert _close(x, want)
    want64 = O.solve_ode({k: v.double() for k, v in ctx["p"].items()}, z.double(), ctx["times"].double(), ctx["ospec"].solver)
    assert _close(x, want64)


def test_decode_heads(ctx):
    g = torch.Generator().manual_seed(4)
    x = torch.rand(ctx["B"], ctx["T"], ctx["S"], generator=g)
    mu, std = ctx["eng"].decode_heads(ctx["flat"], x.to(ctx["dev"]))
    import torch.nn.functional as F
    names = ["decoder.output_mean.0.weight"] if ctx["ospec"].gauss else ["decoder.output_%s.0.weight" % q for q in ("q50", "q75", "q25")]
    for i, n in enumerate(names):
        assert _rel(mu[i], F.linear(x, ctx["p"][n]).permute(0, 2, 1)) < 1e-6
    assert _rel(std, F.softplus(ctx["p"]["decoder.constant_std"])) < 1e-6


def test_decode_heads_backward(ctx):
    """slode_decode_heads_bwd (autograd through the materialising Decoder.forward, models/decoders.py:42-54) against a plain fp32 torch
    reference of the same three contractions: g_x = sum_{q,c} g_mu W_q, g_W_q = sum_{b,t} g_mu x, g_cstd = g_std * sigmoid(constant_std).
    Tolerance 2e-5 norm-wise (fp32 sums over B*T terms in a different order)."""
    eng, dev, B, T, S = ctx["eng"], ctx["dev"], ctx["B"], ctx["T"], ctx["S"]
    sp = ctx["ospec"]
    names = ["output_mean"] if sp.gauss else ["output_q50", "output_q75", "output_q25"]
    Q, C = len(names), ctx["obs"].shape[1]
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, T, S, generator=g)
    g_mu = torch.randn(Q, B, C, T, generator=g)
    g_std = torch.randn(C, T, generator=g)
    g_x, g_heads, g_c = eng.decode_heads_bwd(ctx["flat"], x.to(dev), g_mu.to(dev), g_std.to(dev))
    W = [ctx["p"]["decoder.%s.0.weight" % n] for n in names]
    want_x = sum(torch.einsum("bct,cs->bts", g_mu[q], W[q]) for q in range(Q))
    assert _rel(g_x, want_x) < 2e-5
    for q in range(Q):
        assert _rel(g_heads[q], torch.einsum("bct,bts->cs", g_mu[q], x)) < 2e-5
    assert _rel(g_c, g_std * torch.sigmoid(ctx["p"]["decoder.constant_std"])) < 2e-6
    g_x2, g_heads2, _ = eng.decode_heads_bwd(ctx["flat"], x.to(dev), g_mu.to(dev))          # no std gradient; bitwise repeatable
    assert torch.equal(g_x, g_x2) and torch.equal(g_heads, g_heads2)


def test_eval_side_small_nets(ctx):
    """slode_initialize_state / slode_prior_nets / slode_label_heads (the eval-side entry points: OdeModel.initialize_state, the prior
    nets of recon(is_post=False), classifier / pred_inputs) against the oracle's restatement of the same modules: 1e-6 relative."""
    eng, dev, B, ospec, p = ctx["eng"], ctx["dev"], ctx["B"], ctx["ospec"], ctx["p"]
    g = torch.Generator().manual_seed(6)
    z = torch.randn(B, ospec.latent_dim, generator=g)
    assert _rel(eng.initialize_state(ctx["flat"], z.to(dev)), O.initialize_state(p, z)) < 1e-6
    loc, scale = eng.prior_nets(ctx["flat"], ctx["u_d"])
    wl, ws = O.prior_loc_scale(p, ospec, ctx["u"])
    assert _rel(loc, wl) < 1e-6 and _rel(scale, ws) < 1e-6
    probs = eng.label_heads(ctx["flat"], z.to(dev)).cpu()
    for kind, prefix, zo, zd, uo, ud in ospec.aux_heads:
        zg = z[:, zo:zo + zd]
        want = {"bernoulli": O.classifier_sigmoid, "onehot": O.classifier_softmax}.get(kind)
        want = want(p, prefix, zg) if want is not None else O.regressor_exp_exp(p, prefix, zg)[0]
        assert _rel(probs[:, uo:uo + ud], want) < 2e-6, prefix


def test_elbo_loss_and_trajectories(ctx):
    eng, dev = ctx["eng"], ctx["dev"]
    loss = torch.zeros(1, device=dev)
    x = torch.empty(ctx["B"], ctx["T"], ctx["S"], device=dev)
    z = torch.empty(ctx["B"], ctx["ospec"].latent_dim, device=dev)
    eng.elbo_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss, grads=None, x_out=x, z_out=z)
    with torch.no_grad():
        want, parts = O.main_loss(ctx["p"], ctx["ospec"], ctx["obs"], ctx["u"], ctx["eps"], ctx["times"], return_parts=True)
    assert _rel(z, parts["z"]) < 2e-5
    assert _close(x, parts["dec"][0])
    assert abs(loss.item() - want.item()) / abs(want.item()) < 1e-5, (loss.item(), want.item())


def test_elbo_gradients(ctx):
    eng, dev = ctx["eng"], ctx["dev"]
    loss = torch.zeros(1, device=dev)
    grads = torch.full((eng.n_params,), float("nan"), device=dev)
    eng.elbo_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss, grads=grads)
    p64 = {k: v.double() for k, v in ctx["p"].items()}
    want_loss, want = O.loss_and_grads(p64, ctx["ospec"], ctx["obs"].double(), ctx["u"].double(), ctx["eps"].double(), ctx["times"].double())
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 1e-5
    got = eng.unpack(grads)
    assert torch.isfinite(grads).all()
    worst = {}
    for k, v in got.items():
        worst[k] = _rel(v, want[k])
    bad = {k: e for k, e in worst.items() if e > 5e-4}
    assert not bad, bad
    # forward-only evaluation gives the same loss value (SVI.evaluate_loss vs SVI.step, training_cvs.py:81,152)
    loss2 = torch.zeros(1, device=dev)
    eng.elbo_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss2, grads=None)
    # (to fp32 summation order: the forward-only step runs the encoder in its own launch, enc_fwd2, the training step of the metric
    #  shape inside the ODE kernel with 16-byte loads -- the same products, summed in a different order)
    assert abs(loss2.item() - loss.item()) <= 2e-6 * abs(loss.item())


def test_bitwise_reproducible(ctx):
    eng, dev = ctx["eng"], ctx["dev"]
    outs = []
    for _ in range(2):
        loss = torch.zeros(1, device=dev)
        grads = torch.zeros(eng.n_params, device=dev)
        eng.elbo_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss, grads=grads)
        outs.append((loss.clone(), grads.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_encoder_backward_standalone(ctx):
    eng, dev, B = ctx["eng"], ctx["dev"], ctx["B"]
    L = ctx["ospec"].latent_dim
    g = torch.Generator().manual_seed(8)
    g_loc, g_scale = torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)
    loc, scale, pooled, hid = eng.encoder_fwd(ctx["flat"], ctx["obs_d"])
    grads = torch.zeros(eng.n_params, device=dev)
    eng.encoder_bwd(ctx["flat"], ctx["obs_d"], scale, pooled, hid, g_loc.to(dev), g_scale.to(dev), grads)
    q = {k: v.double().requires_grad_(k.startswith("encoder.")) for k, v in ctx["p"].items()}
    wl, ws = O.encoder_conv(q, ctx["obs"].double(), ctx["ospec"].pool_size)
    ((wl * g_loc.double()).sum() + (ws * g_scale.double()).sum()).backward()
    got = eng.unpack(grads)
    for k in q:
        if k.startswith("encoder."):
            assert _rel(got[k], q[k].grad) < 2e-4, k


def test_ode_solve_backward_standalone(ctx):
    eng, dev, B, T = ctx["eng"], ctx["dev"], ctx["B"], ctx["T"]
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, ctx["ospec"].latent_dim, generator=g)
    gx = torch.randn(B, T, ctx["S"], generator=g)
    grads = torch.zeros(eng.n_params, device=dev)
    gz = eng.ode_solve_bwd(ctx["flat"], z.to(dev), gx.to(dev), grads)
    q = {k: v.double().requires_grad_("ode_model" in k) for k, v in ctx["p"].items()}
    zz = z.double().requires_grad_(True)
    (O.solve_ode(q, zz, ctx["times"].double(), ctx["ospec"].solver) * gx.double()).sum().backward()
    assert _rel(gz, zz.grad) < 5e-4
    got = eng.unpack(grads)
    for k in q:
        if "ode_model" in k:
            assert _rel(got[k], q[k].grad) < 5e-4, k


def test_golden_reference_vectors_on_gpu(golden_dir):
    """The reference's own outputs (tests/golden, generated from the reference modules) through the HIP path."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    g1 = np.load(os.path.join(golden_dir, "g1_encoder_conv.npz"))
    for case in range(5):
        B, C, T, L = [int(v) for v in g1["c%d.meta" % case]]
        spec = E.ModelSpec("golden", False, C, L, L, 0, [])
        eng = E.Engine(spec, T, dev)
        p = {"encoder." + k[len("c%d.p." % case):]: torch.from_numpy(g1[k]) for k in g1.files if k.startswith("c%d.p." % case)}
        flat = torch.zeros(eng.n_params, device=dev)
        for key, off, shp in eng.param_table():
            if key in p:
                flat[off:off + p[key].numel()] = p[key].reshape(-1).to(dev)
        x = torch.from_numpy(g1["c%d.x" % case]).to(dev)
        loc, scale, _, _ = eng.encoder_fwd(flat, x, save=False)
        assert _rel(loc, torch.from_numpy(g1["c%d.loc" % case])) < 2e-5
        assert _rel(scale, torch.from_numpy(g1["c%d.scale" % case])) < 2e-5
    g3 = np.load(os.path.join(golden_dir, "g3_dynamics.npz"))
    for case, (L, S) in enumerate([(4, 5), (8, 5), (15, 5), (50, 8)]):
        spec = E.ModelSpec("golden", False, 3, L, L, 0, [], ode_state_dim=S, solver="euler")
        T = 100
        eng = E.Engine(spec, T, dev)
        eng.set_times(torch.arange(T, dtype=torch.float32) * 0.01)
        p = {k[len("d%d.p." % case):]: torch.from_numpy(g3[k]) for k in g3.files if k.startswith("d%d.p." % case)}
        flat = torch.zeros(eng.n_params, device=dev)
        for key, off, shp in eng.param_table():
            if key in p:
                flat[off:off + p[key].numel()] = p[key].reshape(-1).to(dev)
        z = torch.from_numpy(g3["d%d.z" % case]).to(dev)
        x = eng.ode_solve(flat, z)
        # x[:,0] is OdeModel.initialize_state(z) (blackbox_ode.py:32-34); the first euler step is x0 + dt*f(0, x0)
        x0 = torch.from_numpy(g3["d%d.x0" % case])
        assert (x[:, 0].cpu() - x0).abs().max().item() < 2e-6
        f0 = O.dynamics(p, torch.tensor(0.0), x0, torch.from_numpy(g3["d%d.z" % case]))
        assert (x[:, 1].cpu() - (x0 + 0.01 * f0)).abs().max().item() < 2e-6


def test_full_size_batch_linearity():
    """BASELINE config[1] at full size (B=1024, T=200): the loss and gradient are sums over trajectories, so the
    step over the whole batch equals the sum over its two halves (size-independent property; also the data-parallel
    contract of SURVEY 8e)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    ospec = O.cvs_spec(3, 3, 2, solver="rk4")
    B, T = 1024, 200
    p = O.init_params(ospec, T=T)
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(E.cvs_spec(3, 3, 2, solver="rk4"), T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    u_d, eps_d = u.to(dev), eps.to(dev)

    def run(sl):
        loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
        eng.elbo_step(flat, obs_d[sl], u_d[sl].contiguous(), eps_d[sl].contiguous(), loss, grads)
        return loss.double().cpu(), grads.double().cpu()
    l_all, g_all = run(slice(0, B))
    l_a, g_a = run(slice(0, B // 2))
    l_b, g_b = run(slice(B // 2, B))
    assert abs((l_a + l_b - l_all).item()) / abs(l_all.item()) < 1e-6
    assert ((g_a + g_b - g_all).norm() / g_all.norm()).item() < 1e-5
    # spot-check 32 trajectories of the full-size batch against the oracle
    with torch.no_grad():
        want = O.main_loss(p, ospec, obs[:32], u[:32], eps[:32], times)
    l32, _ = run(slice(0, 32))
    assert abs(l32.item() - want.item()) / abs(want.item()) < 1e-5


@pytest.mark.parametrize("fam", ["cvs", "proc"])
def test_dopri5_forward_solution_level(fam):
    """Adaptive Dormand-Prince solve (BASELINE config[2] solver; per-trajectory controller).  The reference's torchdiffeq
    controller is batch-coupled and unpinned, so parity is at SOLUTION level: against the oracle's per-trajectory
    restatement run in fp64 at the same tolerances, and against a tight-tolerance fp64 solve (scipy-validated oracle)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    if fam == "proc":
        ospec, espec, S, T = O.proc_spec(solver="dopri5"), E.proc_spec(solver="dopri5"), 8, 100
    else:
        ospec, espec, S, T = O.cvs_spec(3, 3, 2, solver="dopri5"), E.cvs_spec(3, 3, 2, solver="dopri5"), 5, 60
    espec.rtol, espec.atol = 1e-6, 1e-8
    B = 38                                         # two 16-trajectory workgroups and a ragged third
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(21)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    _, _, _, times = O.synthetic_batch(ospec, 4, T)
    if fam == "cvs":
        times = times * 0.25
    z = torch.randn(B, ospec.latent_dim, generator=g)
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    x = eng.ode_solve(eng.pack(p), z.to(dev))
    assert torch.isfinite(x).all()
    p64 = {k: v.double() for k, v in p.items()}
    tight = O.solve_ode(p64, z.double(), times.double(), "dopri5", rtol=1e-10, atol=1e-12, per_trajectory=True)
    ref32 = O.solve_ode(p, z, times, "dopri5", rtol=1e-6, atol=1e-8, per_trajectory=True)     # the same algorithm in fp32 on the CPU
    scale = tight.abs().clamp_min(1.0)
    err_gpu = ((x.cpu().double() - tight).abs() / scale).max().item()
    err_ref = ((ref32.double() - tight).abs() / scale).max().item()
    # adaptive step sequences differ between implementations, so the bar is the solver's own accuracy: the HIP solve must be
    # as close to the true solution as the fp32 CPU restatement at the same tolerances (x3 slack), and within 1e-3 absolutely
    assert err_gpu < 3.0 * err_ref + 1e-5, (err_gpu, err_ref)
    assert err_gpu < 1e-3
    assert torch.equal(x[:, 0].cpu(), O.initialize_state(p, z)) or ((x[:, 0].cpu() - O.initialize_state(p, z)).abs().max() < 2e-6)
    # gradients through the adaptive solver come with the ELBO step (test_dopri5_elbo_step_solution_level); the stand-alone
    # solve-backward entry point is fixed-grid only and says so loudly
    from structured_latent_odes_amd._lib import SlodeError
    with pytest.raises(SlodeError):
        eng.ode_solve_bwd(eng.pack(p), z.to(dev), torch.zeros(B, T, S, device=dev), torch.zeros(eng.n_params, device=dev))


def test_dopri5_rejects_grids_it_cannot_walk():
    """The adaptive kernels integrate forward in time only.  torchdiffeq would take a decreasing grid (it integrates in -t); here
    Engine.set_times raises for solver='dopri5', and a caller that reaches the C ABI with such a table anyway (or with a repeated
    time) gets NaN trajectories and a NaN loss with AND without gradients -- never a finite number from an extrapolated dense output."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    ospec, espec, T, B = O.cvs_spec(3, 3, 2, solver="dopri5"), E.cvs_spec(3, 3, 2, solver="dopri5"), 60, 20
    p = O.init_params(ospec, T=T)
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(espec, T, dev)
    with pytest.raises(ValueError, match="strictly increasing"):
        eng.set_times(times.flip(0))
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
    eng.set_times(times)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, None)
    assert torch.isfinite(loss).all()
    repeated = times.clone()
    repeated[30] = repeated[29]
    for bad in (times.flip(0).contiguous(), repeated):
        eng._times = bad.to(dev)                               # what a bare C-ABI caller could pass
        x = eng.ode_solve(flat, torch.zeros(B, ospec.latent_dim, device=dev))
        assert torch.isnan(x[:, 1:]).all()
        eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, None)          # SVI.evaluate_loss: no backward kernel runs
        assert torch.isnan(loss).all()
        eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads)
        assert torch.isnan(loss).all()


@pytest.mark.parametrize("fam,mode", [("cvs", "exact"), ("proc", "exact"), ("cvs", "reference_adjoint"), ("proc_c2", "exact"), ("cvs_odd", "exact")])
def test_dopri5_elbo_step_solution_level(fam, mode):
    """ELBO step with the adaptive solver (BASELINE config[2]): forward solve with recorded steps, reverse mode over the records.
    Parity is at solution level (see test_dopri5_forward_solution_level): -ELBO and every gradient against the fp64 oracle run at tight
    tolerances -- `exact`: autograd through the oracle's per-trajectory dopri5; `reference_adjoint`: the same with the latent detached
    inside the dynamics (oracle solve_ode).  Tolerances: see the bars below (each = a fixed-grid-sized term + 3x the oracle's own sensitivity to
    the adaptive step sequence)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    if fam == "proc_c2":                       # BASELINE config[2] as written: proc, latent dim 50 (4 x 10 + 10), T = 100, dopri5
        fam, kw = "proc", dict(z_g=10, z_eps=10)
    else:
        kw = dict(z_g=3, z_eps=2) if fam == "proc" else dict(z_iext=3, z_rtpr=3, z_eps=2)
    odd = fam == "cvs_odd"                     # T * S odd: the ragged last workgroup's dL/dx block is not a multiple of 16 bytes
    fam = "cvs" if odd else fam
    # 16 trajectories per workgroup of the solver kernels: two full workgroups and a ragged third (the fp64 oracle differentiates every
    # trajectory's own adaptive step sequence in eager mode twice -- its cost, linear in B, is what bounds B here)
    S, T, B = (8, 100, 38) if fam == "proc" else (5, 61 if odd else 60, 22 if odd else 38)
    mk_o, mk_e = (O.proc_spec, E.proc_spec) if fam == "proc" else (O.cvs_spec, E.cvs_spec)
    ospec = mk_o(solver="dopri5", **kw)
    ospec.solver_kw = dict(rtol=1e-8, atol=1e-10, per_trajectory=True)
    espec = mk_e(solver="dopri5", **kw)
    espec.rtol, espec.atol, espec.grad_mode = 1e-6, 1e-8, mode
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(31)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    if fam == "cvs":
        times = times * 0.25
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    loss = torch.zeros(1, device=dev)
    grads = torch.full((eng.n_params,), float("nan"), device=dev)
    x = torch.empty(B, T, S, device=dev)
    eng.workspace(B).fill_(float("nan"))          # nothing may survive from an earlier launch: every slab element has an owner
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads, x_out=x)
    assert torch.isfinite(loss).all() and torch.isfinite(grads).all()
    steps = eng.dopri5_step_counts(B)
    assert steps.shape == (B,) and int(steps.min()) >= 1 and int(steps.max()) < 2048
    p64 = {k: v.double() for k, v in p.items()}
    ospec.grad_mode = mode
    q = {k: v.clone().requires_grad_(True) for k, v in p64.items()}
    want_loss, parts = O.main_loss(q, ospec, obs.double(), u.double(), eps.double(), times.double(), return_parts=True)
    want_loss.backward()
    want = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in q.items()}
    want_loss, tight = want_loss.detach(), parts["dec"][0].detach()
    got = eng.unpack(grads)
    # Bar per tensor: 5e-4 (the fixed-grid bar) + 3x the oracle's own sensitivity to the step sequence -- its gradient at the engine's
    # tolerances (fp64) against the tight one.  The solver controls the error of the SOLUTION; gradients that integrate relu'(.) of
    # the hidden layer over time (dynamics_hidden, and through z the encoder) see an O(step) quadrature error at every kink and differ
    # by ~1e-3 between any two adaptive step sequences, the rest by ~1e-5.
    ospec.solver_kw = dict(rtol=1e-6, atol=1e-8, per_trajectory=True)
    loose_loss, loose = O.loss_and_grads(p64, ospec, obs.double(), u.double(), eps.double(), times.double())
    # -ELBO: the likelihood scale is 0.01, so the solver's own tolerance shows in the loss amplified a hundredfold -- the oracle at the
    # engine's tolerances sits 1.6e-5 (relative) from the oracle at tight ones in the cvs case.  Bar: 2e-5 + 3x that sensitivity.
    # Round 4 measured where the engine sits inside that band and why (tools/dp5_accuracy.py, profiles/r04_d_dp5_accuracy_*.txt): at
    # rtol 1e-6 the shipped kernel lands 1.6e-5 from the tight loss -- exactly where the fp64 oracle at the same tolerances lands -- and a
    # diagnostic build with IEEE division / sqrt / pow in the step-size controller and the dense output evaluated in fp64 lands at 2.9e-5:
    # MORE precise arithmetic moves the loss FURTHER away.  What moves it is the accepted-step sequence (an accept / reject decision that
    # flips at the tolerance boundary), not the rounding of v_rcp / v_sqrt / v_exp or of the fp32 Horner form; both solutions are 3e-4
    # from the tight trajectories, as the oracle's is (3.6e-4).  At torchdiffeq's DEFAULT tolerances the question does not arise:
    # test_dopri5_elbo_step_at_default_tolerances holds a plain 2e-5 bar (observed 2e-7).
    loss_sens = abs(loose_loss.item() - want_loss.item()) / abs(want_loss.item())
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 2e-5 + 3.0 * loss_sens, (loss.item(), want_loss.item(), loose_loss.item())
    bad = {k: (_rel(v, want[k]), _rel(loose[k], want[k])) for k, v in got.items() if _rel(v, want[k]) > 5e-4 + 3.0 * _rel(loose[k], want[k])}
    assert not bad, bad
    assert all(_rel(v, want[k]) < 1e-2 for k, v in got.items())
    # the trajectories handed back are the adaptive solver's
    err_x = ((x.cpu().double() - tight).abs() / tight.abs().clamp_min(1.0)).max().item()
    assert err_x < 1e-3, err_x                                    # same absolute bar as test_dopri5_forward_solution_level
    # loss-only evaluation (SVI.evaluate_loss) scores the same solution
    loss2 = torch.zeros(1, device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss2, grads=None)
    # (to fp32 summation order: the forward-only step runs the encoder in its own launch, enc_fwd2, the training step of the metric
    #  shape inside the ODE kernel with 16-byte loads -- the same products, summed in a different order)
    assert abs(loss2.item() - loss.item()) <= 2e-6 * abs(loss.item())
    # bitwise reproducible
    grads2 = torch.zeros_like(grads)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss2, grads=grads2)
    assert torch.equal(grads, grads2)


def test_dopri5_elbo_step_at_default_tolerances():
    """The cvs family at torchdiffeq's default tolerances (rtol 1e-7, atol 1e-9: what `solver="dopri5"` means in the reference,
    models/blackbox_ode.py:41-45): -ELBO against the fp64 oracle at tight tolerances within a PLAIN 2e-5 -- no sensitivity term -- and the
    trajectories within 1e-4 (observed: 2e-7 and 7e-5; tools/dp5_accuracy.py)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    S, T, B = 5, 60, 38
    kw = dict(z_iext=3, z_rtpr=3, z_eps=2)
    ospec, espec = O.cvs_spec(solver="dopri5", **kw), E.cvs_spec(solver="dopri5", **kw)      # espec.rtol / atol: the defaults 1e-7 / 1e-9
    ospec.solver_kw = dict(rtol=1e-10, atol=1e-12, per_trajectory=True)
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(31)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    times = times * 0.25
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    loss, grads, x = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev), torch.empty(B, T, S, device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads, x_out=x)
    p64 = {k: v.double() for k, v in p.items()}
    with torch.no_grad():
        want_loss, parts = O.main_loss(p64, ospec, obs.double(), u.double(), eps.double(), times.double(), return_parts=True)
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 2e-5, (loss.item(), want_loss.item())
    assert ((x.cpu().double() - parts["dec"][0]).abs() / parts["dec"][0].abs().clamp_min(1.0)).max().item() < 1e-4
    assert torch.isfinite(grads).all()


@pytest.mark.parametrize("lpt", ["16", "32", "64", "0"])   # "0": chosen by batch size (the default: sixteen up to 4 x SIMDs trajectories)
@pytest.mark.parametrize("fam", ["cvs", "proc"])
def test_dopri5_wider_lane_groups_equal_the_eight_lane_kernel_bitwise(fam, lpt, monkeypatch):
    """Round 4: the forward adaptive solve with 16 / 32 / 64 lanes per trajectory (lane = stage evaluation x state component: the stage times
    of a step side by side, dopri5_lpt_kernel) instead of eight.  Same operations in the same order: trajectories, step counts, -ELBO and
    every gradient element are bit for bit those of the eight-lane kernel (SLODE_DP5_LPT=8), in the bare solve and in the training step;
    ragged batch (B not a multiple of the trajectories per workgroup)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    S, T, B = (8, 100, 37) if fam == "proc" else (5, 60, 38)
    kw = dict(z_g=10, z_eps=10) if fam == "proc" else dict(z_iext=3, z_rtpr=3, z_eps=2)
    mk_o, mk_e = (O.proc_spec, E.proc_spec) if fam == "proc" else (O.cvs_spec, E.cvs_spec)
    ospec = mk_o(solver="dopri5", **kw)
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(31)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    if fam == "cvs":
        times = times * 0.25
    outs = []
    for w64 in ("8", lpt):
        monkeypatch.setenv("SLODE_DP5_LPT", w64)
        eng = E.Engine(mk_e(solver="dopri5", **kw), T, dev)          # the switch is read in slode_create
        eng.set_times(times)
        flat = eng.pack(p)
        obs_d = obs.contiguous().to(dev) if fam == "proc" else obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
        loss, grads = torch.zeros(1, device=dev), torch.full((eng.n_params,), float("nan"), device=dev)
        x = torch.empty(B, T, S, device=dev)
        eng.workspace(B).fill_(float("nan"))
        eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads, x_out=x)
        z = torch.randn(B, ospec.latent_dim, generator=torch.Generator().manual_seed(1)).to(dev)
        outs.append((loss.clone(), grads.clone(), x.clone(), eng.dopri5_step_counts(B).clone(), eng.ode_solve(flat, z).clone()))
    a, b = outs
    assert torch.isfinite(a[0]).all() and torch.isfinite(b[1]).all()
    assert torch.equal(a[3], b[3]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
    assert a[0].item() == b[0].item() and torch.equal(a[1], b[1])


def test_dopri5_config2_full_size_properties():
    """BASELINE config[2] at full size (proc, B = 4096, T = 100, latent dim 50, dopri5 at torchdiffeq's default tolerances): the
    size-independent properties of the step -- finite, bitwise reproducible, and additive over trajectories (per-trajectory step-size
    control: the whole batch equals the sum of its halves up to fp32 summation order)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    ospec = O.proc_spec(z_g=10, z_eps=10, solver="dopri5")
    B, T = 4096, 100
    p = O.init_params(ospec, T=T, S=8)
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(E.proc_spec(z_g=10, z_eps=10, solver="dopri5"), T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d, u_d, eps_d = obs.contiguous().to(dev), u.to(dev), eps.to(dev)

    def run(sl):
        loss, grads = torch.zeros(1, device=dev), torch.full((eng.n_params,), float("nan"), device=dev)
        eng.elbo_step(flat, obs_d[sl].contiguous(), u_d[sl].contiguous(), eps_d[sl].contiguous(), loss, grads)
        return loss.clone(), grads.clone()
    l_all, g_all = run(slice(0, B))
    assert torch.isfinite(l_all).all() and torch.isfinite(g_all).all()
    l_again, g_again = run(slice(0, B))
    assert torch.equal(l_all, l_again) and torch.equal(g_all, g_again)
    l_a, g_a = run(slice(0, B // 2))
    l_b, g_b = run(slice(B // 2, B))
    assert abs((l_a + l_b - l_all).item()) / abs(l_all.item()) < 1e-6
    assert ((g_a + g_b - g_all).double().norm() / g_all.double().norm()).item() < 1e-5
    # 16 of its trajectories against the fp64 oracle (tight tolerances): the loss
    sl = slice(100, 116)
    ospec.solver_kw = dict(rtol=1e-8, atol=1e-10, per_trajectory=True)
    with torch.no_grad():
        want = O.main_loss({k: v.double() for k, v in p.items()}, ospec, obs[sl].double(), u[sl].double(), eps[sl].double(), times.double())
    got, _ = run(sl)
    assert abs(got.item() - want.item()) / abs(want.item()) < 2e-5


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
def test_config4_full_shard_properties(mode):
    """BASELINE config[4] as written: mechanistic_challenge_Gauss, T=300, batch 2048 over 4 GPUs = a shard of B=512 per GPU, latent 15
    (5,5,5), rk4.  The fp64 oracle cannot finish 512 x 300 in seconds, so the full shard is checked through size-independent
    properties -- finite, bitwise repeatable, additive over halves (the loss is a plain sum over trajectories, training_challenge.py
    divides afterwards) -- and 16 of its trajectories against the fp64 oracle: -ELBO 1e-5 relative, every gradient tensor 5e-4 norm-wise
    (the bars of the small cases).  reference_adjoint = the config's own adjoint_solver=True gradients."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    B, T = 512, 300
    import dataclasses
    ospec = dataclasses.replace(O.challenge_spec(gauss=True, solver="rk4"), grad_mode=mode)
    espec = dataclasses.replace(E.challenge_spec(gauss=True, solver="rk4"), grad_mode=mode)
    p = O.init_params(ospec, T=T)
    g = torch.Generator().manual_seed(5)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)       # the challenge batches' native layout (training_challenge.py:31)
    u_d, eps_d = u.to(dev), eps.to(dev)

    def run(sl):
        loss, grads = torch.zeros(1, device=dev), torch.full((eng.n_params,), float("nan"), device=dev)
        eng.elbo_step(flat, obs_d[sl], u_d[sl].contiguous(), eps_d[sl].contiguous(), loss, grads)
        return loss.clone(), grads.clone()
    l_all, g_all = run(slice(0, B))
    assert torch.isfinite(l_all).all() and torch.isfinite(g_all).all()
    l_again, g_again = run(slice(0, B))
    assert torch.equal(l_all, l_again) and torch.equal(g_all, g_again)
    l_a, g_a = run(slice(0, B // 2))
    l_b, g_b = run(slice(B // 2, B))
    assert abs((l_a + l_b - l_all).item()) / abs(l_all.item()) < 1e-6
    assert ((g_a + g_b - g_all).double().norm() / g_all.double().norm()).item() < 1e-5
    sl = slice(200, 216)
    p64 = {k: v.double() for k, v in p.items()}
    want_loss, want = O.loss_and_grads(p64, ospec, obs[sl].double(), u[sl].double(), eps[sl].double(), times.double())
    got_loss, got = run(sl)
    assert abs(got_loss.item() - want_loss.item()) / abs(want_loss.item()) < 1e-5
    for k, v in eng.unpack(got).items():
        w = want[k].double()
        if w.norm() == 0:
            assert v.abs().max().item() == 0, k
            continue
        assert _rel(v, w) < 5e-4, (k, _rel(v, w))


@pytest.mark.parametrize("layout", ["c_major", "strided"])
def test_elbo_other_observation_layouts(layout):
    """slode_elbo_step takes the folded-encoder path for dense [B,T,C] / [B,C,T] rows and the layer-by-layer kernels for
    any other strides; all must agree with the oracle (and with each other)."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    ospec = O.cvs_spec(3, 3, 2, solver="rk4")
    B, T = 20, 100
    p = O.init_params(ospec, T=T)
    g = torch.Generator().manual_seed(2)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(E.cvs_spec(3, 3, 2, solver="rk4"), T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    if layout == "c_major":
        obs_d = obs.contiguous().to(dev)                              # [B,C,T] contiguous (the proc data layout)
        assert obs_d.stride() == (3 * T, T, 1)
    else:
        big = torch.zeros(B, T + 7, 5, device=dev)                    # rows are NOT dense: padded in both trailing dims
        big[:, :T, :3] = obs.permute(0, 2, 1).to(dev)
        obs_d = big[:, :T, :3].permute(0, 2, 1)
        assert obs_d.stride() == ((T + 7) * 5, 1, 5)
    loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads)
    p64 = {k: v.double() for k, v in p.items()}
    want_loss, want = O.loss_and_grads(p64, ospec, obs.double(), u.double(), eps.double(), times.double())
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 1e-5
    got = eng.unpack(grads)
    bad = {k: _rel(v, want[k]) for k, v in got.items() if _rel(v, want[k]) > 5e-4}
    assert not bad, bad


@pytest.mark.parametrize("policy", ["one_workgroup_per_trajectory", "persistent_loop"])
def test_more_trajectories_than_workgroups(policy, monkeypatch):
    """B > CUs x occupancy.  Default policy (B <= 65,536): one workgroup per trajectory, loop-free kernel, the hardware queues the
    workgroups.  Persistent-loop policy (B > 65,536; forced here with SLODE_ODE_LOOP): a resident grid where every workgroup
    integrates several trajectories, carrying the register accumulators and the input prefetch across them.  Either way loss/gradient
    must equal the sum over chunks that each fit one pass, and a slice must match the oracle."""
    from structured_latent_odes_amd import engine as E
    if policy == "persistent_loop":
        monkeypatch.setenv("SLODE_ODE_LOOP", "1")
    dev = torch.device("cuda:0")
    ospec = O.cvs_spec(3, 3, 2, solver="rk4")
    B, T = 2500, 86                      # 2500 trajectories over <= 1024..2048 workgroups, ragged
    p = O.init_params(ospec, T=T)
    g = torch.Generator().manual_seed(4)
    p = {k: v + 0.03 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(E.cvs_spec(3, 3, 2, solver="rk4"), T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.to(dev)
    u_d, eps_d = u.to(dev), eps.to(dev)

    def run(sl):
        loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
        eng.elbo_step(flat, obs_d[sl], u_d[sl].contiguous(), eps_d[sl].contiguous(), loss, grads)
        return loss.double().cpu(), grads.double().cpu()
    l_all, g_all = run(slice(0, B))
    parts = [run(slice(i, min(B, i + 500))) for i in range(0, B, 500)]
    l_sum, g_sum = sum(x[0] for x in parts), sum(x[1] for x in parts)
    assert abs((l_sum - l_all).item()) / abs(l_all.item()) < 1e-6
    assert ((g_sum - g_all).norm() / g_all.norm()).item() < 1e-5
    sl = slice(2400, 2432)               # trajectories handled late in the persistent loops
    with torch.no_grad():
        want = O.main_loss(p, ospec, obs[sl], u[sl], eps[sl], times)
    assert abs(run(sl)[0].item() - want.item()) / abs(want.item()) < 1e-5
    x = torch.empty(B, T, 5, device=dev)
    loss = torch.zeros(1, device=dev)
    eng.elbo_step(flat, obs_d, u_d, eps_d, loss, None, x_out=x)
    with torch.no_grad():
        _, pr = O.main_loss(p, ospec, obs[sl], u[sl], eps[sl], times, return_parts=True)
    assert _close(x[sl], pr["dec"][0])


def test_aux_step_matches_oracle(ctx):
    """slode_aux_step (the second SVI object: model_meta / guide_meta) vs the oracle's aux_loss and its autograd gradient."""
    eng, dev = ctx["eng"], ctx["dev"]
    loss = torch.zeros(1, device=dev)
    grads = torch.full((eng.n_params,), float("nan"), device=dev)
    eng.aux_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss, grads)
    q = {k: v.double().requires_grad_(True) for k, v in ctx["p"].items()}
    want = O.aux_loss(q, ctx["ospec"], ctx["obs"].double(), ctx["u"].double(), ctx["eps"].double())
    want.backward()
    assert abs(loss.item() - want.item()) / abs(want.item()) < 1e-5, (loss.item(), want.item())
    assert torch.isfinite(grads).all()
    got = eng.unpack(grads)
    for k, v in got.items():
        w = q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])
        if float(w.abs().max()) == 0.0:
            assert float(v.abs().max()) == 0.0, k
        else:
            assert _rel(v, w) < 5e-4, (k, _rel(v, w))
    loss2 = torch.zeros(1, device=dev)
    eng.aux_step(ctx["flat"], ctx["obs_d"], ctx["u_d"], ctx["eps_d"], loss2, None)      # evaluate_loss
    # (to fp32 summation order: the forward-only step runs the encoder in its own launch, enc_fwd2, the training step of the metric
    #  shape inside the ODE kernel with 16-byte loads -- the same products, summed in a different order)
    assert abs(loss2.item() - loss.item()) <= 2e-6 * abs(loss.item())


def test_abi_error_paths():
    """Status-code error convention of the C ABI: bad arguments -> SLODE_EINVAL / SLODE_ENOSPC with a message, no crash."""
    import ctypes as C
    from structured_latent_odes_amd import _lib as L, engine as E
    dev = torch.device("cuda:0")
    eng = E.Engine(E.cvs_spec(3, 3, 2, solver="rk4"), 100, dev)
    eng.set_times(torch.arange(100.0))
    lib, h = eng.lib, eng.handle
    B = 8
    flat = torch.zeros(eng.n_params, device=dev)
    obs = torch.rand(B, 100, 3, device=dev).permute(0, 2, 1)
    u, eps, loss, grads = torch.zeros(B, 2, device=dev), torch.zeros(B, 8, device=dev), torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
    ws = eng.workspace(B)
    p = lambda t: C.c_void_p(t.data_ptr())
    strides = (C.c_int64 * 3)(*obs.stride())
    args = lambda **kw: [h, C.byref(eng.shape(B)), C.byref(eng.layout), kw.get("params", p(flat)), p(eng._times), p(eng._stage_t), p(obs), strides,
                         p(u), p(eps), p(loss), p(grads), None, None, p(ws), kw.get("ws_bytes", ws.numel() * 4), None]
    assert lib.slode_elbo_step(*args()) == 0
    assert lib.slode_elbo_step(*args(params=None)) == -1 and b"NULL" in lib.slode_last_error(h)
    assert lib.slode_elbo_step(*args(ws_bytes=1024)) == -3 and b"workspace" in lib.slode_last_error(h)
    bad = eng.shape(B).__class__.from_buffer_copy(eng.shape(B))
    bad.S = 7                                   # no kernel instantiation for ode_state_dim 7
    a = args(); a[1] = C.byref(bad)
    assert lib.slode_elbo_step(*a) == -1 and b"instantiated" in lib.slode_last_error(h)
    bad2 = eng.shape(B).__class__.from_buffer_copy(eng.shape(B))
    bad2.method = L.DOPRI5
    a = args(); a[1] = C.byref(bad2)
    assert lib.slode_elbo_step(*a) == -3 and b"workspace" in lib.slode_last_error(h)   # dopri5 training needs its record workspace
    assert lib.slode_adam_step(h, 10, p(flat), p(grads), p(flat), p(flat), 1e-3, 0.9, 0.999, 1e-8, 0, None) == -1   # step < 1
    # label heads that read overlapping latent ranges: only the auxiliary step refuses them (its kernel gives every head the latent-gradient
    # slots of its own dims); the main step, which does not score the heads of this family, still runs
    bad3 = eng.shape(B).__class__.from_buffer_copy(eng.shape(B))
    bad3.aux[1].z_off = bad3.aux[0].z_off
    aux_args = [h, C.byref(bad3), C.byref(eng.layout), p(flat), p(obs), strides, p(u), p(eps), p(loss), p(grads), p(ws), ws.numel() * 4,
                eng.n_params, None, None, C.c_float(0.0), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), 1, None]
    assert lib.slode_aux_step(*aux_args) == -1 and b"overlapping" in lib.slode_last_error(h)
    a = args(); a[1] = C.byref(bad3)
    assert lib.slode_elbo_step(*a) == 0
    assert lib.slode_elbo_step(*args()) == 0 and torch.isfinite(loss).all()      # the handle stays usable after errors


# ---- grad_mode = "reference_adjoint": torchdiffeq.odeint_adjoint's gradients, the reference default ------------
RA_CASES = {
    "cvs_rk4": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="rk4"), 9, 120),
    "cvs_c1_rk4_metric_shape": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="rk4"), 6, 200),   # shape-specialised instantiation
    "cvs_c0_rk4": ("cvs", dict(z_iext=1, z_rtpr=1, z_eps=2, solver="rk4"), 5, 100),                # shape-specialised instantiation
    "cvs_ref_default_midpoint": ("cvs", dict(), 7, 86),            # training_cvs.py defaults: midpoint, adjoint_solver=True
    "cvs_euler_gauss": ("cvs", dict(gauss=True, solver="euler"), 5, 64),
    "challenge_gauss_rk4": ("challenge", dict(gauss=True, solver="rk4"), 6, 150),
    "proc_rk4": ("proc", dict(z_g=3, z_eps=2, solver="rk4"), 6, 100),
}


@pytest.mark.parametrize("case", list(RA_CASES))
def test_reference_adjoint_gradients(case):
    """The HIP backward in reference_adjoint mode == the oracle's restatement of odeint_adjoint (fp64): continuous adjoint stepped
    backwards with the same fixed-grid method, no z -> dynamics gradient; the loss value is the same as in exact mode."""
    import dataclasses
    from structured_latent_odes_amd import engine as E
    fam, kw, B, T = RA_CASES[case]
    ospec = {"cvs": O.cvs_spec, "challenge": O.challenge_spec, "proc": O.proc_spec}[fam](**kw)
    ospec = dataclasses.replace(ospec, grad_mode="reference_adjoint")
    espec = dataclasses.replace({"cvs": E.cvs_spec, "challenge": E.challenge_spec, "proc": E.proc_spec}[fam](**kw), grad_mode="reference_adjoint")
    S = 8 if fam == "proc" else 5
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(5)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    dev = torch.device("cuda:0")
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    loss = torch.zeros(1, device=dev)
    grads = torch.full((eng.n_params,), float("nan"), device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads)
    p64 = {k: v.double() for k, v in p.items()}
    want_loss, want = O.loss_and_grads(p64, ospec, obs.double(), u.double(), eps.double(), times.double())
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 1e-5
    assert torch.isfinite(grads).all()
    got = eng.unpack(grads)
    bad = {k: _rel(v, want[k]) for k, v in got.items() if _rel(v, want[k]) > 5e-4}
    assert not bad, bad
    # and it is NOT the exact-mode gradient: the encoder sees no z -> dynamics term
    exact_spec = dataclasses.replace(ospec, grad_mode="exact")
    _, exact = O.loss_and_grads(p64, exact_spec, obs.double(), u.double(), eps.double(), times.double())
    assert _rel(got["encoder.z_loc.weight"], exact["encoder.z_loc.weight"]) > 1e-3
