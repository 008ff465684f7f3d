"""Every compiled instantiation of the fused ODE/ELBO kernel is a parity-test case.

The launcher (csrc/ode_kernel.hip, slode_launch_ode) picks, for a backward launch, one of
    {five shape-specialised instantiations, generic}  x  {loop-free, persistent loop}  x  {exact, reference_adjoint}
and for a forward-only launch {specialised, generic}.  Round 1 shipped persistent-loop kernels no test reached, one of which was
wrong (a compiler-placed VGPR spill under the wrong EXEC mask: DESIGN 3.1).  Here every combination runs against the fp64 oracle:
  * loop-free: one workgroup per trajectory (the default up to 65,536 trajectories);
  * persistent loop: forced through the handle flags SLODE_ODE_LOOP + SLODE_ODE_GRID=5 (read once at slode_create), so that each of 5
    workgroups integrates 2-3 trajectories (ragged) and the per-workgroup slab really is accumulated across trajectories;
  * generic: the same shapes with SLODE_ODE_GENERIC (no compile-time shape), plus the randomised sweep of test_gpu_fuzz.py;
  * the measured A/B arms of the metric shape (SLODE_ODE_ALG = 1: direct evaluation of the dynamics heads; 2: + MFMA contraction).
The workspace is filled with NaN before every step: a gradient element no thread of the launch owns shows up as a NaN.
Tolerances as in test_gpu_parity.py: -ELBO 1e-5 relative, gradients 5e-4 norm-wise per tensor vs the fp64 oracle.
"""
import dataclasses

import pytest
import torch

from oracle import slode_oracle as O

pytestmark = pytest.mark.gpu

SHAPES = {
    # the launcher's SLODE_STATIC list (ode_kernel.hip): (family, spec kwargs, T)
    "c1_cvs_T200_L8_rk4": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="rk4"), 200),
    "c0_cvs_T100_L4_rk4": ("cvs", dict(z_iext=1, z_rtpr=1, z_eps=2, solver="rk4"), 100),
    "c2_proc_T100_L50_rk4": ("proc", dict(z_g=10, z_eps=10, solver="rk4"), 100),
    "c4_challenge_gauss_T300_L15_rk4": ("challenge", dict(gauss=True, solver="rk4"), 300),
    "ref_cvs_T86_L15_midpoint": ("cvs", dict(solver="midpoint"), 86),
}
B = 12
_cache = {}


def _case(shape, mode):
    """Inputs + fp64 oracle loss/gradients, computed once per (shape, gradient mode)."""
    key = (shape, mode)
    if key not in _cache:
        fam, kw, T = SHAPES[shape]
        ospec = {"cvs": O.cvs_spec, "challenge": O.challenge_spec, "proc": O.proc_spec}[fam](**kw)
        ospec = dataclasses.replace(ospec, grad_mode=mode)
        S = 8 if fam == "proc" else 5
        p = O.init_params(ospec, T=T, S=S)
        g = torch.Generator().manual_seed(17)
        p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
        obs, u, eps, times = O.synthetic_batch(ospec, B, T)
        p64 = {k: v.double() for k, v in p.items()}
        want_loss, want = O.loss_and_grads(p64, ospec, obs.double(), u.double(), eps.double(), times.double())
        _cache[key] = dict(fam=fam, kw=kw, T=T, S=S, p=p, obs=obs, u=u, eps=eps, times=times, want_loss=want_loss, want=want)
    return _cache[key]


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _run(c, mode, env, monkeypatch):
    from structured_latent_odes_amd import engine as E
    for k in ("SLODE_ODE_LOOP", "SLODE_ODE_GRID", "SLODE_ODE_GENERIC", "SLODE_ODE_ALG", "SLODE_ODE_PACK", "SLODE_ENC_FUSE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = torch.device("cuda:0")
    espec = {"cvs": E.cvs_spec, "challenge": E.challenge_spec, "proc": E.proc_spec}[c["fam"]](**c["kw"])
    espec = dataclasses.replace(espec, grad_mode=mode)
    eng = E.Engine(espec, c["T"], dev)                # a fresh handle: the flags are read in slode_create
    eng.set_times(c["times"])
    flat = eng.pack(c["p"])
    obs = c["obs"]
    obs_d = obs.contiguous().to(dev) if c["fam"] == "proc" else obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    u_d, eps_d = c["u"].to(dev).contiguous(), c["eps"].to(dev).contiguous()
    outs = []
    for rep in range(2):
        eng.workspace(B).fill_(float("nan"))          # nothing may survive from an earlier launch
        loss = torch.full((1,), float("nan"), device=dev)
        grads = torch.full((eng.n_params,), float("nan"), device=dev)
        eng.elbo_step(flat, obs_d, u_d, eps_d, loss, grads=grads)
        outs.append((loss.clone(), grads.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # bitwise reproducible
    loss, grads = outs[0]
    assert torch.isfinite(loss).all() and torch.isfinite(grads).all()
    assert abs(loss.item() - c["want_loss"].item()) / abs(c["want_loss"].item()) < 1e-5
    got = eng.unpack(grads)
    bad = {k: _rel(v, c["want"][k]) for k, v in got.items() if _rel(v, c["want"][k]) > 5e-4}
    assert not bad, bad
    # the forward-only instantiation scores the same loss (SVI.evaluate_loss)
    loss2 = torch.zeros(1, device=dev)
    eng.elbo_step(flat, obs_d, u_d, eps_d, loss2, grads=None)
    assert abs(loss2.item() - loss.item()) <= 2e-6 * abs(loss.item())
    return loss, grads


FORMS = {
    "loop_free": {},
    "persistent_loop": {"SLODE_ODE_LOOP": "1", "SLODE_ODE_GRID": "5"},
    "generic_loop_free": {"SLODE_ODE_GENERIC": "1"},
    "generic_persistent_loop": {"SLODE_ODE_GENERIC": "1", "SLODE_ODE_LOOP": "1", "SLODE_ODE_GRID": "5"},
}


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("shape", list(SHAPES))
def test_instantiation_matches_oracle(shape, form, mode, monkeypatch):
    _run(_case(shape, mode), mode, FORMS[form], monkeypatch)


def test_specialised_and_generic_forms_agree(monkeypatch):
    """The specialised and the generic loop-free kernels run the same arithmetic; the only difference in ORDER is the weight-gradient
    contraction's whole-chunk part (prefix / suffix sums of the chunk sums in the specialised kernels, masked adds in the generic one):
    same loss bit for bit, gradients to rounding."""
    c = _case("c1_cvs_T200_L8_rk4", "exact")
    a = _run(c, "exact", FORMS["loop_free"], monkeypatch)
    b = _run(c, "exact", FORMS["generic_loop_free"], monkeypatch)
    assert torch.equal(a[0], b[0])
    assert _rel(a[1], b[1]) < 2e-6


@pytest.mark.parametrize("alg", [1, 2])
def test_ab_arms_of_the_metric_shape(alg, monkeypatch):
    """The A/B arms measured in DESIGN 5 compute the same step: ALG 1 evaluates the dynamics heads directly (round-1 code), ALG 2
    also contracts the weight gradients on v_mfma_f32_16x16x4_f32."""
    c = _case("c1_cvs_T200_L8_rk4", "exact")
    _run(c, "exact", {"SLODE_ODE_ALG": str(alg)}, monkeypatch)


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
def test_packed_arm_of_the_metric_shape(mode, monkeypatch):
    """The measured-and-rejected decomposition of round 3: four trajectories per workgroup of 1024 threads, each on its own
    waves, LDS region and slab row, sharing only the barriers (SLODE_ODE_PACK=4; B = 12 = three packed workgroups).  Same step."""
    c = _case("c1_cvs_T200_L8_rk4", mode)
    _run(c, mode, {"SLODE_ODE_PACK": "4"}, monkeypatch)


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
@pytest.mark.parametrize("pack", ["12", "14"])
def test_packed_arms_with_shared_encoder_pass_and_own_barriers(pack, mode, monkeypatch):
    """Round 4: 2 (SLODE_ODE_PACK=12) or 4 (=14) trajectories per workgroup on disjoint waves, ONE pass over W_eff for
    all of them in the set-up (the encoder forward), then independent progress -- every trajectory on barriers of its own (an LDS arrival
    counter per trajectory, soft_barrier in ode_kernel.hip) instead of the workgroup's s_barrier.  Same step as the shipped form: against the
    oracle, NaN-poisoned workspace, bitwise repeat (inside _run), and against the shipped kernel to summation order."""
    c = _case("c1_cvs_T200_L8_rk4", mode)
    a = _run(c, mode, {"SLODE_ODE_PACK": pack}, monkeypatch)
    b = _run(c, mode, {}, monkeypatch)
    assert abs(a[0].item() - b[0].item()) <= 2e-6 * abs(b[0].item())
    assert _rel(a[1], b[1]) < 2e-5


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
def test_metric_shape_with_the_separate_encoder_launch(mode, monkeypatch):
    """The metric shape's loop-free kernel runs the encoder forward of its own trajectory by default (ENCF, ode_kernel.hip);
    SLODE_ENC_FUSE=0 restores the separate enc_fwd2 launch.  Both score the same step: each against the oracle, and the two losses agree
    to fp32 summation order (the folded product is summed in a different order: wave-per-13-rows against wave-per-4-rows)."""
    c = _case("c1_cvs_T200_L8_rk4", mode)
    a = _run(c, mode, {"SLODE_ENC_FUSE": "0"}, monkeypatch)
    b = _run(c, mode, {}, monkeypatch)
    assert abs(a[0].item() - b[0].item()) <= 2e-6 * abs(b[0].item())
    assert _rel(a[1], b[1]) < 2e-5


def test_fused_encoder_forward_reads_either_dense_layout():
    """The metric shape's kernel reads its trajectory's observation row in MEMORY order for the fused encoder forward (the fold launch lays
    W_eff's columns out the same way): a [B,C,T]-contiguous batch must score the same step as the native [B,T,C]-contiguous one."""
    from structured_latent_odes_amd import engine as E
    c = _case("c1_cvs_T200_L8_rk4", "exact")
    dev = torch.device("cuda:0")
    eng = E.Engine(E.cvs_spec(**c["kw"]), c["T"], dev)
    eng.set_times(c["times"])
    flat = eng.pack(c["p"])
    u_d, eps_d = c["u"].to(dev).contiguous(), c["eps"].to(dev).contiguous()
    outs = []
    for obs_d in (c["obs"].permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1), c["obs"].contiguous().to(dev)):
        loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
        eng.workspace(B).fill_(float("nan"))
        eng.elbo_step(flat, obs_d, u_d, eps_d, loss, grads=grads)
        outs.append((loss.clone(), grads.clone()))
    assert abs(outs[0][0].item() - c["want_loss"].item()) / abs(c["want_loss"].item()) < 1e-5
    assert abs(outs[0][0].item() - outs[1][0].item()) <= 2e-6 * abs(outs[0][0].item())
    assert _rel(outs[1][1], outs[0][1]) < 2e-5


def test_non_monotone_time_grid_is_rejected():
    """torchdiffeq raises on a grid that is not strictly monotone; so does the host side, and the kernel turns the loss into NaN."""
    from structured_latent_odes_amd import engine as E
    dev = torch.device("cuda:0")
    eng = E.Engine(E.cvs_spec(3, 3, 2, solver="rk4"), 64, dev)
    t = torch.arange(64.0)
    t[10] = 30.0
    with pytest.raises(ValueError):
        eng.set_times(t)
