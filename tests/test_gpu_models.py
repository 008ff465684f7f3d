"""GPU tests of the Python drop-in layer (models.*, svi.SVI/Adam/Trace_ELBO, training_cvs) against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import slode_oracle as O

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs() / b.abs().clamp_min(1.0)).max().item() < tol


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _cvs(gauss=False, solver="midpoint", T=86, B=24):
    from structured_latent_odes_amd.configs import load_config_cvs
    from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
    from structured_latent_odes_amd.models.mechanistic_cvs_Gauss import MechanisticModelGauss
    from structured_latent_odes_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    cfg = load_config_cvs()
    cfg.update(seq_len=T, solver=solver)
    torch.manual_seed(3)
    times = torch.arange(0.0, T * 1.0, 1.0, device=dev)
    m = (MechanisticModelGauss if gauss else MechanisticModel)(cfg, dev, times)
    with torch.no_grad():                      # move the classifier nets off their N(0, 1e-3) init so the aux test is sensitive
        for n, p in m.named_parameters():
            if n.startswith("q_"):
                p.add_(0.3 * torch.randn_like(p))
    obs, labels, _ = synthetic_batch("cvs", B, T, 3, seed=7)
    batch = {"observations": obs.to(dev), "iext": labels["iext"].to(dev), "rtpr": labels["rtpr"].to(dev)}
    import dataclasses
    ospec = dataclasses.replace(O.cvs_spec(gauss=gauss, solver=solver), grad_mode=m.model_spec().grad_mode)   # config: adjoint_solver=True
    assert ospec.grad_mode == "reference_adjoint"
    return m, cfg, batch, ospec, dev


def _oracle_params(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("gauss", [False, True])
def test_svi_main_matches_oracle_and_updates_parameters(gauss):
    from structured_latent_odes_amd.svi import SVI, Adam, Trace_ELBO
    m, cfg, batch, ospec, dev = _cvs(gauss=gauss)
    p0 = _oracle_params(m)
    svi = SVI(m.model, m.guide, Adam({"lr": 1e-3, "betas": (0.9, 0.999)}), loss=Trace_ELBO(num_particles=1))
    eps = torch.randn(24, m.latent_dim, generator=torch.Generator().manual_seed(5))
    u = torch.cat([batch["iext"], batch["rtpr"]], 1).cpu()
    want, g = O.loss_and_grads(p0, ospec, batch["observations"].cpu(), u, eps, m.times.cpu())
    ev = svi.evaluate_loss(eps=eps.to(dev), **batch)
    assert abs(ev - want.item()) / abs(want.item()) < 1e-5
    got = svi.step(eps=eps.to(dev), **batch)
    assert got == ev
    # one Adam step (first step: p -= lr * sign-ish(g)); compare against torch.optim.Adam on the oracle gradient
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items() if k in g and not k.startswith("q_") and ".prod." not in k and ".degr." not in k}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-3, betas=(0.9, 0.999))
    for k, v in ref.items():
        v.grad = g[k].clone()
    opt.step()
    p1 = _oracle_params(m)
    for k, v in ref.items():
        assert (p1[k] - v.detach()).abs().max().item() < 2e-6, k
    # parameters the main loss does not touch (auxiliary classifiers) got a zero gradient => Adam leaves them in place
    for k in p0:
        if k.startswith("q_"):
            assert torch.equal(p0[k], p1[k]), k


def test_aux_svi_matches_oracle():
    from structured_latent_odes_amd.svi import SVI, Adam
    m, cfg, batch, ospec, dev = _cvs()
    p0 = _oracle_params(m)
    svi = SVI(m.model_meta, m.guide_meta, Adam({"lr": 1e-3}))
    eps = torch.randn(24, m.latent_dim, generator=torch.Generator().manual_seed(6))
    u = torch.cat([batch["iext"], batch["rtpr"]], 1).cpu()
    q = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    want = O.aux_loss(q, ospec, batch["observations"].cpu(), u, eps)
    want.backward()
    ev = svi.evaluate_loss(eps=eps.to(dev), **batch)
    assert abs(ev - want.item()) / abs(want.item()) < 2e-5
    svi._impl.optimizer = None                      # inspect the raw gradient of one step
    svi.step(eps=eps.to(dev), **batch)
    b = m._bind()
    checked = 0
    for k, sl in b.slices.items():
        want_g = q[k].grad
        got_g = svi._impl.gbuf[sl]
        if k.startswith("encoder.") or k.startswith("q_"):
            assert _rel(got_g, want_g.reshape(-1)) < 5e-4, k
            checked += 1
        else:
            assert float(got_g.abs().max()) == 0.0, k           # the auxiliary loss does not touch the decoder / priors
    assert checked == 8 + 8


def test_recon_classifier_and_state_dict_roundtrip():
    m, cfg, batch, ospec, dev = _cvs(solver="rk4")
    for is_post in (True, False):
        r = m.recon(is_post=is_post, **batch)
        assert r["solution_xt"].shape == (24, 86, 5) and r["mu_50"].shape == (24, 3, 86) and r["std"].shape == (24, 3, 86)
        assert r["z"].shape == (24, 15) and torch.isfinite(r["l1"])
    pred = m.classifier(observations=batch["observations"])
    assert pred["iext"].shape == (24, 1) and set(pred["iext"].unique().tolist()) <= {0.0, 1.0}
    # decoder outputs of recon equal the oracle's for the same z
    r = m.recon(is_post=True, **batch)
    p = _oracle_params(m)
    sol, mu75, mu50, mu25, std = O.decoder_ald(p, r["z"].cpu(), m.times.cpu(), "rk4")
    assert _close(r["solution_xt"], sol)
    assert _rel(r["mu_75"], mu75) < 1e-5 and _rel(r["mu_25"], mu25) < 1e-5 and _rel(r["std"], std) < 1e-6
    # best-model copy (training_cvs.py:330) keeps the destination bound to its own flat vector
    from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
    best = MechanisticModel(cfg, dev, m.times)
    best._bind()
    best.load_state_dict(m.state_dict())
    bb = best._bind()
    assert torch.equal(bb.flat[:bb.engine.n_params], m._bind().flat[:bb.engine.n_params])
    r2 = best.decoder.forward(r["z"])
    assert torch.equal(r2[0], m.decoder.forward(r["z"])[0])


def test_module_level_autograd_matches_oracle():
    """EncoderCONV / OdeModel / Decoder used as ordinary autograd modules (standalone engines)."""
    from structured_latent_odes_amd.models.blackbox_ode import OdeModel
    from structured_latent_odes_amd.models.encoder_conv import EncoderCONV
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = EncoderCONV(n_channels=3, n_filters=10, filter_size=10, pool_size=5, n_time=100, latent_dim=8, hidden_dim=50).to(dev)
    x = torch.rand(6, 100, 3, device=dev).permute(0, 2, 1)
    loc, scale = enc(x)
    (loc.sum() + (scale * scale).sum()).backward()
    p = {"encoder." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    wl, ws = O.encoder_conv(p, x.cpu(), 5)
    (wl.sum() + (ws * ws).sum()).backward()
    assert _rel(loc, wl) < 2e-5 and _rel(scale, ws) < 2e-5
    assert _rel(enc.lin.weight.grad, p["encoder.lin.weight"].grad) < 3e-4
    assert _rel(enc.conv.weight.grad, p["encoder.conv.weight"].grad) < 3e-4
    assert _rel(enc.z_scale[0].bias.grad, p["encoder.z_scale.0.bias"].grad) < 3e-4

    for adjoint in (True, False):   # torchdiffeq.odeint_adjoint (reference default) / odeint
        om = OdeModel()
        times = torch.arange(0.0, 40.0, device=dev) * 0.5
        om.init_with_params(times=times, ode_state_dim=5, latent_dim=8, ode_hidden_dim=25, adjoint_solver=adjoint, solver="midpoint", device=dev)
        om.to(dev)
        z = torch.randn(7, 8, device=dev, requires_grad=True)
        sol = om.solve_ODE(z)
        w = torch.randn(7, 40, 5, device=dev)
        (sol * w).sum().backward()
        q = {"decoder.ode_model." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in om.state_dict().items() if ".prod." not in k and ".degr." not in k}
        zz = z.detach().cpu().clone().requires_grad_(True)
        want = O.solve_ode(q, zz, times.cpu(), "midpoint", grad_mode="reference_adjoint" if adjoint else "exact")
        (want * w.cpu()).sum().backward()
        assert _close(sol.detach(), want.detach())
        assert _rel(z.grad, zz.grad) < 5e-4
        assert _rel(om.dynamics.dynamics_hidden.weight.grad, q["decoder.ode_model.dynamics.dynamics_hidden.weight"].grad) < 5e-4
        assert _rel(om.latent_to_ode_net[2].weight.grad, q["decoder.ode_model.latent_to_ode_net.2.weight"].grad) < 5e-4
    # OdeFunc.forward(t, state) and initialize_state
    f = om.gen_dynamics(z.detach())
    st = torch.rand(7, 5, device=dev)
    got = f(torch.tensor(0.75), st)
    wantf = O.dynamics({k: v.detach() for k, v in q.items()}, torch.tensor(0.75), st.cpu(), z.detach().cpu())
    assert (got.cpu() - wantf).abs().max().item() < 2e-6
    assert (om.initialize_state(z.detach()).cpu() - O.initialize_state({k: v.detach() for k, v in q.items()}, z.detach().cpu())).abs().max().item() < 2e-6


def test_adam_kernel_matches_torch_adam():
    from structured_latent_odes_amd import engine as E
    from structured_latent_odes_amd.svi import FlatAdam
    dev = torch.device("cuda:0")
    eng = E.Engine(E.cvs_spec(), 86, dev)
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(5000, generator=g)
    flat = p0.clone().to(dev)
    opt = FlatAdam(eng, flat, lr=3e-4)
    ref = p0.clone().requires_grad_(True)
    ropt = torch.optim.Adam([ref], lr=3e-4, betas=(0.9, 0.999))
    for i in range(5):
        gr = torch.randn(5000, generator=g) * (1.0 if i != 3 else 0.0)     # incl. a zero-gradient step (two-SVI case)
        opt.step(gr.to(dev))
        ref.grad = gr.clone()
        ropt.step()
    assert (flat.cpu() - ref.detach()).abs().max().item() < 1e-6


@pytest.mark.parametrize("family", ["cvs", "challenge", "proc", "proc_dopri5"])
def test_training_entry_points_run(family):
    import importlib
    solver = "dopri5" if family.endswith("_dopri5") else "rk4"     # proc_dopri5: BASELINE config[2]'s solver, trained end to end
    family = family.split("_")[0]
    tc = importlib.import_module("training_" + family)
    cfg = tc.load_config()
    cfg.num_epochs, cfg.mini_batch_size = 2, 48
    if family == "proc":
        cfg.solver = solver
    var_model, best_model, best_epoch = tc.train(cfg, batches_per_epoch=3)
    assert 0 <= best_epoch <= 2
    assert all(torch.isfinite(p).all() for p in var_model.parameters())
    assert all(torch.isfinite(p).all() for p in best_model.parameters())


def test_recon_samples_is_one_batched_launch_of_recon(tmp_path):
    """`multiple_samples` (num_samples Monte-Carlo reconstructions) as one batched solve; same numbers as looping
    over `recon`-style single draws, reference file naming for the .npy dump."""
    import numpy as np
    m, cfg, batch, ospec, dev = _cvs(solver="rk4")
    ns, B = 5, batch["observations"].shape[0]
    eps = torch.randn(ns, B, m.latent_dim, generator=torch.Generator().manual_seed(2)).to(dev)
    res = m.recon_samples(batch["observations"], True, ns, eps=eps, iext=batch["iext"], rtpr=batch["rtpr"])
    assert res["mu_50"].shape == (B, 3, cfg.seq_len, ns) and res["z"].shape == (ns, B, m.latent_dim)
    p = _oracle_params(m)
    with torch.no_grad():
        loc, scale = O.encoder_conv(p, batch["observations"].cpu(), ospec.pool_size)
        for i in range(ns):
            z = loc + scale * eps[i].cpu()
            sol, mu75, mu50, mu25, std = O.decoder_ald(p, z, m.times.cpu(), "rk4")
            assert _close(res["mu_50"][..., i], mu50, 2e-5) and _close(res["mu_75"][..., i], mu75, 2e-5) and _close(res["mu_25"][..., i], mu25, 2e-5)
    # prior samples: conditional prior for the label groups, N(0, 1) for z_epsilon
    resp = m.recon_samples(batch["observations"], False, 3, iext=batch["iext"], rtpr=batch["rtpr"])
    assert resp["mu_50"].shape == (B, 3, cfg.seq_len, 3) and torch.isfinite(resp["mu_50"]).all()
    files = m.save_recon_samples(str(tmp_path / "results_Mechanistic"), batch["observations"], True, 4, iext=batch["iext"], rtpr=batch["rtpr"])
    assert sorted(os.path.basename(f) for f in files) == ["mu_25_post_sample.npy", "mu_50_post_sample.npy", "mu_75_post_sample.npy"]
    assert np.load(files[0]).shape == (B, 3, cfg.seq_len, 4)


def test_training_on_reference_format_files(tmp_path):
    """SURVEY row N3 end to end: cvs / challenge / proc files in the reference's on-disk formats -> data.py readers, transforms, splits ->
    pinned-buffer feeder -> the epoch loop (two SVI objects, one Adam)."""
    import pickle
    import training_cvs, training_challenge
    from structured_latent_odes_amd import training as TR
    rng = np.random.default_rng(0)
    d = str(tmp_path) + "/"
    obs = {"train": rng.random((70, 86, 3)) * 40 + 60, "test": rng.random((9, 86, 3)) * 40 + 60}
    torch.save(obs, d + "processed_data.pkl")
    for name, n in (("train", 70), ("test", 9)):
        torch.save({"i_ext": rng.choice([0.0, -0.2], size=n), "r_tpr_mod": rng.choice([0.0, 0.5], size=n)}, d + name + "_params_data.pkl")
    from structured_latent_odes_amd.data import find_norm_params
    torch.save(find_norm_params(obs["train"]), d + "data_norm_params.pkl")
    cfg = training_cvs.load_config()
    cfg.num_epochs, cfg.mini_batch_size = 1, 24
    trb, vab, _, teb = TR.real_batches(cfg, "cvs", d)
    assert len(trb) == 3 and len(vab) == 1 and len(teb) == 1     # 63 train / 7 val / 9 test series
    b0 = next(iter(trb))
    assert b0["observations"].is_cuda and b0["observations"].shape == (24, 3, 86) and b0["observations"].stride() == (258, 1, 3)   # [B,C,T] view of [B,T,C]
    assert -1e-6 <= float(b0["observations"].min()) and float(b0["observations"].max()) <= 1.0 + 1e-6
    vm, bm, be = training_cvs.train(cfg, train_batches=trb, val_batches=vab, test_batches=teb)   # incl. the final test passes
    assert all(torch.isfinite(p).all() for p in vm.parameters())
    with open(d + "data.pkl", "wb") as fh:
        pickle.dump({"observations": rng.random((35, 142, 4)), "shedding": rng.integers(0, 2, (35, 1)).astype(float),
                     "symptoms": rng.integers(0, 2, (35, 1)).astype(float), "n_time": 142}, fh)
    cfg = training_challenge.load_config()
    cfg.num_epochs, cfg.mini_batch_size = 1, 16
    trb, vab, _ = TR.real_batches(cfg, "challenge", d)
    assert len(trb) == 2 and len(vab) == 1                       # 28 / 7 by the seeded 5-fold split
    vm, bm, be = training_challenge.train(cfg, train_batches=trb, val_batches=vab)
    assert all(torch.isfinite(p).all() for p in vm.parameters())
    # proc: plate-reader CSVs -> cassettes / inputs labels, the data's own (non-uniform) time grid
    import csv
    import training_proc
    devs = ["Pcat_Y81C76", "RS100S32_Y81C76", "RS100S34_Y81C76", "R33S32_Y81C76", "R33S34_Y81C76", "R33S175_Y81C76"]
    T = 40
    tgrid = np.cumsum(rng.uniform(0.1, 0.3, T))
    sigs = ["EYFP", "ECFP", "mRFP1", "OD"]
    with open(d + "proc140916.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Content", "Colony", "Well Col", "Well Row", "Content"] + ["Raw Data (%s) %d - x" % (sg, k + 1) for sg in sigs for k in range(T)])
        w.writerow(["", "", "", "", ""] + [float(t) for _ in sigs for t in tgrid])
        for i in range(40):
            cond = ("C6=%g" % rng.choice([0.0, 1.5, 250.0])) if i % 2 else ("C12=%g" % rng.choice([0.0, 3.8, 2777.0]))
            curve = [float(50 + 10 * si + 30 * (1 - np.exp(-t)) + rng.normal() * 0.5) for si in range(4) for t in tgrid]
            w.writerow([devs[i % 6], "", str(i % 12 + 1), "A", cond] + curve)
    cfg = training_proc.load_config()
    cfg.num_epochs, cfg.mini_batch_size, cfg.solver = 1, 12, "rk4"
    trb, vab, times = TR.real_batches(cfg, "proc", d)
    assert cfg.seq_len == T and times.shape == (T,) and len(trb) == 3 and len(vab) == 1      # 30 / 10 by the seeded 4-fold split
    b0 = next(iter(trb))
    assert b0["observations"].shape == (12, 4, T) and b0["aR"].shape == (12, 3) and b0["aS"].shape == (12, 4) and b0["C6"].shape == (12, 1)
    vm, bm, be = training_proc.train(cfg, train_batches=trb, val_batches=vab, times=times)
    assert all(torch.isfinite(p).all() for p in vm.parameters())


def test_two_svi_objects_share_adam_with_per_parameter_step_counts():
    """the reference alternates SVI(model, guide).step and SVI(model_meta, guide_meta).step on ONE pyro.optim.Adam
    (training_cvs.py:147-157, 226-249).  pyro.optim keeps a torch.optim.Adam per parameter; both objects register every parameter
    (pyro.module(..., self)), so each parameter is stepped in both, with a zero gradient by the loss that does not use it -- except
    that a parameter whose .grad is still None is skipped: the label heads in the very first main step.  Emulated here on the CPU
    with the oracle's gradients; the HIP path must end up with the same parameters after main, aux, main, aux."""
    from structured_latent_odes_amd.svi import SVI, Adam, Trace_ELBO
    m, cfg, batch, ospec, dev = _cvs(solver="rk4")
    lr, b1, b2, aeps = 1e-3, 0.9, 0.999, 1e-8
    opt = Adam({"lr": lr, "betas": (b1, b2)})
    main = SVI(m.model, m.guide, opt, loss=Trace_ELBO(num_particles=1))
    aux = SVI(m.model_meta, m.guide_meta, opt, loss=Trace_ELBO(num_particles=1))
    p = {k: v.double() for k, v in _oracle_params(m).items() if ".prod." not in k and ".degr." not in k}
    st = {k: dict(m=torch.zeros_like(v), v=torch.zeros_like(v), n=0, seen=False) for k, v in p.items()}
    obs, u = batch["observations"].cpu().double(), torch.cat([batch["iext"], batch["rtpr"]], 1).cpu().double()
    times = m.times.cpu().double()
    g = torch.Generator().manual_seed(9)
    for it in range(4):
        eps = torch.randn(24, m.latent_dim, generator=g)
        if it % 2 == 0:
            main.step(eps=eps.to(dev), **batch)
            _, grads = O.loss_and_grads(p, ospec, obs, u, eps.double(), times)
            used = lambda k: not k.startswith("q_")
        else:
            aux.step(eps=eps.to(dev), **batch)
            q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
            O.aux_loss(q, ospec, obs, u, eps.double()).backward()
            grads = {k: (q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])) for k in q}
            used = lambda k: k.startswith("q_") or k.startswith("encoder.")
        for k in p:                                   # one torch.optim.Adam per parameter
            s = st[k]
            if used(k):
                s["seen"] = True
            if not s["seen"]:
                continue                              # .grad is None: torch.optim.Adam skips it
            gk = grads[k] if used(k) else torch.zeros_like(p[k])
            s["n"] += 1
            s["m"] = s["m"] + (1 - b1) * (gk - s["m"])
            s["v"] = b2 * s["v"] + (1 - b2) * gk * gk
            denom = s["v"].sqrt() / (1 - b2 ** s["n"]) ** 0.5 + aeps
            p[k] = p[k] - (lr / (1 - b1 ** s["n"])) * s["m"] / denom
    assert st["q_iext.sequential_mlp.0.0.module.weight" if "q_iext.sequential_mlp.0.0.module.weight" in st else next(k for k in st if k.startswith("q_"))]["n"] == 3
    assert st["encoder.conv.weight"]["n"] == 4
    got = _oracle_params(m)
    for k, v in p.items():
        err = (got[k].double() - v).abs().max().item()
        assert err < 3e-6, (k, err)
