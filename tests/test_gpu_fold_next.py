"""The in-launch fold (round 4; a measured arm, off by default -- SLODE_FOLD_NEXT=1): the chain-rule launch of a weight-updating step also folds the UPDATED encoder weights (W_eff, b_eff, row
sums, w', the likelihood-scale table) for the next step, which then starts without a fold launch (csrc/encoder_fused.hip, FOLD-NEXT;
include/slode.h, slode_fold_invalidate).  The fold inside the chain launch runs the same operations in the same order as weff_kernel, so
a training run with it must equal the same run with SLODE_FOLD_NEXT=0 BIT FOR BIT -- losses, gradients, weights, Adam moments."""
import numpy as np
import pytest
import torch

from oracle import slode_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    "cvs_metric_shape": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="rk4"), 40, 200, 5),          # fused encoder forward: no launch before the ODE kernel
    "cvs_ref_midpoint": ("cvs", dict(solver="midpoint"), 13, 86, 5),
    "challenge_gauss": ("challenge", dict(gauss=True, solver="rk4"), 9, 120, 5),                       # C = 4
    "proc_c_major": ("proc", dict(z_g=3, z_eps=2, solver="midpoint"), 7, 86, 8),                       # [B,C,T]-contiguous rows: kappa = c T + t
    "cvs_dopri5": ("cvs", dict(z_iext=3, z_rtpr=3, z_eps=2, solver="dopri5"), 20, 40, 5),
}


def _setup(case, monkeypatch, fold_next):
    from structured_latent_odes_amd import engine as E
    fam, kw, B, T, S = CASES[case]
    monkeypatch.setenv("SLODE_FOLD_NEXT", "1" if fold_next else "0")   # (the arm is off by default: include/slode.h)
    ospec = {"cvs": O.cvs_spec, "challenge": O.challenge_spec, "proc": O.proc_spec}[fam](**kw)
    espec = {"cvs": E.cvs_spec, "challenge": E.challenge_spec, "proc": E.proc_spec}[fam](**kw)
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(5)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    eng = E.Engine(espec, T, torch.device(DEV))          # a fresh handle: the switch is read in slode_create
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.contiguous().to(DEV) if fam == "proc" else obs.permute(0, 2, 1).contiguous().to(DEV).permute(0, 2, 1)
    return eng, flat, obs_d, u.to(DEV).contiguous(), eps.to(DEV).contiguous(), B


def _train(eng, flat, obs_d, u_d, eps_d, B, steps, with_aux, names=None):
    from structured_latent_odes_amd import _lib as L
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    loss, grads = torch.zeros(1, device=DEV), torch.zeros(eng.n_params, device=DEV)
    out, t = [], 0
    for it in range(steps):
        for kind in ([L.SVI_MAIN, L.SVI_AUX] if with_aux else [L.SVI_MAIN]):
            t += 1
            if names is not None:
                eng.profile_enable(True)
            eng.svi_step(kind, flat, eng.make_batch(obs_d, [u_d], eps_d), B, loss, grads, adam=(m, v, 1e-3, t, (0.9, 0.999), 1e-8))
            if names is not None:
                names.append([n for n, _ in eng.profile_read()])
                eng.profile_enable(False)
            out.append((loss.clone(), grads.clone()))
    return out, flat.clone(), m, v


@pytest.mark.parametrize("with_aux", [False, True])
@pytest.mark.parametrize("case", list(CASES))
def test_training_with_the_in_launch_fold_is_bitwise_the_same(case, with_aux, monkeypatch):
    names = []
    a = _train(*_setup(case, monkeypatch, True), steps=4, with_aux=with_aux, names=names)
    b = _train(*_setup(case, monkeypatch, False), steps=4, with_aux=with_aux)
    for (la, ga), (lb, gb) in zip(a[0], b[0]):
        assert torch.isfinite(la).all() and la.item() == lb.item()
        assert torch.equal(ga, gb)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # the first step folds at its start; every later one does not launch the fold kernel at all
    assert "weff" in names[0]
    assert all("weff" not in n for n in names[1:]), names


def test_weights_written_behind_the_engine_are_folded_again(monkeypatch):
    """load_state_dict between two steps (training_cvs.py:325-331 keeps a best-model copy that way): the model tells the engine, the next step
    folds the loaded weights -- and equals a fresh engine's step on those weights bit for bit.  Without the notice the kept fold would be
    that of the OLD weights: the test also shows that the two differ, i.e. that it can see a stale fold."""
    from structured_latent_odes_amd import _lib as L
    eng, flat, obs_d, u_d, eps_d, B = _setup("cvs_metric_shape", monkeypatch, True)
    _train(eng, flat, obs_d, u_d, eps_d, B, steps=2, with_aux=False)
    other = flat.clone()
    other[: eng.layout.lin_b] += 0.01                                  # conv + lin.weight: what the fold is made of
    bt = eng.make_batch(obs_d, [u_d], eps_d)

    def one_step(e, f):
        loss, grads = torch.zeros(1, device=DEV), torch.zeros(e.n_params, device=DEV)
        m, v = torch.zeros_like(f), torch.zeros_like(f)
        e.svi_step(L.SVI_MAIN, f, e.make_batch(obs_d, [u_d], eps_d), B, loss, grads, adam=(m, v, 1e-3, 1, (0.9, 0.999), 1e-8))
        return loss.clone(), grads.clone()

    # (a) a raw device-to-device copy the torch version counter does not see (what a re-pointed nn.Parameter's copy_ amounts to), WITH the notice
    torch.cuda.synchronize()
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(flat.data_ptr()), C.c_void_p(other.data_ptr()), C.c_size_t(flat.numel() * 4), 3) == 0
    eng.fold_invalidate()
    got = one_step(eng, flat)
    eng2, flat2, *_ = _setup("cvs_metric_shape", monkeypatch, True)
    flat2.copy_(other)
    want = one_step(eng2, flat2)
    assert got[0].item() == want[0].item() and torch.equal(got[1], want[1])
    # (b) the same write WITHOUT the notice scores the old fold: a different (wrong) loss -- the hazard slode_fold_invalidate exists for
    eng3, flat3, *_ = _setup("cvs_metric_shape", monkeypatch, True)
    _train(eng3, flat3, obs_d, u_d, eps_d, B, steps=2, with_aux=False)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(flat3.data_ptr()), C.c_void_p(other.data_ptr()), C.c_size_t(flat3.numel() * 4), 3) == 0
    stale = one_step(eng3, flat3)
    assert stale[0].item() != want[0].item()
    # (c) a torch-side write to the flat vector (or to a workspace) is seen by the engine's own guard: no notice needed
    eng4, flat4, *_ = _setup("cvs_metric_shape", monkeypatch, True)
    _train(eng4, flat4, obs_d, u_d, eps_d, B, steps=2, with_aux=False)
    flat4.copy_(other)
    seen = one_step(eng4, flat4)
    assert seen[0].item() == want[0].item() and torch.equal(seen[1], want[1])
    eng4.workspace(B).fill_(float("nan"))                               # the workspace carries the kept fold: scribbling on it is noticed too
    flat4.copy_(other)
    again = one_step(eng4, flat4)
    assert again[0].item() == want[0].item()


def test_model_level_load_state_dict_notifies_the_engine():
    from structured_latent_odes_amd.configs import load_config_cvs
    from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
    from structured_latent_odes_amd.svi import SVI, Adam, Trace_ELBO
    from structured_latent_odes_amd.synthetic import synthetic_batch
    dev = torch.device(DEV)
    cfg = load_config_cvs()
    cfg.update(seq_len=200, z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4", mini_batch_size=16)
    times = torch.arange(0.0, 200.0, device=dev)
    obs, labels, _ = synthetic_batch("cvs", 16, 200, 3, seed=3)
    batch = dict(observations=obs.to(dev), **{k: v.to(dev) for k, v in labels.items()})

    def fresh(seed):
        torch.manual_seed(seed)
        m = MechanisticModel(cfg, dev, times)
        return m, SVI(m.model, m.guide, Adam({"lr": 1e-3}), loss=Trace_ELBO())

    import os
    os.environ["SLODE_FOLD_NEXT"] = "1"
    try:
        m1, s1 = fresh(1)
        m1._bind()
    finally:
        del os.environ["SLODE_FOLD_NEXT"]
    m2, s2 = fresh(2)
    s1.step(**batch); s1.step(**batch)                       # m1's engine now keeps the fold of its own (updated) weights
    s2.step(**batch)
    sd = {k: v.clone() for k, v in m2.state_dict().items()}
    m1.load_state_dict(sd)                                   # the reference's best-model copy (training_cvs.py:329)
    m3, s3 = fresh(3)
    m3._bind()
    m3.load_state_dict(sd)
    for s in (s1, s3):
        s._impl.engine.rng_seed(77)
    assert s1.evaluate_loss(**batch) == s3.evaluate_loss(**batch)
