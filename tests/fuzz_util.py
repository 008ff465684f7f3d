"""Randomised parity cases shared by tests/test_gpu_fuzz.py and tools/fuzz_parity.py: random family / T / latent split / batch /
solver / likelihood / gradient mode, HIP ELBO step vs the fp64 oracle."""
import dataclasses

import torch

from oracle import slode_oracle as O

T_CHOICES = [24, 33, 64, 86, 100, 129, 200, 257, 300, 333]
T_CHOICES_PROC = [24, 50, 100, 150]


def run_case(rng, case: int, dev):
    """Returns (ok, description)."""
    from structured_latent_odes_amd import engine as E
    fam = rng.choice(["cvs", "cvs", "challenge", "proc"])
    solver = rng.choice(["euler", "midpoint", "rk4", "rk4"])
    gauss = rng.random() < 0.4
    mode = rng.choice(["exact", "exact", "reference_adjoint"])
    T = rng.choice(T_CHOICES) if fam != "proc" else rng.choice(T_CHOICES_PROC)
    B = rng.randint(1, 14)
    if fam == "cvs":
        kw = dict(z_iext=rng.randint(1, 6), z_rtpr=rng.randint(1, 6), z_eps=rng.randint(1, 6))
    elif fam == "challenge":
        kw = dict(z_shed=rng.randint(1, 6), z_symp=rng.randint(1, 6), z_eps=rng.randint(1, 6))
    else:
        kw = dict(z_g=rng.randint(1, 5), z_eps=rng.randint(1, 6))
    kw.update(gauss=gauss, solver=solver)
    ospec = dataclasses.replace({"cvs": O.cvs_spec, "challenge": O.challenge_spec, "proc": O.proc_spec}[fam](**kw), grad_mode=mode)
    espec = dataclasses.replace({"cvs": E.cvs_spec, "challenge": E.challenge_spec, "proc": E.proc_spec}[fam](**kw), grad_mode=mode)
    S = 8 if fam == "proc" else 5
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(case)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T, seed=100 + case)
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1) if fam != "proc" else obs.to(dev)
    loss = torch.zeros(1, device=dev)
    grads = torch.full((eng.n_params,), float("nan"), device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, grads=grads)
    p64 = {k: v.double() for k, v in p.items()}
    want_loss, want = O.loss_and_grads(p64, ospec, obs.double(), u.double(), eps.double(), times.double())
    lrel = abs(loss.item() - want_loss.item()) / abs(want_loss.item())
    got = eng.unpack(grads)
    worst, wk = 0.0, ""
    for k, v in got.items():
        e = ((v.double().cpu() - want[k]).norm() / want[k].norm().clamp_min(1e-30)).item()
        if not (e <= worst):
            worst, wk = e, k
    ok = lrel < 1e-5 and worst < 5e-4 and bool(torch.isfinite(grads).all())
    # the auxiliary step (second SVI object) on the same model and batch
    aloss = torch.zeros(1, device=dev)
    agrads = torch.full((eng.n_params,), float("nan"), device=dev)
    eng.aux_step(flat, obs_d, u.to(dev), eps.to(dev), aloss, agrads)
    q = {k: v.double().requires_grad_(True) for k, v in p.items()}
    awant = O.aux_loss(q, ospec, obs.double(), u.double(), eps.double())
    awant.backward()
    alrel = abs(aloss.item() - awant.item()) / abs(awant.item())
    aworst = 0.0
    for k, v in eng.unpack(agrads).items():
        w = q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])
        if float(w.abs().max()) == 0.0:
            e = float(v.abs().max())
        else:
            e = ((v.double().cpu() - w).norm() / w.norm()).item()
        if not (e <= aworst):
            aworst = e
    ok = ok and alrel < 1e-5 and aworst < 5e-4 and bool(torch.isfinite(agrads).all())
    desc = "%-3d %-9s T=%-3d B=%-2d L=%-2d %-8s %-5s %-17s loss rel %.1e  worst grad %.1e (%s)  aux: loss rel %.1e worst grad %.1e" % (
        case, fam, T, B, espec.latent_dim, solver, "gauss" if gauss else "ald", mode, lrel, worst, wk, alrel, aworst)
    return ok, desc
