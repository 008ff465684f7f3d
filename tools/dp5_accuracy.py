"""Diagnostic: where the fp32 adaptive solve sits relative to the fp64 oracle at EQUAL tolerances, against the tight
solution -- the cvs case of tests/test_gpu_parity.py::test_dopri5_elbo_step_solution_level, B = 38.
  python tools/dp5_accuracy.py                      (the shipped library)
  SLODE_LIB_PATH=.../libslode_dp5p.so python tools/dp5_accuracy.py   (make dp5precise: IEEE controller arithmetic, fp64 dense output)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slode_oracle as O
from structured_latent_odes_amd import engine as E

torch.set_num_threads(8)
dev = torch.device("cuda:0")
S, T, B = 5, 60, 38
kw = dict(z_iext=3, z_rtpr=3, z_eps=2)
for rtol, atol in ((1e-6, 1e-8), (1e-7, 1e-9)):
    ospec = O.cvs_spec(solver="dopri5", **kw)
    espec = E.cvs_spec(solver="dopri5", **kw)
    espec.rtol, espec.atol = rtol, atol
    p = O.init_params(ospec, T=T, S=S)
    g = torch.Generator().manual_seed(31)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    times = times * 0.25
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
    loss, x = torch.zeros(1, device=dev), torch.empty(B, T, S, device=dev)
    eng.elbo_step(flat, obs_d, u.to(dev), eps.to(dev), loss, None, x_out=x)
    p64 = {k: v.double() for k, v in p.items()}
    with torch.no_grad():
        ospec.solver_kw = dict(rtol=1e-10, atol=1e-12, per_trajectory=True)
        tight_loss, tp = O.main_loss(p64, ospec, obs.double(), u.double(), eps.double(), times.double(), return_parts=True)
        ospec.solver_kw = dict(rtol=rtol, atol=atol, per_trajectory=True)
        same_loss, sp = O.main_loss(p64, ospec, obs.double(), u.double(), eps.double(), times.double(), return_parts=True)
    tight, same = tp["dec"][0], sp["dec"][0]
    rel = lambda a: abs(a - tight_loss.item()) / abs(tight_loss.item())
    ex = lambda a: (a - tight).abs().max().item()
    print("rtol %.0e atol %.0e | -ELBO vs tight: engine fp32 %.2e  oracle fp64 at the same tolerances %.2e | max |x - x_tight|: engine %.2e  oracle %.2e | steps %s"
          % (rtol, atol, rel(loss.item()), rel(same_loss.item()), ex(x.cpu().double()), ex(same), eng.dopri5_step_counts(B).float().mean().item()))
