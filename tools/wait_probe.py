"""Diagnostic (GPU box): what the runtime's wait mode costs a 20-step block of the metric step (the driver's bench command brackets every
20 steps by synchronize).  Run once per setting of ROC_ACTIVE_WAIT_TIMEOUT (microseconds the host spins on a completion signal before it
sleeps on the interrupt; the runtime's default is 0).   Usage: [ROC_ACTIVE_WAIT_TIMEOUT=us] python tools/wait_probe.py"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structured_latent_odes_amd.configs import load_config_cvs
from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed

dev = torch.device("cuda:0")
cfg = load_config_cvs(); cfg.update(seq_len=200, z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4")
set_seed(cfg.seed)
m = MechanisticModel(cfg, dev, torch.arange(0.0, 200.0, device=dev)); b = m._bind()
obs, labels, _ = synthetic_batch("cvs", 1024, 200, 3); obs_d = obs.to(dev)
u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
eps_d = torch.randn(1024, 8, generator=torch.Generator().manual_seed(99)).to(dev)
svi = ELBOStep(b.engine, b.flat, FlatAdam(b.engine, b.flat, lr=cfg.learning_rate))
step = lambda: svi.step_async(obs_d, eps=eps_d, u=u_d)
for _ in range(20):
    for _ in range(50): step()
    torch.cuda.synchronize(dev)
for K in (20, 200):
    out = []
    for _ in range(41):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize(dev)
        out.append(1e6 * (time.perf_counter() - t0) / K)
    print("ROC_ACTIVE_WAIT_TIMEOUT=%s  blocks of %3d steps: median %.2f us/step  min %.2f" % (os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT", "(unset)"), K, statistics.median(out), min(out)), flush=True)
