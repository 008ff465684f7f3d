"""Diagnostic: per-call host enqueue time over many async steps, to catch one-off runtime stalls (GPU box)."""
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from structured_latent_odes_amd.configs import load_config_cvs
from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed
dev = torch.device("cuda:0")
cfg = load_config_cvs(); cfg.update(seq_len=200, z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4")
set_seed(12)
times = torch.arange(0.0, 200.0, device=dev)
m = MechanisticModel(cfg, dev, times); b = m._bind(); eng, flat = b.engine, b.flat
obs, labels, _ = synthetic_batch("cvs", 1024, 200, 3); obs_d = obs.to(dev)
u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
eps_d = torch.randn(1024, 8, generator=torch.Generator().manual_seed(99)).to(dev)
svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-3))
for _ in range(20): svi.step_async(obs_d, eps=eps_d, u=u_d)
torch.cuda.synchronize()
nlaunch = 0
for trial in range(12):
    ts = [time.perf_counter()]
    for _ in range(200):
        svi.step_async(obs_d, eps=eps_d, u=u_d); ts.append(time.perf_counter())
    t_enq = time.perf_counter(); torch.cuda.synchronize(); tend = time.perf_counter()
    d = [1e6*(ts[i+1]-ts[i]) for i in range(200)]
    big = [(i, round(x)) for i, x in enumerate(d) if x > 1000]
    print("trial %2d: total %.1f us/step; enqueue %.1f ms, final sync wait %.1f ms; slow calls (idx, us): %s" % (trial, 1e6*(tend-ts[0])/200, sum(d)/1e3, 1e3*(tend-t_enq), big[:8]), flush=True)
