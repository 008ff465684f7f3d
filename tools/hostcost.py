"""Diagnostic: host cost per call and step time as the weights evolve (GPU box)."""
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
from structured_latent_odes_amd.configs import load_config_cvs
from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed
dev = torch.device("cuda:0")
cfg = load_config_cvs(); cfg.update(seq_len=200, z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4")
if len(sys.argv) > 1: set_seed(int(sys.argv[1]))
times = torch.arange(0.0, 200.0, device=dev)
m = MechanisticModel(cfg, dev, times); b = m._bind(); eng, flat = b.engine, b.flat
obs, labels, _ = synthetic_batch("cvs", 1024, 200, 3); obs_d = obs.to(dev)
u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
eps_d = torch.randn(1024, 8, generator=torch.Generator().manual_seed(99)).to(dev)
svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-3))
for blk in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): svi.step_async(obs_d, eps=eps_d, u=u_d)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.profile_enable(True); svi.step_async(obs_d, eps=eps_d, u=u_d); pr = eng.profile_read(); eng.profile_enable(False)
    print("block %d: %.1f us/step  loss/traj %.2f  kernels %s" % (blk, 1e6*(t1-t0)/50, svi.loss.item()/1024, {k: round(v,1) for k,v in pr}))
