"""Short dopri5 run of BASELINE config[2] (proc, B=4096, T=100, L=50, S=8) for `rocprofv3 --kernel-trace --stats`."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structured_latent_odes_amd import configs as CF
from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed

dev = torch.device("cuda:0")
B, T = 4096, 100
set_seed(12)
cfg = CF.load_config_proc(); cfg.update(seq_len=T, solver="dopri5")
mod = importlib.import_module("structured_latent_odes_amd.models.mechanistic_proc")
obs, labels, times = synthetic_batch("proc", B, T, cfg.obs_dim)
m = mod.MechanisticModel(cfg, dev, times.to(dev)); b = m._bind(); eng, flat = b.engine, b.flat
obs_d = obs.to(dev); u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()}); eps_d = torch.randn(B, m.latent_dim, device=dev)
svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-4))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    t0 = time.perf_counter()
    loss = svi.step(obs_d, eps=eps_d, u=u_d)
    print("step %d  -ELBO/B %.4f  %.2f ms" % (i, loss / B, 1e3 * (time.perf_counter() - t0)), flush=True)
n = eng.dopri5_step_counts(B).float()
print("accepted steps per trajectory: min %d  mean %.1f  max %d;  mean over groups-of-64 maxima %.1f" % (n.min(), n.mean(), n.max(), n.view(-1, 64).max(1).values.mean()))
