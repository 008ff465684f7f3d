"""Throughput of the other BASELINE.json configs (parity-test shapes, not the headline bench line): full step = main ELBO fwd+bwd + Adam."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structured_latent_odes_amd import configs as CF
from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed
import importlib

dev = torch.device("cuda:0")
CASES = [("config[0] cvs B=32 T=100 L=4 rk4", "cvs", False, 32, 100, dict(z_iext_dim=1, z_rtpr_dim=1, z_epsilon_dim=2, solver="rk4")),
         ("config[1] cvs B=1024 T=200 L=8 rk4", "cvs", False, 1024, 200, dict(z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4")),
         ("config[2] proc B=4096 T=100 L=50 S=8 rk4 (fixed-grid stand-in for dopri5)", "proc", False, 4096, 100, dict(solver="rk4")),
         ("config[2] proc B=4096 T=100 L=50 S=8 dopri5 (rtol 1e-7, atol 1e-9: torchdiffeq defaults)", "proc", False, 4096, 100, dict(solver="dopri5")),
         ("config[4] challenge-Gauss B=512 T=300 L=15 rk4", "challenge", True, 512, 300, dict(solver="rk4")),
         ("reference default cvs B=128 T=86 L=15 midpoint", "cvs", False, 128, 86, dict())]
only = sys.argv[1] if len(sys.argv) > 1 else ""      # substring filter on the case name
for name, fam, gauss, B, T, kw in CASES:
    if only not in name:
        continue
    set_seed(12)
    cfg = getattr(CF, "load_config_" + fam)(); cfg.update(seq_len=T, **kw)
    mod = importlib.import_module("structured_latent_odes_amd.models.mechanistic_%s%s" % (fam, "_Gauss" if gauss else ""))
    cls = getattr(mod, "MechanisticModelGauss" if gauss else "MechanisticModel")
    obs, labels, times = synthetic_batch(fam, B, T, cfg.obs_dim)
    m = cls(cfg, dev, times.to(dev)); b = m._bind(); eng, flat = b.engine, b.flat
    obs_d = obs.to(dev); u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()}); eps_d = torch.randn(B, m.latent_dim, device=dev)
    svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-4))
    for _ in range(3):
        for _ in range(30): svi.step_async(obs_d, eps=eps_d, u=u_d)
        torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): svi.step_async(obs_d, eps=eps_d, u=u_d)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    eng.profile_enable(True); svi.step_async(obs_d, eps=eps_d, u=u_d); pr = eng.profile_read(); eng.profile_enable(False)
    print(json.dumps({"config": name, "params": int(flat.numel()), "us_per_step": round(1e6 * dt / n, 1), "traj_per_s": round(B * n / dt),
                      "kernel_us": {k: round(v, 1) for k, v in pr}}), flush=True)
