"""Child process of tests/test_gpu_distributed.py: one data-parallel rank on cuda:0 (gloo rendezvous on 127.0.0.1) with the REAL engine.
argv: rank world port steps out_path"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(dev, B, T=200):
    """Model, binding and the WHOLE synthetic batch (every rank builds the same one and takes its shard)."""
    from structured_latent_odes_amd.configs import load_config_cvs
    from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
    from structured_latent_odes_amd.synthetic import synthetic_batch
    from structured_latent_odes_amd.utils.utils import set_seed
    cfg = load_config_cvs()
    cfg.update(seq_len=T, z_iext_dim=3, z_rtpr_dim=3, z_epsilon_dim=2, solver="rk4", mini_batch_size=B)
    set_seed(cfg.seed)
    times = torch.arange(0.0, T * cfg.delta_t, cfg.delta_t, device=dev)
    model = MechanisticModel(cfg, dev, times)
    obs, labels, _ = synthetic_batch("cvs", B, T, 3, seed=4321)
    u = model.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
    eps = torch.randn(B, model.latent_dim, generator=torch.Generator().manual_seed(7)).to(dev)
    return cfg, model, obs.to(dev), u, eps


def run(rank, world, steps, B=64, payload="G", unfused=False, draw=False):
    """payload: what the data-parallel step all-reduces ("G": [G | head products | ODE-half row], "grad": [flat gradient | loss]);
    unfused: take the data-parallel code path at world size 1 too; draw: noise drawn in the kernels (keyed by the GLOBAL trajectory index,
    so the sharded run and the whole-batch run see the same noise) instead of the explicit eps tensor."""
    from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
    dev = torch.device("cuda", 0)
    cfg, model, obs, u, eps = build(dev, B)
    b = model._bind()
    b.engine.rng_seed(2026)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    svi = ELBOStep(b.engine, b.flat, FlatAdam(b.engine, b.flat, lr=cfg.learning_rate))
    svi.dp_payload, svi.unfused = payload, unfused
    assert svi.world == world
    e = (lambda: None) if draw else (lambda: eps[sl].contiguous())
    losses = []
    for _ in range(steps):
        losses.append(float(svi.step(obs[sl], eps=e(), u=u[sl].contiguous())))
    ev = svi.evaluate_loss(obs[sl], eps=e(), u=u[sl].contiguous())
    res = dict(losses=losses, eval_loss=ev, params=b.flat.detach().cpu().clone(), grads=svi.grads.detach().cpu().clone(),
               collective_bytes=svi.collective_bytes)
    if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
        res["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    return res


if __name__ == "__main__":
    rank, world, port, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    payload = sys.argv[7] if len(sys.argv) > 7 else "G"
    draw = len(sys.argv) > 8 and sys.argv[8] == "draw"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))   # as bench.py does
    else:
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    res = run(rank, world, steps, payload=payload, unfused=(world == 1), draw=draw)
    if rank == 0:
        torch.save(res, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
