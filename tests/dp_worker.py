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


def run(rank, world, steps, B=64):
    from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
    dev = torch.device("cuda", 0)
    cfg, model, obs, u, eps = build(dev, B)
    b = model._bind()
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    svi = ELBOStep(b.engine, b.flat, FlatAdam(b.engine, b.flat, lr=cfg.learning_rate))
    assert svi.world == world
    losses = []
    for _ in range(steps):
        losses.append(float(svi.step(obs[sl], eps=eps[sl].contiguous(), u=u[sl].contiguous())))
    ev = svi.evaluate_loss(obs[sl], eps=eps[sl].contiguous(), u=u[sl].contiguous())
    return dict(losses=losses, eval_loss=ev, params=b.flat.detach().cpu().clone(), grads=svi.grads.detach().cpu().clone())


if __name__ == "__main__":
    rank, world, port, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    res = run(rank, world, steps)
    if rank == 0:
        torch.save(res, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
