"""World-size-2 gloo test (CPU) of the data-parallel stepper logic in structured_latent_odes_amd/svi.py: shard the batch,
ONE SUM all-reduce over [flat gradient | loss], identical Adam on every rank  ==  the single-process step on the whole batch.

The HIP engine cannot run here, so the stepper is driven with a TEST DOUBLE of the engine (same method names) whose
arithmetic is the CPU oracle -- this exercises the collective / optimizer plumbing only, which is all that differs at N>1."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import slode_oracle as O


class OracleEngine:
    """Engine test double: elbo_step / adam_step computed on the CPU by the oracle / torch ops."""

    def __init__(self, ospec, T, times):
        self.ospec, self.T, self.times = ospec, T, times
        self.keys = None
        self.device = torch.device("cpu")
        p = O.init_params(ospec, T=T)
        self.keys = [(k, v.shape, v.numel()) for k, v in p.items()]
        self.n_params = sum(n for _, _, n in self.keys)
        self.spec = type("S", (), {"latent_dim": ospec.latent_dim})()

    def pack(self, p):
        return torch.cat([p[k].reshape(-1) for k, _, _ in self.keys]).clone()

    def unpack(self, flat):
        out, off = {}, 0
        for k, shp, n in self.keys:
            out[k] = flat[off:off + n].view(shp)
            off += n
        return out

    def elbo_step(self, params, obs, u, eps, loss_out, grads=None, x_out=None, z_out=None):
        p = self.unpack(params)
        if grads is None:
            with torch.no_grad():
                loss_out[0] = O.main_loss(p, self.ospec, obs, u, eps, self.times)
            return loss_out
        loss, g = O.loss_and_grads(p, self.ospec, obs, u, eps, self.times)
        loss_out[0] = loss
        grads[:self.n_params] = torch.cat([g[k].reshape(-1) for k, _, _ in self.keys])
        return loss_out

    # the one-call step interface of the real engine (Engine.make_batch / svi_step / rng_state): explicit eps only -- the double has no
    # in-kernel generator -- and the label tensors concatenated here, on the test side
    def rng_state(self):
        return 0, 0, 0

    def make_batch(self, obs, labels, eps=None):
        assert eps is not None, "the engine double needs explicit eps"
        return (obs, torch.cat([t.reshape(t.shape[0], -1) for t in labels], dim=1) if labels else None, eps)

    def svi_step(self, kind, params, batch, B, loss_out, grads=None, adam=None):
        assert kind == 0
        obs, u, eps = batch
        self.elbo_step(params, obs, u, eps, loss_out, grads)
        if adam is not None:
            m, v, lr, step, betas, aeps = adam
            full = torch.zeros_like(params)
            full[:self.n_params] = grads[:self.n_params]
            self.adam_step(params, full, m, v, lr, step, betas, aeps)
        return loss_out

    def adam_step(self, params, grads, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
        b1, b2 = betas
        m.lerp_(grads, 1 - b1)
        v.mul_(b2).addcmul_(grads, grads, value=1 - b2)
        denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
        params.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def _run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
        torch.set_num_threads(2)
        ospec = O.cvs_spec(1, 1, 2, solver="rk4")
        B, T = 8, 30
        obs, u, eps, times = O.synthetic_batch(ospec, B, T)
        eng = OracleEngine(ospec, T, times)
        flat = eng.pack(O.init_params(ospec, T=T))
        sl = slice(rank * B // world, (rank + 1) * B // world)
        svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-2))
        losses = [svi.step(obs[sl], eps=eps[sl], u=u[sl]) for _ in range(3)]
        ev = svi.evaluate_loss(obs[sl], eps=eps[sl], u=u[sl])
        q.put((rank, losses, ev, flat.clone()))
    finally:
        torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the whole batch
    from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
    ospec = O.cvs_spec(1, 1, 2, solver="rk4")
    obs, u, eps, times = O.synthetic_batch(ospec, 8, 30)
    eng = OracleEngine(ospec, 30, times)
    flat = eng.pack(O.init_params(ospec, T=30))
    svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=1e-2))
    want = [svi.step(obs, eps=eps, u=u) for _ in range(3)]
    want_ev = svi.evaluate_loss(obs, eps=eps, u=u)
    (r0, l0, e0, f0), (r1, l1, e1, f1) = res
    assert l0 == l1 and torch.equal(f0, f1)                       # every rank holds the same loss and the same weights
    assert e0 == e1
    for a, b in zip(l0, want):
        assert abs(a - b) / abs(b) < 1e-5                         # summed over shards == whole batch (fp32 reduction order differs)
    assert abs(e0 - want_ev) / abs(want_ev) < 1e-5
    assert ((f0 - flat).norm() / flat.norm()).item() < 1e-4
