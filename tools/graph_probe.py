"""Diagnostic (GPU box): would a hipGraph of K metric steps beat K x 4 stream launches?  Timing only -- the captured Adam launch
carries a frozen step count, so the replayed updates are not a training run; the kernels and their order are the shipped step's.
Also prints the host's enqueue cost per step (K enqueues timed without waiting for the GPU).
Usage: python tools/graph_probe.py"""
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
times = torch.arange(0.0, 200.0, device=dev)
m = MechanisticModel(cfg, dev, times); b = m._bind(); eng, flat = b.engine, b.flat
obs, labels, _ = synthetic_batch("cvs", 1024, 200, 3); obs_d = obs.to(dev)
u_d = m.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
eps_d = torch.randn(1024, 8, generator=torch.Generator().manual_seed(99)).to(dev)
svi = ELBOStep(eng, flat, FlatAdam(eng, flat, lr=cfg.learning_rate))
step = lambda: svi.step_async(obs_d, eps=eps_d, u=u_d)
sync = lambda: torch.cuda.synchronize(dev)


def blocks(fn, per_block, R=25):
    out = []
    for _ in range(R):
        sync(); t0 = time.perf_counter(); fn(); sync()
        out.append(1e6 * (time.perf_counter() - t0) / per_block)
    return statistics.median(out), min(out)


for _ in range(20):
    for _ in range(50): step()
    sync()
for K in (20, 200):
    med, lo = blocks(lambda: [step() for _ in range(K)], K)
    print("stream launches, blocks of %3d steps: median %.2f us/step  min %.2f" % (K, med, lo), flush=True)
# host enqueue cost: K enqueues behind a long queue (the GPU is busy throughout, nothing waits)
sync()
for _ in range(100): step()
t0 = time.perf_counter()
for _ in range(200): step()
host = 1e6 * (time.perf_counter() - t0) / 200
sync()
print("host enqueue cost (one ctypes call = 4 launches): %.2f us/step" % host, flush=True)

try:
    K = 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(3): step()      # the capture stream's first use of the library outside the capture
    torch.cuda.current_stream(dev).wait_stream(s)
    sync()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): step()
    sync()
    for _ in range(20): g.replay()
    sync()
    med, lo = blocks(lambda: g.replay(), K)
    print("hipGraph of %d steps, one replay per block:   median %.2f us/step  min %.2f" % (K, med, lo), flush=True)
    med, lo = blocks(lambda: [g.replay() for _ in range(10)], 10 * K)
    print("hipGraph of %d steps, ten replays per block:  median %.2f us/step  min %.2f" % (K, med, lo), flush=True)
    print("loss after the replays: %.4f (finite: %s)" % (float(svi.loss.item()) / 1024, bool(torch.isfinite(svi.loss).all())))
except Exception as e:   # a library call that cannot be captured: say which
    print("capture failed: %r" % (e,))


# ---- the reference's whole minibatch (main + auxiliary SVI step: nine launches) ----------------------------------------------------
def probe(name, fn, K, units):
    """stream launches against a graph of K calls of fn, in blocks of K calls and of 10 K calls; `units` = steps per call"""
    try:
        for _ in range(5):
            for _ in range(30): fn()
            sync()
        m1, _ = blocks(lambda: [fn() for _ in range(K)], K * units)
        m2, _ = blocks(lambda: [fn() for _ in range(10 * K)], 10 * K * units)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream(dev).wait_stream(s)
        sync()
        with torch.cuda.graph(g, stream=s):
            for _ in range(K): fn()
        sync()
        for _ in range(10): g.replay()
        sync()
        g1, _ = blocks(lambda: g.replay(), K * units)
        g2, _ = blocks(lambda: [g.replay() for _ in range(10)], 10 * K * units)
        print("%s: stream %.2f / %.2f us, graph %.2f / %.2f us (blocks of %d / %d)" % (name, m1, m2, g1, g2, K, 10 * K), flush=True)
    except Exception as e:
        print("%s: capture failed: %r" % (name, e), flush=True)


from structured_latent_odes_amd.svi import AuxStep
labels_d = {k: v.to(dev) for k, v in labels.items()}
aux = AuxStep(m, svi.optimizer)
probe("run_batch (main + auxiliary step, per minibatch)", lambda: (step(), aux.step_async(obs_d, eps=eps_d, **labels_d)), 20, 1)

# ---- config[2] with its own solver (seven launches, 0.30 ms) -------------------------------------------------------------------------
import importlib
from structured_latent_odes_amd import configs as CF
cfg2 = CF.load_config_proc(); cfg2.update(seq_len=100, solver="dopri5")
set_seed(12)
mod = importlib.import_module("structured_latent_odes_amd.models.mechanistic_proc")
obs2, labels2, times2 = synthetic_batch("proc", 4096, 100, cfg2.obs_dim)
m2 = mod.MechanisticModel(cfg2, dev, times2.to(dev)); b2 = m2._bind()
obs2_d = obs2.to(dev); u2_d = m2.labels_to_u(**{k: v.to(dev) for k, v in labels2.items()}); eps2_d = torch.randn(4096, m2.latent_dim, device=dev)
svi2 = ELBOStep(b2.engine, b2.flat, FlatAdam(b2.engine, b2.flat, lr=1e-4))
probe("config[2] dopri5 step", lambda: svi2.step_async(obs2_d, eps=eps2_d, u=u2_d), 20, 1)
