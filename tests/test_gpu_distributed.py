"""Data-parallel step with the REAL engine at world_size 2: two fresh child processes share cuda:0 and
rendezvous over gloo on 127.0.0.1 (the driver's multi-GPU runs use one rank per GPU over RCCL -- the collective call is the same
torch.distributed.all_reduce of the [flat gradient | loss] buffer).  Each rank steps its shard: unfused slode_elbo_step ->
SUM all-reduce -> slode_adam_step; the result must equal the single-process step on the whole batch (fused slode_elbo_adam_step)
up to fp32 summation order: losses 1e-6 relative, weights after 3 Adam steps 2e-6 absolute (Adam's normalised update is
lr * m / (sqrt(v) + eps) with lr = 1e-3 here)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _launch(world, steps, backend, payload, draw=False, timeout=280):
    """`world` fresh child processes on cuda:0 (tests/dp_worker.py); returns rank 0's result dict."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dp.pt")
        # each child writes to its own file: a rank that floods a pipe nobody drains would stall, and its peer would then wait in the
        # collective until the timeout while holding the GPU
        log_paths = [os.path.join(tmp, "rank%d.log" % r) for r in range(world)]
        handles = [open(lp, "wb") for lp in log_paths]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), port, str(steps), out,
                                   backend, payload] + (["draw"] if draw else []),
                                  env=env, stdout=handles[r], stderr=subprocess.STDOUT) for r in range(world)]
        try:
            for p in procs:
                p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                       # exactly the children this test started
            raise
        finally:
            for h in handles:
                h.close()
        logs = [open(lp, "rb").read().decode(errors="replace")[-2000:] for lp in log_paths]
        assert all(p.returncode == 0 for p in procs), logs
        return torch.load(out)


def _single(steps, draw=False):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    return dp_worker.run(0, 1, steps, draw=draw)            # whole batch, one process: the fused one-call step


def _same(dp, single):
    for a, b in zip(dp["losses"], single["losses"]):
        assert abs(a - b) <= 1e-6 * abs(b), (dp["losses"], single["losses"])
    assert abs(dp["eval_loss"] - single["eval_loss"]) <= 1e-6 * abs(single["eval_loss"])
    assert (dp["params"] - single["params"]).abs().max().item() < 2e-6
    # the reduced gradient of the last step equals the whole-batch gradient
    g_dp, g_one = dp["grads"].double(), single["grads"].double()
    assert ((g_dp - g_one).norm() / g_one.norm()).item() < 1e-5


@pytest.mark.parametrize("payload", ["G", "grad"])
def test_two_ranks_real_engine_match_single_process(payload):
    """payload "G": slode_grad_partial -> all-reduce of [G | head products | ODE-half row | loss] -> slode_grad_apply (chain rule + Adam once,
    on the reduced payload); "grad": slode_svi_step -> all-reduce of [flat gradient | loss] -> slode_adam_step."""
    steps = 3
    dp = _launch(2, steps, "gloo", payload)
    _same(dp, _single(steps))
    # per step and rank: [G 50 x 601 | 2 x (8 x 51) | loss + 1786 ODE-half floats] = 32,656 floats = 131 KB against 96,463 floats = 386 KB
    assert dp["collective_bytes"] == {"G": 4 * (30052 + 2 * 408 + 1788), "grad": 4 * 96463}[payload], dp["collective_bytes"]


def test_two_ranks_drawing_their_own_noise_match_single_process():
    """eps == NULL on every rank: the in-kernel generator is keyed by the GLOBAL trajectory index (rank r's shard starts at r * B / N), so
    the sharded job and the single-process job integrate the same latent samples."""
    steps = 3
    _same(_launch(2, steps, "gloo", "G", draw=True), _single(steps, draw=True))


@pytest.mark.parametrize("payload", ["G", "grad"])
def test_rccl_at_world_size_one_runs_the_data_parallel_path(payload):
    """RCCL itself (backend "nccl" IS RCCL on ROCm): a fresh process initialises it at world size 1 exactly as bench.py does at N > 1
    (device_id=...), runs the N > 1 code path -- gradient-only call, all_reduce on the RCCL communicator, Adam call -- and must land on
    the single-process fused step; torch.cuda.nccl.version() is recorded."""
    steps = 3
    dp = _launch(1, steps, "nccl", payload)
    _same(dp, _single(steps))
    assert dp.get("nccl_version"), dp.keys()
    print("RCCL version:", dp["nccl_version"])
