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


def test_two_ranks_real_engine_match_single_process():
    steps, world = 3, 2
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dp.pt")
        # each child writes to its own file: a rank that floods a pipe nobody drains would stall, and its peer would then wait in the
        # collective until the timeout while holding the GPU
        log_paths = [os.path.join(tmp, "rank%d.log" % r) for r in range(world)]
        handles = [open(lp, "wb") for lp in log_paths]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), port, str(steps), out],
                                  env=env, stdout=handles[r], stderr=subprocess.STDOUT) for r in range(world)]
        try:
            for p in procs:
                p.wait(timeout=280)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                       # exactly the children this test started
            raise
        finally:
            for h in handles:
                h.close()
        logs = [open(lp, "rb").read().decode(errors="replace")[-2000:] for lp in log_paths]
        assert all(p.returncode == 0 for p in procs), logs
        dp = torch.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    single = dp_worker.run(0, 1, steps)                     # whole batch, one process: the fused elbo + Adam path
    for a, b in zip(dp["losses"], single["losses"]):
        assert abs(a - b) <= 1e-6 * abs(b), (dp["losses"], single["losses"])
    assert abs(dp["eval_loss"] - single["eval_loss"]) <= 1e-6 * abs(single["eval_loss"])
    assert (dp["params"] - single["params"]).abs().max().item() < 2e-6
    # the reduced gradient of the last step equals the whole-batch gradient
    g_dp, g_one = dp["grads"].double(), single["grads"].double()
    assert ((g_dp - g_one).norm() / g_one.norm()).item() < 1e-5
