#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec of one ELBO step (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic CVS-shaped minibatch shard: encoder -> latent sample -> rk4
(3/8-rule) latent-ODE solve over T=200 -> 3 quantile heads -> asymmetric-Laplace likelihood + latent log-probs ->
-ELBO -> exact gradient of all 96,462 parameters (one slode_elbo_step call) -> [N>1: one RCCL SUM all-reduce of the flat
gradient + loss scalar] -> Adam.  Workload = BASELINE config[1] "Synthetic CVS batch=1024, T=200, latent_dim=8,
blackbox_ode RK4" per GPU; N GPUs run N such shards (config[3]: 8 x 1024 = 8192) => weak scaling; the N>1 line also carries
the strong-scaling figure (the same global batch on one GPU / the data-parallel step) and the collective's own time.
Inputs are generated on the host from a seed and are resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (a fresh
`python -m torch.distributed.run` child, before this process touches a GPU) and exits with the child's status; launched under
torch.distributed.run it is one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then the K-step block (barrier + synchronize on both sides, MAX over ranks) is repeated R >= 25 times and
`ms_per_step` is the MEDIAN block / K (K = 20 steps is 1.3 ms: one sample of that says little).  `roofline` is for the dominant
kernel of the step on ONE clock: the dispatch's own begin -> end device timestamps (hipExtLaunchKernelGGL start / stop events
through slode_profile_enable), the quantity rocprofv3 --kernel-trace reports (profiles/).  `other_configs` runs the remaining
BASELINE configs the same way; `run_batch` is the reference's whole minibatch (main + auxiliary SVI step, two Adam passes,
training_cvs.py:147-157).  `cpu_baseline` is the oracle (reference-equivalent eager-PyTorch CPU restatement,
oracle/slode_oracle.py) timed on this box's host cores on the same workload (rank 0, N=1 only).
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, T, Z_SPLIT = 1024, 200, (3, 3, 2)
PEAK_FP32 = 157.3                               # TFLOP/s, fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM = 8000.0                               # GB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=25, help="timed K-step blocks; ms_per_step is their median / K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-run-batch", action="store_true", help="skip the run_batch (main + auxiliary step) block: profiling runs of the metric step alone")
    ap.add_argument("--ab", action="store_true", help="A/B runs (tools/ab_env.sh, ab_libs.sh): the metric step and its kernel clocks only")
    ap.add_argument("--grad-mode", choices=["exact", "reference_adjoint"], default="exact",
                    help="exact: gradient of the discrete scheme (adjoint_solver=False); reference_adjoint: torchdiffeq.odeint_adjoint's")
    a = ap.parse_args()
    if a.ab:
        a.no_cpu_baseline = a.no_other_configs = a.no_run_batch = True
    return a


def self_launch(args):
    """--gpus N from a bare shell: start N ranks as a FRESH child process (this parent has not touched a GPU and never will)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ---- algorithmic FLOPs per trajectory (hidden z-part hoisted, every distinct stage time evaluated once; fwd + bwd = 3 x fwd)
def flops_fwd(shape):
    """shape: dict(T, C, L, S, Q) with the reference's H=25, F=10, K=10, P=5, Hc=50; the fixed-grid count with rk4's 3(T-1)+1 stage
    times -- config[2]'s adaptive solver is priced at this rk4-equivalent count too (column C2)."""
    T_, C, L, S, Q = shape["T"], shape["C"], shape["L"], shape["S"], shape["Q"]
    H, F, K, P, Hc = 25, 10, 10, 5, 50
    n_conv, n_pool = T_ - K + 1, T_ - K + 1 - P + 1
    R = {"euler": 1, "midpoint": 2}.get(shape.get("solver", "rk4"), 3)
    n_t = R * (T_ - 1) + 1
    return {"enc": 2 * F * C * K * n_conv + P * F * n_pool + 2 * F * n_pool * Hc + 4 * Hc * L,
            "ode": n_t * (2 * H + 4 * H * S + 2 * S) + 2 * L * H, "scan": 8 * S * (T_ - 1), "heads": 2 * Q * S * C * T_, "ll": 10 * Q * C * T_}


def kernel_flops(shape):
    """fwd+bwd algorithmic FLOPs per trajectory attributed to each kernel of the step (the kernels execute far fewer: DESIGN 3.1)."""
    f = flops_fwd(shape)
    gemm = 2 * 10 * (shape["T"] - 13) * 50
    solver, score = f["ode"] + f["scan"], f["heads"] + f["ll"]
    return {"weff": 0, "enc_fwd2": f["enc"], "enc_fwd": f["enc"], "ode_elbo": 3 * (solver + score) if shape.get("solver") != "dopri5" else 3 * score,
            "dopri5_fwd": solver, "dopri5_bwd": 2 * solver, "enc_bwd2": 3200, "enc_bwd": 2 * f["enc"] - gemm, "enc_bwd_lin": gemm,
            "enc_chain": 2 * f["enc"] - gemm - 3200, "slab_stage1": 0, "reduce": 0, "adam": 0, "aux": 0}


def build_case(fam, gauss, B, T_, kw, dev, seed):
    import torch
    from structured_latent_odes_amd import configs as CF
    from structured_latent_odes_amd.synthetic import synthetic_batch
    from structured_latent_odes_amd.utils.utils import set_seed
    cfg = getattr(CF, "load_config_" + fam)()
    cfg.update(seq_len=T_, mini_batch_size=B, **kw)
    set_seed(cfg.seed)                                              # identical weights on every rank (config_cvs.py:28)
    mod = importlib.import_module("structured_latent_odes_amd.models.mechanistic_%s%s" % (fam, "_Gauss" if gauss else ""))
    cls = getattr(mod, "MechanisticModelGauss" if gauss else "MechanisticModel")
    obs, labels, times = synthetic_batch(fam, B, T_, cfg.obs_dim, seed=seed)
    model = cls(cfg, dev, times.to(dev))                            # reference initialisers, random init
    obs_d = obs.to(dev)                                             # cvs / challenge: [B,C,T] view of a contiguous [B,T,C] tensor (native layout)
    u_d = model.labels_to_u(**{k: v.to(dev) for k, v in labels.items()})
    eps_d = torch.randn(B, model.latent_dim, generator=torch.Generator().manual_seed(99 + seed)).to(dev)
    return cfg, model, obs, obs_d, u_d, eps_d, times, {k: v.to(dev) for k, v in labels.items()}


def timed_blocks(step, sync, K, R, world, dev):
    """R blocks of exactly K steps, each bracketed by barrier + synchronize; per block the MAX over ranks.  Returns seconds per block."""
    import torch
    out = []
    for _ in range(R):
        sync()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        out.append(dt)
    return out


def prewarm(step, sync, blocks=80, n=50):
    """Engine / device pre-warm (untimed): a fresh box's first process sees one-off stalls of tens of ms (lazy HIP runtime pool growth,
    clocks ramping).  n-step blocks until three consecutive ones agree within 3 %."""
    prev, agree = None, 0
    for _ in range(blocks):
        sync()
        b0 = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        cur = time.perf_counter() - b0
        agree = agree + 1 if (prev is not None and abs(cur - prev) <= 0.03 * prev) else 0
        if agree >= 2:
            break
        prev = cur


def kernel_clocks(eng, calls, n=40):
    """Average own duration (us) of every kernel of one step: `calls` is a list of callables, each ONE profiled entry point."""
    eng.profile_enable(True)
    acc, order = {}, []
    for it in range(n + 5):
        for call in calls:
            call()
            for name, us in eng.profile_read():
                if it < 5:
                    continue                      # (the first launches through the event-carrying launch path are not steady state)
                if name not in acc:
                    acc[name] = []
                    order.append(name)
                acc[name].append(us)
    eng.profile_enable(False)
    # median: a launch that shares the chip with the host's own traffic (the read-back of the previous launch's events) is an outlier
    return {k: statistics.median(acc[k]) for k in order}


def roofline_of(kern_us, shape, B, variants=None):
    """Dominant kernel, its algorithmic FLOPs per launch (SURVEY 8d's per-trajectory count x B) and TFLOP/s on its own clock.  Where the
    ODE kernel also runs the encoder forward (ENCF: no encoder launch in the step) the reference's encoder-forward count is part of what
    the launch computes and is credited -- `variants` (a dict, filled in place) then carries the same clock under the two narrower
    counts as well, so the attribution is visible: the solver / ELBO phases alone, and those plus the folded mat-vec the kernel actually
    executes for the encoder (2 Hc C T instead of the reference's conv + pool + flatten-linear)."""
    kf = kernel_flops(shape)
    dom = max(kern_us, key=kern_us.get)
    fl = kf.get(dom, 0) * B
    fused_enc = dom == "ode_elbo" and "enc_fwd2" not in kern_us and "enc_fwd" not in kern_us
    if variants is not None and dom == "ode_elbo":
        sec = kern_us[dom] * 1e-6
        variants["solver_and_elbo_phases_only"] = {"flops_per_launch": fl, "frac": fl / sec / 1e12 / PEAK_FP32}
        if fused_enc:
            ex = fl + 2 * 50 * shape["C"] * shape["T"] * B
            variants["with_encoder_as_executed_folded_matvec"] = {"flops_per_launch": ex, "frac": ex / sec / 1e12 / PEAK_FP32}
            ref = fl + kf["enc_fwd2"] * B
            variants["with_encoder_at_reference_count"] = {"flops_per_launch": ref, "frac": ref / sec / 1e12 / PEAK_FP32}
    if fused_enc:
        fl += kf["enc_fwd2"] * B   # the step has no encoder launch: the ODE kernel ran the encoder forward of its trajectories itself (ENCF)
    achieved = fl / (kern_us[dom] * 1e-6) / 1e12
    return dom, fl, achieved


def other_config_line(name, fam, gauss, B, T_, kw, shape, dev, grad_mode_ra=True):
    import torch
    from structured_latent_odes_amd.svi import ELBOStep, FlatAdam
    kw = dict(kw, adjoint_solver=grad_mode_ra)
    cfg, model, _, obs_d, u_d, eps_d, _, _ = build_case(fam, gauss, B, T_, kw, dev, seed=1234)
    b = model._bind()
    svi = ELBOStep(b.engine, b.flat, FlatAdam(b.engine, b.flat, lr=cfg.learning_rate))
    step = lambda: svi.step_async(obs_d, eps=eps_d, u=u_d)
    sync = lambda: torch.cuda.synchronize(dev)
    prewarm(step, sync, blocks=12, n=30)
    K = 50
    blocks = timed_blocks(step, sync, K, 9, 1, dev)
    ms = 1e3 * statistics.median(blocks) / K
    kern = kernel_clocks(b.engine, [step], n=20)
    dom, fl, ach = roofline_of(kern, shape, B)
    step_fl = 3 * sum(flops_fwd(shape).values()) * B
    loss = float(svi.loss.item()) / B
    assert loss == loss, "NaN loss in " + name
    return {"config": name, "params": int(b.flat.numel()), "ms_per_step": ms, "traj_per_s": B / (ms * 1e-3), "kernel_us": kern,
            "dominant_kernel": dom, "dominant_kernel_us": kern[dom], "frac": ach / PEAK_FP32, "achieved_tflops": ach,
            "step_frac_fp32": step_fl / (ms * 1e-3) / 1e12 / PEAK_FP32, "loss_per_traj": loss,
            "grad_mode": "reference_adjoint" if grad_mode_ra else "exact"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    # SLODE_BENCH_REHEARSE=1: control-flow rehearsal of the N>1 path on a one-GPU box (every rank on cuda:0, gloo); never a measurement
    rehearse = world > 1 and os.environ.get("SLODE_BENCH_REHEARSE") == "1"
    if world > 1 and not rehearse and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d HIP devices are visible (SLODE_BENCH_REHEARSE=1 rehearses the control flow on one GPU over gloo)"
                         % (world, torch.cuda.device_count()))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)

    from structured_latent_odes_amd.svi import AuxStep, ELBOStep, FlatAdam

    cvs_kw = dict(z_iext_dim=Z_SPLIT[0], z_rtpr_dim=Z_SPLIT[1], z_epsilon_dim=Z_SPLIT[2], solver="rk4",
                  adjoint_solver=(args.grad_mode == "reference_adjoint"))
    cfg, model, obs, obs_d, u_d, eps_d, times, labels_d = build_case("cvs", False, B_PER_GPU, T, cvs_kw, dev, seed=1234 + rank)   # this rank's shard
    assert obs_d.stride() == (T * 3, 1, 3), obs_d.stride()
    binding = model._bind()
    eng, flat = binding.engine, binding.flat
    opt = FlatAdam(eng, flat, lr=cfg.learning_rate)
    svi = ELBOStep(eng, flat, opt)
    shape1 = dict(T=T, C=3, L=sum(Z_SPLIT), S=5, Q=3, solver="rk4")
    # snapshot for the CPU baseline (the timed loop below updates `flat` in place)
    cpu_params = {k: v.detach().cpu().clone() for k, v in eng.unpack(flat[:eng.n_params]).items()} if rank == 0 and world == 1 else None

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    step = lambda: svi.step_async(obs_d, eps=eps_d, u=u_d)
    prewarm(step, sync)
    for _ in range(args.warmup):
        step()
    R = max(1, args.repeats)
    blocks = timed_blocks(step, sync, args.steps, R, world, dev)
    dt = statistics.median(blocks)
    final_loss = float(svi.loss.item())
    ms_per_step = 1e3 * dt / args.steps
    value = world * B_PER_GPU * args.steps / dt

    # ---- per-kernel durations: every dispatch's own begin -> end timestamps, in an instrumented pass over the same steps ----
    if world == 1:
        calls = [step]
    else:   # N > 1: the step is two entry points around the collective -- slode_grad_partial | all-reduce of the payload | slode_grad_apply
        from structured_latent_odes_amd import _lib as SL
        bt = eng.make_batch(obs_d, [u_d], eps_d)
        if isinstance(svi._payload, torch.Tensor):
            pay = svi._payload
            calls = [lambda: eng.grad_partial(SL.SVI_MAIN, flat, bt, B_PER_GPU, pay),
                     lambda: eng.grad_apply(SL.SVI_MAIN, flat, bt, B_PER_GPU, pay, svi.loss, svi.grads, adam=(opt.exp_avg, opt.exp_avg_sq, 0.0, 1, opt.betas, opt.eps))]
        else:
            calls = [lambda: eng.elbo_step(flat, obs_d, u_d, eps_d, svi.loss, svi.grads), lambda: opt.step(svi.gbuf[:flat.numel()])]
    kern_us = kernel_clocks(eng, calls, n=40)
    frac_variants = {}
    dom, flops_launch, achieved = roofline_of(kern_us, shape1, B_PER_GPU, frac_variants)
    step_flops = 3 * sum(flops_fwd(shape1).values()) * B_PER_GPU
    bytes_per_traj = 4 * (3 * T + sum(Z_SPLIT) + 2)     # algorithmic HBM bytes: obs once + eps + labels = 2,440 B

    # PMC figures (HBM traffic, instruction counts) come from profiles/: only attached when they were measured on THIS kernel source
    import glob
    import hashlib
    sha = hashlib.sha1(open(os.path.join(ROOT, "structured_latent_odes_amd", "csrc", "ode_kernel.hip"), "rb").read()).hexdigest()
    traffic, issue, rocprof_us, traffic_all = None, None, None, {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("source_sha1_ode_kernel_hip") == sha and dom in d:
                traffic = d[dom].get("hbm_bytes_per_launch")
                traffic_all = {k: v.get("hbm_bytes_per_launch") for k, v in d.items() if isinstance(v, dict)}
                break
        except Exception:
            pass
    # the launches beside the dominant kernel: each with its algorithmic FLOP count (kernel_flops: SURVEY 8d split over the launches),
    # its own clock, and -- when the PMC pass was taken on this source -- its measured HBM bytes.  All of them sit far below both
    # roofs: they are launch- and latency-bound, which is what the two fractions say.
    pmc_name = {"weff": "fold", "enc_fwd2": "enc_fwd", "enc_bwd_lin": "gemm", "enc_chain": "chain"}
    kf1 = kernel_flops(shape1)
    side = []
    for kn, us in kern_us.items():
        if kn == dom:
            continue
        fl = kf1.get(kn, 0) * B_PER_GPU + (2 * 10 * 3 * 14 * 50 * T if kn == "weff" else 0)   # (the fold itself: Hc*C*T*F*(K+P-1) MACs per step)
        hb = traffic_all.get(pmc_name.get(kn, kn))
        side.append({"kernel": kn, "us": us, "algorithmic_flops_per_launch": fl, "tflops": fl / (us * 1e-6) / 1e12,
                     "frac_fp32": fl / (us * 1e-6) / 1e12 / PEAK_FP32, "hbm_bytes_per_launch": hb,
                     "hbm_gb_s": (hb / (us * 1e-6) / 1e9) if hb else None, "frac_hbm": (hb / (us * 1e-6) / 1e9 / PEAK_HBM) if hb else None})
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_ode_elbo_ab.json")), reverse=True):
        try:
            d = json.load(open(path))
            sq = d.get("arms", {}).get("alg0", {}).get("pmc_sq")
            if d.get("source_sha1_ode_kernel_hip") == sha and dom == "ode_elbo":
                rocprof_us = d.get("arms", {}).get("alg0", {}).get("ode_elbo_avg_us")
                if sq:
                    # issue-rate roofline: vector instructions of one launch x the measured issue cost of a wave-instruction at >= 4 waves
                    # per SIMD (tools/ubench/valu_rate.hip: 2.6 cycles) / 1024 SIMDs / 2.4 GHz = the time the vector pipes alone need
                    floor_us = sq["SQ_INSTS_VALU"] * 2.6 / 1024 / 2.4e9 * 1e6
                    issue = {"valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"], "cycles_per_inst": 2.6, "simds": 1024, "clock_ghz": 2.4,
                             "floor_us": floor_us, "frac_of_floor": floor_us / kern_us[dom],
                             "flop_per_lane_inst": flops_launch / (64.0 * sq["SQ_INSTS_VALU"]), "source": os.path.basename(path)}
                break
        except Exception:
            pass

    out = {
        "metric": "trajectories/sec ELBO step (CVS, batch=1024, T=200)", "value": value, "unit": "trajectories/s",
        "n_gpus": world, "nranks": world,
        "backend": ("gloo (rehearsal)" if rehearse else "nccl (RCCL over xGMI)") if world > 1 else "none (single process)",
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "timing": {"repeats": R, "statistic": "median of R blocks of K steps, each block barrier+synchronize bracketed, max over ranks",
                   "ms_per_step_min": 1e3 * min(blocks) / args.steps, "ms_per_step_max": 1e3 * max(blocks) / args.steps,
                   "ms_per_step_first_block": 1e3 * blocks[0] / args.steps},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        **({"rehearsal": "all ranks on cuda:0 over gloo: NOT a measurement"} if rehearse else {}),
        "config": {"workload": "BASELINE config[1]: synthetic CVS, B=1024/GPU, T=200, C=3, latent_dim=8 (3,3,2), S=5, "
                               "rk4(3/8) fixed grid dt=1, ALD 3-quantile likelihood; step = ELBO fwd+bwd (all 96,462 parameters) "
                               "+ grad all-reduce (N>1) + Adam",
                   "global_batch": world * B_PER_GPU, "T": T, "parallelism": "dp%d" % world, "grad_mode": args.grad_mode},
        "final_loss_per_traj": final_loss / (world * B_PER_GPU),
        "roofline": {"bound": "fp32_valu", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32, "traffic": traffic,
                     "frac_by_flop_count": frac_variants,
                     "frac_note": "frac = SURVEY 8d's algorithmic count of everything the launch computes (the fused kernel also runs the "
                                  "encoder forward, credited at the reference's operation count); frac_by_flop_count gives the same clock under "
                                  "the narrower counts; step_frac_fp32 (whole step, fixed count) is the figure to compare across rounds",
                     "pipe": "fp32 VALU (the dominant kernel issues no MFMA; on gfx950 the f32 MFMA peak equals the f32 vector peak, 157.3 TF)",
                     "clock": "the dispatch's own begin->end device timestamps (hipExtLaunchKernelGGL start/stop events), median of 40 "
                              "launches in this run: the quantity rocprofv3 --kernel-trace reports",
                     "kernel_avg_us": kern_us[dom], "rocprof_kernel_avg_us": rocprof_us,
                     "frac_on_rocprof_clock": (flops_launch / (rocprof_us * 1e-6) / 1e12 / PEAK_FP32) if rocprof_us else None,
                     "issue_roofline": issue, "algorithmic_flops_per_launch": flops_launch, "kernel_us": kern_us,
                     "kernels_sum_us": sum(kern_us.values()), "tail_us": sum(v for k, v in kern_us.items() if k != dom),
                     "side_launches": side,
                     "step_frac_fp32": step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32,
                     "step_frac_hbm": (bytes_per_traj * B_PER_GPU / (ms_per_step * 1e-3)) / 1e9 / PEAK_HBM,
                     "note": "intensity ~850 FLOP/B => compute side of the ridge; fp32 vector peak == fp32 MFMA peak (157.3 TF)"},
    }

    # ---- one continuous leg of the metric step, >= 2 s of back-to-back enqueues with ONE synchronize at the end: long enough for an
    # outside sampler (the driver's gpu_busy) to see the steady state the K-step blocks above measure
    if world == 1 and not args.ab:
        n_sus = int(min(400000, max(1000, 2.6 / (ms_per_step * 1e-3))))
        sync()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        sync()
        sus = time.perf_counter() - t0
        out["sustained"] = {"seconds": sus, "steps": n_sus, "traj_per_s": B_PER_GPU * n_sus / sus, "ms_per_step": 1e3 * sus / n_sus}

    # ---- the same step with the reference's DEFAULT gradients (adjoint_solver=True: torchdiffeq.odeint_adjoint's) ----------------------
    if world == 1 and not args.ab:
        other_mode = "exact" if args.grad_mode == "reference_adjoint" else "reference_adjoint"
        kw2 = dict(cvs_kw, adjoint_solver=(other_mode == "reference_adjoint"))
        cfg2, model2, _, obs2, u2, eps2, _, _ = build_case("cvs", False, B_PER_GPU, T, kw2, dev, seed=1234 + rank)
        b2 = model2._bind()
        svi2 = ELBOStep(b2.engine, b2.flat, FlatAdam(b2.engine, b2.flat, lr=cfg2.learning_rate))
        step2 = lambda: svi2.step_async(obs2, eps=eps2, u=u2)
        prewarm(step2, sync, blocks=12, n=50)
        ms2 = 1e3 * statistics.median(timed_blocks(step2, sync, args.steps, 9, 1, dev)) / args.steps
        out["value_" + other_mode] = B_PER_GPU / (ms2 * 1e-3)
        out["ms_per_step_" + other_mode] = ms2
        out["value_" + args.grad_mode] = value

    # ---- the reference's whole minibatch (training_cvs.py:147-157): main SVI step, auxiliary SVI step, Adam after each ----------------
    if world == 1 and not args.no_run_batch:
        aux = AuxStep(model, opt)
        aux_step = lambda: aux.step_async(obs_d, eps=eps_d, **labels_d)
        both = lambda: (step(), aux_step())
        prewarm(aux_step, sync, blocks=12, n=30)
        K = max(args.steps, 50)
        t_main = 1e3 * statistics.median(timed_blocks(step, sync, K, 9, 1, dev)) / K
        t_aux = 1e3 * statistics.median(timed_blocks(aux_step, sync, K, 9, 1, dev)) / K
        t_both = 1e3 * statistics.median(timed_blocks(both, sync, K, 9, 1, dev)) / K
        g0 = torch.zeros_like(flat)
        adam_only = lambda: eng.adam_step(flat, g0, opt.exp_avg, opt.exp_avg_sq, 0.0, 1)      # lr = 0: the weights stay put
        t_adam = 1e3 * statistics.median(timed_blocks(adam_only, sync, K, 9, 1, dev)) / K
        nograd = torch.zeros(flat.numel() + 1, device=dev)
        main_nofuse = lambda: eng.elbo_step(flat, obs_d, u_d, eps_d, nograd[-1:], nograd[:eng.n_params])
        t_main_noadam = 1e3 * statistics.median(timed_blocks(main_nofuse, sync, K, 9, 1, dev)) / K
        aux_kern = kernel_clocks(eng, [aux_step], n=20)
        # ... and through the reference's own call: run_batch(batch, losses) (training_cvs.py:147-157) on a device-resident batch dict with
        # the two SVI objects the entry point builds -- labels handed over as the loader yields them, noise drawn in the kernels, one
        # C-ABI call per SVI.step and the .item() the API demands (a host synchronisation per step, twice per minibatch)
        from structured_latent_odes_amd.svi import SVI, Adam, Trace_ELBO
        from structured_latent_odes_amd.training import run_batch
        popt = Adam({"lr": cfg.learning_rate, "betas": (0.9, 0.999)})
        popt._flat = opt                                      # (the FlatAdam the timed steps above already use: same state, same step counts)
        losses = [SVI(model.model, model.guide, popt, loss=Trace_ELBO(num_particles=1)),
                  SVI(model.model_meta, model.guide_meta, popt, loss=Trace_ELBO(num_particles=1))]
        batch = dict(observations=obs_d, **labels_d)
        api = lambda: run_batch(batch, losses)
        api_main = lambda: losses[0].step(**batch)
        for _ in range(50):
            api()
        t_api = 1e3 * statistics.median(timed_blocks(api, sync, K, 9, 1, dev)) / K
        t_api_main = 1e3 * statistics.median(timed_blocks(api_main, sync, K, 9, 1, dev)) / K
        # the same two steps enqueued without the per-step .item(): what the API's synchronisation costs
        api_async = lambda: (losses[0].step_async(**batch), losses[1].step_async(**batch))
        t_api_async = 1e3 * statistics.median(timed_blocks(api_async, sync, K, 9, 1, dev)) / K
        out["run_batch"] = {"main_ms": t_main, "aux_ms": t_aux, "adam_ms": t_adam, "total_ms": t_both,
                            "api_ms": t_api, "api_main_step_ms": t_api_main, "api_without_item_sync_ms": t_api_async,
                            "traj_per_s_api_run_batch": B_PER_GPU / (t_api * 1e-3),
                            "api_note": "api_ms: training.run_batch(batch, losses) = SVI.step(**batch) twice, each ONE slode_svi_step call (label "
                                        "tensors as the loader yields them, noise drawn in-kernel) + the .item() the reference API returns; "
                                        "api_without_item_sync_ms: the same calls through step_async (no host sync); total_ms: pre-made eps / u",
                            "main_without_adam_ms": t_main_noadam, "traj_per_s_full_run_batch": B_PER_GPU / (t_both * 1e-3),
                            "aux_kernel_us": aux_kern,
                            "note": "main / aux: one SVI step each with Adam applied by the last kernel of the step (what training runs); "
                                    "adam_ms: a stand-alone slode_adam_step over all parameters (the pass the fusion removes, twice per "
                                    "minibatch); total_ms: main + aux back to back = one reference run_batch"}

    # ---- N > 1: the collective alone, and strong scaling of the same global batch --------------------------------------------------------
    if world > 1:
        n_rep = 200
        def coll_time(buf):
            scratch = buf.clone()            # (the step's own buffer keeps its contents)
            for _ in range(20):
                torch.distributed.all_reduce(scratch, op=torch.distributed.ReduceOp.SUM)
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_rep):
                torch.distributed.all_reduce(scratch, op=torch.distributed.ReduceOp.SUM)
            e1.record()
            e1.synchronize()
            cmax = torch.tensor([1e3 * e0.elapsed_time(e1) / n_rep], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(cmax, op=torch.distributed.ReduceOp.MAX)
            return float(cmax.item())
        used = svi._payload if isinstance(svi._payload, torch.Tensor) else svi.gbuf
        out["collective_us"] = coll_time(used)
        out["collective"] = {"bytes": int(used.numel() * 4), "repeats": n_rep,
                             "op": "SUM all-reduce of [G = g_pre^T [X|1] | head-layer products | loss | ODE-half gradient row] (slode_grad_partial -> "
                                   "all-reduce -> slode_grad_apply: chain rule + Adam once, on the reduced payload)"
                                   if used is not svi.gbuf else "SUM all-reduce of [flat gradient | loss]",
                             "flat_gradient_bytes": int(svi.gbuf.numel() * 4), "flat_gradient_collective_us": coll_time(svi.gbuf),
                             "clock": "HIP events on the step's stream around %d back-to-back all-reduces, max over ranks" % n_rep}
        try:
            out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if not rehearse else None
        except Exception:
            out["nccl_version"] = None
        out["compute_us"] = sum(kern_us.values())
        sync()
        if rank == 0:   # the same GLOBAL batch on this one GPU (single-process fused step), outside the data-parallel region
            cfg1, model1, _, obs1, u1, eps1, _, _ = build_case("cvs", False, world * B_PER_GPU, T, cvs_kw, dev, seed=4321)
            b1 = model1._bind()
            one = ELBOStep(b1.engine, b1.flat, FlatAdam(b1.engine, b1.flat, lr=cfg1.learning_rate))
            one.world = 1
            step1 = lambda: one.step_async(obs1, eps=eps1, u=u1)
            lsync = lambda: torch.cuda.synchronize(dev)
            prewarm(step1, lsync, blocks=12, n=30)
            K1 = max(20, min(args.steps, 200))
            ms_one = 1e3 * statistics.median(timed_blocks(step1, lsync, K1, 9, 1, dev)) / K1
            out["strong_scaling"] = {"global_batch": world * B_PER_GPU, "one_gpu_ms_per_step": ms_one, "n_gpu_ms_per_step": ms_per_step,
                                     "strong_speedup": ms_one / ms_per_step,
                                     "note": "one GPU stepping the whole global batch (single-process fused step) / the N-GPU data-parallel step"}
        torch.distributed.barrier()

    # ---- the other BASELINE configs, driver-timed (parity-test shapes; reference_adjoint gradients as the configs' adjoint_solver=True asks) --
    if rank == 0 and world == 1 and not args.no_other_configs:
        others = []
        for name, fam, gauss, B, T_, kw, shape in [
                ("config[0]: cvs B=32 T=100 L=4 rk4", "cvs", False, 32, 100, dict(z_iext_dim=1, z_rtpr_dim=1, z_epsilon_dim=2, solver="rk4"),
                 dict(T=100, C=3, L=4, S=5, Q=3, solver="rk4")),
                ("config[2]: proc B=4096 T=100 L=50 S=8, rk4 (fixed-grid form of the same shapes)", "proc", False, 4096, 100, dict(solver="rk4"),
                 dict(T=100, C=4, L=50, S=8, Q=3, solver="rk4")),
                ("config[2]: proc B=4096 T=100 L=50 S=8, dopri5 (rtol 1e-7, atol 1e-9: torchdiffeq defaults)", "proc", False, 4096, 100,
                 dict(solver="dopri5"), dict(T=100, C=4, L=50, S=8, Q=3, solver="dopri5")),
                ("config[4] shard: challenge-Gauss B=512 T=300 L=15 rk4", "challenge", True, 512, 300, dict(solver="rk4"),
                 dict(T=300, C=4, L=15, S=5, Q=1, solver="rk4"))]:
            try:
                others.append(other_config_line(name, fam, gauss, B, T_, kw, shape, dev))
            except Exception as exc:   # a broken side config must not cost the headline line
                others.append({"config": name, "error": repr(exc)[:300]})
        out["other_configs"] = others

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import slode_oracle as O        # the ONLY use of the oracle here: the reported CPU baseline
        ospec = O.cvs_spec(*Z_SPLIT, solver="rk4")
        c_obs, c_u, c_eps, c_t = obs.cpu(), u_d.cpu(), eps_d.cpu(), times.cpu()
        # eager PyTorch with one thread per core thrashes on these tiny ops; pick the fastest of a few thread counts
        best_n, best_t = torch.get_num_threads(), float("inf")
        for n in sorted({8, 16, 32, torch.get_num_threads()}):
            if n > (os.cpu_count() or n):
                continue
            torch.set_num_threads(n)
            O.loss_and_grads(cpu_params, ospec, c_obs, c_u, c_eps, c_t)
            c0 = time.perf_counter()
            O.loss_and_grads(cpu_params, ospec, c_obs, c_u, c_eps, c_t)
            if time.perf_counter() - c0 < best_t:
                best_n, best_t = n, time.perf_counter() - c0
        torch.set_num_threads(best_n)
        nthreads = best_n
        n_cpu, c0 = 0, time.perf_counter()
        while n_cpu < 4 or (time.perf_counter() - c0 < 12.0 and n_cpu < 64):
            O.loss_and_grads(cpu_params, ospec, c_obs, c_u, c_eps, c_t)
            n_cpu += 1
        cdt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": B_PER_GPU * n_cpu / cdt, "unit": "trajectories/s", "cores": nthreads, "kind": "port",
                               "sample": "same workload (B=1024, T=200, rk4), %d ELBO fwd+bwd steps (no Adam) after 2 warm-ups, "
                                         "oracle/slode_oracle.py eager PyTorch fp32, torch threads=%d (fastest of 8/16/32/all)" % (n_cpu, nthreads)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
