"""Diagnostic: per-phase timeline of workgroup 0 of the three big kernels (needs `make -C structured_latent_odes_amd/csrc stamps`).
Usage (GPU box): SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so python tools/stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slode_oracle as O
from structured_latent_odes_amd import _lib, engine as E

dev = torch.device("cuda:0")
CFG = os.environ.get("STAMPS_CFG", "c1")   # c1: the metric shape; c2 / c2dp5: BASELINE config[2] with rk4 / dopri5
if CFG == "c1":
    ospec, espec, T, S = O.cvs_spec(3, 3, 2, solver="rk4"), E.cvs_spec(3, 3, 2, solver="rk4"), 200, 5
    B = int(os.environ.get("STAMPS_B", "1024"))
else:
    sol = "dopri5" if CFG == "c2dp5" else "rk4"
    ospec, espec, T, S = O.proc_spec(z_g=10, z_eps=10, solver=sol), E.proc_spec(z_g=10, z_eps=10, solver=sol), 100, 8
    B = int(os.environ.get("STAMPS_B", "4096"))
p = O.init_params(ospec, T=T, S=S)
obs, u, eps, times = O.synthetic_batch(ospec, B, T)
eng = E.Engine(espec, T, dev)
eng.set_times(times)
flat = eng.pack(p)
obs_d = obs.contiguous().to(dev) if CFG != "c1" else obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)
loss, grads = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
m_, v_ = torch.zeros_like(flat), torch.zeros_like(flat)
u_d, eps_d = u.to(dev), eps.to(dev)
for it in range(5):   # the bench's step: fused Adam
    eng.elbo_adam_step(flat, obs_d, u_d, eps_d, loss, grads, m_, v_, 1e-3, it + 1)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
assert lib.slode_debug_stamps_ode(buf) == 0
v = list(buf)
late = None
if hasattr(lib, "slode_debug_stamps_ode_late") and B > 1000:
    buf2 = (C.c_ulonglong * 32)()
    if lib.slode_debug_stamps_ode_late(buf2) == 0:
        late = list(buf2)
print("== ode kernel, workgroup 0 (us)   [second column: workgroup 1000, last residency slot of its CU]")
tot = 0.0
for lab, i, j in (("  setup detail: start -> set-up loads requested", 0, 23), ("  setup detail: -> encoder products done (loads returned)", 23, 24),
                  ("  setup detail: -> wave sums + tanh stored", 24, 25), ("  setup detail: -> LDS-DMA drained (vmcnt 0)", 25, 13),
                  ("setup: table loads -> LDS (+ encoder forward when fused)", 0, 13), ("setup: zero acc + barrier", 13, 14), ("setup: w_t, table check", 14, 1),
                  ("P0a latent sample / log-probs", 1, 15), ("P0b u, init hidden, switching indices, rank", 15, 16),
                  ("P0c x0 || piecewise-linear table", 16, 2), ("P1 stage evaluations + exchange", 2, 3), ("P1 step coefficients", 3, 4),
                  ("P2 forward scan", 4, 5), ("P3 heads + likelihood", 5, 6), ("P4 adjoint scan | head grads", 6, 7),
                  ("P5 step reverse mode + sample rows", 7, 8), ("P6 chunk sums + switching-index sums", 8, 9), ("P6 into the segment", 9, 17),
                  ("P7 aux, gp0", 17, 18), ("P7 latent gradient | owner accumulation", 18, 20), ("P7 last barrier", 20, 21),
                  ("g_pre block, prefetch", 21, 10), ("epilogue: loss", 10, 22), ("epilogue: slab write", 22, 11)):
    if v[i] and v[j]:
        extra = "  %8.2f" % ((late[j] - late[i]) / 100.0) if late and late[i] and late[j] else ""
        print("  %-58s %8.2f%s" % (lab, (v[j] - v[i]) / 100.0, extra))
        if not lab.startswith("  setup detail"):
            tot += (v[j] - v[i]) / 100.0
print("  %-58s %8.2f" % ("total", tot))
import numpy as np
n = 1024
buf = (C.c_ulonglong * (2 * n))()
if hasattr(lib, "slode_debug_wg_span") and lib.slode_debug_wg_span(buf, n) == 0:
    a = np.array(list(buf), dtype=np.float64).reshape(n, 2) / 100.0
    t0 = a[:, 0].min()
    dur = a[:, 1] - a[:, 0]
    print("== ode kernel, all %d workgroups (us): first start 0.0, last start %.2f, first end %.2f, last end %.2f; duration min/median/max %.2f / %.2f / %.2f"
          % (n, a[:, 0].max() - t0, a[:, 1].min() - t0, a[:, 1].max() - t0, dur.min(), np.median(dur), dur.max()))

if hasattr(lib, "slode_debug_wg_hw"):
    hb = (C.c_uint * (2 * n))()
    if lib.slode_debug_wg_hw(hb, n) == 0:
        hw = np.array(list(hb), dtype=np.int64).reshape(n, 2)
        cu, sh, se, xcc = (hw[:, 0] >> 8) & 15, (hw[:, 0] >> 12) & 1, (hw[:, 0] >> 13) & 7, hw[:, 1] & 15
        print("== ode kernel: workgroup duration (us) by XCC:", {int(x): round(float(np.median(dur[xcc == x])), 1) for x in np.unique(xcc)})
        print("   by SE:", {int(x): round(float(np.median(dur[se == x])), 1) for x in np.unique(se)})
        print("   by blockIdx>>8:", {int(x): round(float(np.median(dur[(np.arange(n) >> 8) == x])), 1) for x in range(4)})
        key = xcc * 1000 + se * 100 + sh * 16 + cu
        cnt = np.array([np.sum(key == kk) for kk in np.unique(key)])
        print("   distinct (xcc,se,sh,cu):", len(np.unique(key)), "workgroups per CU min/max:", cnt.min(), cnt.max())
        per_cu = np.array([dur[key == kk].max() for kk in np.unique(key)])
        print("   per-CU max duration: min/median/max", per_cu.min(), np.median(per_cu), per_cu.max())
        order = np.argsort(dur)[-8:]
        print("   slowest workgroups:", [(int(i), round(float(dur[i]), 1), int(xcc[i]), int(se[i]), int(cu[i]), int(cnt[list(np.unique(key)).index(key[i])])) for i in order])
        simd, wave = (hw[:, 0] >> 4) & 3, hw[:, 0] & 15
        print("   wave 0 of a workgroup sits on SIMD:", {int(x): int(np.sum(simd == x)) for x in np.unique(simd)}, " wave slot:", {int(x): int(np.sum(wave == x)) for x in np.unique(wave)})
        same = [len(set(simd[key == kk])) for kk in np.unique(key)]
        print("   distinct SIMDs hosting the wave 0s of a CU's co-resident workgroups: min/median/max", min(same), int(np.median(same)), max(same))
        print("   start-time by blockIdx>>8 (median):", {int(x): round(float(np.median(a[(np.arange(n) >> 8) == x, 0] - t0)), 2) for x in range(4)})

if hasattr(lib, "slode_debug_stamps_fold"):
    buf = (C.c_ulonglong * 32)()
    assert lib.slode_debug_stamps_fold(buf) == 0
    v = list(buf)
    print("== folded-encoder kernels, workgroup 0 (us)")
    for lab, i, j in (("weff: w' to LDS", 0, 1), ("weff: W_eff rows", 1, 2), ("enc_fwd2: loads", 8, 9), ("enc_fwd2: lin+tanh", 9, 10), ("enc_fwd2:   W_eff stream+FMA (wave 0)", 9, 12), ("enc_fwd2:   wave sums+tanh (wave 0)", 12, 13), ("enc_fwd2:   wait for other waves", 13, 10),
                      ("enc_fwd2: heads", 10, 11), ("chain: staging", 16, 17), ("chain: (i) lin.w", 17, 18), ("chain: (ii) on the other waves + barrier", 18, 19),
                      ("chain: conv taps", 19, 20), ("chain: barrier (block 0)", 20, 21), ("chain: Adam pass (block 0)", 21, 22), ("chain: 2nd barrier (block 0)", 22, 23),
                      ("chain: start -> last block enters conv sum", 16, 24), ("chain: last block conv sum + Adam", 24, 25), ("chain: start -> rider 0 start", 16, 26), ("chain: rider 0", 26, 27)):
        if v[i] and v[j]:
            print("  %-42s %8.2f" % (lab, (v[j] - v[i]) / 100.0))
