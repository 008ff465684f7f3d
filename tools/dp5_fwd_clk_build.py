"""Diagnostic build of libslode with cycle counters in the loop of the dopri5 FORWARD solve (printf from workgroups 0 and 100): where an
attempted step's time goes.  Usage (here): python tools/dp5_fwd_clk_build.py -> structured_latent_odes_amd/libslode_clk.so; on the GPU box:
SLODE_LIB_PATH=structured_latent_odes_amd/libslode_clk.so python tools/dp5_profile.py 2"""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "structured_latent_odes_amd", "csrc")
s = open(os.path.join(src, "dopri5_kernel.hip")).read()
i0 = s.index("__global__ void __launch_bounds__(DNT) dopri5_kernel(const DpK k) {")
i1 = s.index("// ---- LPT = 16 / 32 / 64 lanes per trajectory")
body = s[i0:i1]
def rep(old, new):
    global body
    assert old in body, old[:70]
    body = body.replace(old, new, 1)
rep("  int j = 1;\n  int steps = bad_grid", "  unsigned long long c_set = __builtin_readcyclecounter(), acc[6] = {0, 0, 0, 0, 0, 0}, c0, c1;\n  int j = 1;\n  int steps = bad_grid")
rep("    ++steps;\n    // the five evaluation times", "    ++steps;\n    c0 = __builtin_readcyclecounter();\n    // the five evaluation times")
rep("    const float a2 = av[0], d2 = dv[0]", "    c1 = __builtin_readcyclecounter(); acc[0] += c1 - c0; c0 = c1;\n    const float a2 = av[0], d2 = dv[0]")
rep("    const float ratio = group_rms_fast<S>(", "    asm volatile(\"\" :: \"v\"(e)); c1 = __builtin_readcyclecounter(); acc[1] += c1 - c0; c0 = c1;\n    const float ratio = group_rms_fast<S>(")
rep("    if (accept) {\n      const float t1 = t + dt;", "    asm volatile(\"\" :: \"v\"(ratio)); c1 = __builtin_readcyclecounter(); acc[2] += c1 - c0; c0 = c1;\n    if (accept) {\n      const float t1 = t + dt;")
rep("    if (act) {\n      float factor;", "    asm volatile(\"\" :: \"v\"(y), \"v\"(t)); c1 = __builtin_readcyclecounter(); acc[3] += c1 - c0; c0 = c1;\n    if (act) {\n      float factor;")
rep("      dt *= factor;\n    }\n  }", "      dt *= factor;\n    }\n    asm volatile(\"\" :: \"v\"(dt)); c1 = __builtin_readcyclecounter(); acc[4] += c1 - c0;\n  }\n"
    "  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(\"dp5fwd wg %d attempts %d accepted %d: set-up %llu cycles; per attempt: evals %llu  stages+error %llu  norm %llu  accept block (records, outputs) %llu  factor %llu\\n\","
    " (int)blockIdx.x, steps, nacc, c_set - c_begin, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps);")
rep("  extern __shared__ __attribute__((aligned(16))) float smem[];", "  extern __shared__ __attribute__((aligned(16))) float smem[];\n  const unsigned long long c_begin = __builtin_readcyclecounter();")
# set-up detail: stamps inside load_units (workgroup 0, thread 0) through a __device__ array
pre = s[:i0]
def rep_pre(old, new):
    global pre
    assert old in pre, old[:70]
    pre = pre.replace(old, new, 1)
rep_pre("template <int S, int H, int LPT = G>\n__device__ __forceinline__ unsigned load_units(",
        "__device__ unsigned long long g_clk[12];\n#define CLK(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_clk[i] = __builtin_readcyclecounter()\n"
        "template <int S, int H, int LPT = G>\n__device__ __forceinline__ unsigned load_units(")
rep_pre("  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");              // this wave's LDS-DMA has landed; the barrier covers the others'\n  __syncthreads();",
        "  CLK(0);\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  __syncthreads();\n  CLK(1);")
rep_pre("  const unsigned dirmask = group_or(dm);\n  __syncthreads();", "  const unsigned dirmask = group_or(dm);\n  __syncthreads();\n  CLK(2);")
rep_pre("  __syncthreads();\n  // segment table, event by event", "  __syncthreads();\n  CLK(3);\n  // segment table, event by event")
rep_pre("  __syncthreads();\n  return dirmask;", "  __syncthreads();\n  CLK(4);\n  return dirmask;")
rep("  InitRegs<S, H> ir;\n  init_request<S, H>(k.w2, k.b2, g, ir);\n  latent_store<G>(k, bb, live, g, L, LP, s_z + slot * LP, zr);",
    "  CLK(7);\n  InitRegs<S, H> ir;\n  init_request<S, H>(k.w2, k.b2, g, ir);\n  asm volatile(\"\" ::: \"memory\"); CLK(8);\n  latent_store<G>(k, bb, live, g, L, LP, s_z + slot * LP, zr);")
rep("  float y = init_state<S, H>(ir, pre0, g, own);", "  float y = init_state<S, H>(ir, pre0, g, own);\n  asm volatile(\"\" :: \"v\"(y)); CLK(5);")
rep("  unsigned long long c_set = __builtin_readcyclecounter()", "  asm volatile(\"\" :: \"v\"(dt)); CLK(6);\n  unsigned long long c_set = __builtin_readcyclecounter()")
rep("  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(", "  if (blockIdx.x == 0 && tid == 0) printf(\"dp5fwd first phase: units_request issued %llu, init_request issued %llu\\n\", g_clk[7] - c_begin, g_clk[8] - c_begin);\n  if (blockIdx.x == 0 && tid == 0) printf(\"dp5fwd set-up (cycles from kernel start): requests issued, latent rows staged %llu, all set-up operands landed %llu, unit sums %llu, rank %llu, segment table %llu, init state %llu, initial step %llu\\n\","
    " g_clk[0] - c_begin, g_clk[1] - c_begin, g_clk[2] - c_begin, g_clk[3] - c_begin, g_clk[4] - c_begin, g_clk[5] - c_begin, g_clk[6] - c_begin);\n"
    "  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(")
s = pre + body + s[i1:]
tmp = os.path.join(src, "_dp5_clk.hip")
open(tmp, "w").write(s)
others = ["slode_api.hip", "ode_kernel.hip", "encoder_kernels.hip", "misc_kernels.hip", "encoder_fused.hip", "aux_kernel.hip"]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-shared", "-o",
       os.path.join(ROOT, "structured_latent_odes_amd", "libslode_clk.so"), "_dp5_clk.hip"] + others
try:
    subprocess.check_call(cmd, cwd=src)
finally:
    os.remove(tmp)
print("built libslode_clk.so")
