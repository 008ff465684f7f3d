"""Diagnostic build of libslode with cycle counters in the loop of the sixteen-lanes-per-trajectory dopri5 FORWARD solve (printf from workgroups
0 and 100).  Usage: python tools/dp5_lpt_clk_build.py -> structured_latent_odes_amd/libslode_clk.so; on the GPU box (copy it to a name that
.gpurunignore does not list): SLODE_LIB_PATH=... python tools/dp5_profile.py 2"""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "structured_latent_odes_amd", "csrc")
s = open(os.path.join(src, "dopri5_kernel.hip")).read()
i0 = s.index("__global__ void __launch_bounds__(WNTH) dopri5_lpt_kernel(const DpK k) {")
i1 = s.index("// ---- reverse mode over the recorded steps")
body = s[i0:i1]
def rep(old, new):
    global body
    assert old in body, old[:70]
    body = body.replace(old, new, 1)
rep("  extern __shared__ __attribute__((aligned(16))) float smem[];", "  extern __shared__ __attribute__((aligned(16))) float smem[];\n  const unsigned long long c_begin = __builtin_readcyclecounter();")
rep("  int j = 1;\n  int steps = bad_grid", "  unsigned long long c_set = __builtin_readcyclecounter(), acc[6] = {0, 0, 0, 0, 0, 0}, c0, c1;\n  int j = 1;\n  int steps = bad_grid")
rep("    ++steps;\n    const float te5[5]", "    ++steps;\n    c0 = __builtin_readcyclecounter();\n    const float te5[5]")
rep("    float av[5], dv[5];\n    if (LPT == 16) {", "    asm volatile(\"\" :: \"v\"(ar[0])); c1 = __builtin_readcyclecounter(); acc[0] += c1 - c0; c0 = c1;\n    float av[5], dv[5];\n    if (LPT == 16) {")
rep("    const float a2 = av[0], d2 = dv[0]", "    asm volatile(\"\" :: \"v\"(av[0]), \"v\"(dv[4])); c1 = __builtin_readcyclecounter(); acc[5] += c1 - c0; c0 = c1;\n    const float a2 = av[0], d2 = dv[0]")
rep("    const float ratio = group_rms_fast<S>(", "    asm volatile(\"\" :: \"v\"(er)); c1 = __builtin_readcyclecounter(); acc[1] += c1 - c0; c0 = c1;\n    const float ratio = group_rms_fast<S>(")
rep("    if (accept) {\n      const float t1 = t + dt;", "    asm volatile(\"\" :: \"v\"(ratio)); c1 = __builtin_readcyclecounter(); acc[2] += c1 - c0; c0 = c1;\n    if (accept) {\n      const float t1 = t + dt;")
rep("    if (act) {\n      float factor;", "    asm volatile(\"\" :: \"v\"(y), \"v\"(t)); c1 = __builtin_readcyclecounter(); acc[3] += c1 - c0; c0 = c1;\n    if (act) {\n      float factor;")
rep("      dt *= factor;\n    }\n  }", "      dt *= factor;\n    }\n    asm volatile(\"\" :: \"v\"(dt)); c1 = __builtin_readcyclecounter(); acc[4] += c1 - c0;\n  }\n"
    "  if (LPT == 16 && (blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(\"dp5fwd16 wg %d attempts %d accepted %d: set-up %llu cycles; per attempt: evals %llu  exchange %llu  combination+error %llu  norm %llu  accept block %llu  factor %llu\\n\","
    " (int)blockIdx.x, steps, nacc, c_set - c_begin, acc[0] / steps, acc[5] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps);")
s = s[:i0] + body + s[i1:]
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
