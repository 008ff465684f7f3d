"""Diagnostic build of libslode with cycle stamps in the dopri5 reverse sweep (printf from workgroups 0 and 100): where the kernel's time goes.
Usage (here): python tools/dp5_clk_build.py   -> structured_latent_odes_amd/libslode_clk.so ; on the GPU box:
SLODE_LIB_PATH=structured_latent_odes_amd/libslode_clk.so python tools/dp5_profile.py 3"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "structured_latent_odes_amd", "csrc")
s = open(os.path.join(src, "dopri5_kernel.hip")).read()
def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)
rep("  GroupLds<H> m;\n  float* s_gu = m.carve(smem, BTP);", "  unsigned long long clk[12]; int nclk = 0;\n#define CLK() clk[nclk++] = __builtin_readcyclecounter()\n  CLK();\n  GroupLds<H> m;\n  float* s_gu = m.carve(smem, BTP);")
rep("  {\n    constexpr int ZQ = SLODE_MAX_L / G;\n    float zv[ZQ];", "  unsigned long long c_a = __builtin_readcyclecounter();\n  {\n    constexpr int ZQ = SLODE_MAX_L / G;\n    float zv[ZQ];")
rep("  Units w;\n  unsigned dirmask;\n  if (k.tabs) {", "  unsigned long long c_b = __builtin_readcyclecounter(), c_c = c_b;\n  CLK();\n  Units w;\n  unsigned dirmask;\n  if (k.tabs) {")
rep("  const float* s_us = m.u + slot * 32;\n  const int* s_rnk", "  CLK();\n  const float* s_us = m.u + slot * 32;\n  const int* s_rnk")
rep("  const bool any_bad = __syncthreads_or(live && bad) != 0;", "  CLK();\n  const bool any_bad = __syncthreads_or(live && bad) != 0;\n  CLK();")
rep("    __syncthreads();\n    for (int col = tid; col < (H + 1) * NTMP; col += BNT) {", "    CLK();\n    __syncthreads();\n    for (int col = tid; col < (H + 1) * NTMP; col += BNT) {")
rep("  // ---- init net: x0 = sigmoid(W2 relu(W1 z + b1) + b2); the j = 0 output is x0 itself", "  CLK();\n  // ---- init net: x0 = sigmoid(W2 relu(W1 z + b1) + b2); the j = 0 output is x0 itself")
rep("    // latent gradient of this trajectory: through the init net and (exact mode)", "    CLK();\n    // latent gradient of this trajectory: through the init net and (exact mode)")
rep("    // sums over the workgroup's trajectories (fixed order): outer products with z", "    CLK();\n    // sums over the workgroup's trajectories (fixed order): outer products with z")
i = s.index("}  // namespace grp")
j = s.rfind("}\n", 0, i)
s = s[:j] + "  CLK();\n  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(\"dp5bwd wg %d stage detail: requests + zero-fill %llu, latent rows %llu (%llu %llu)\\n\", (int)blockIdx.x, c_a-clk[0], c_b-c_a, c_c-c_b, clk[1]-c_c);\n  if ((blockIdx.x == 0 || blockIdx.x == 100) && tid == 0) printf(\"dp5bwd wg %d K %d cycles: stage %llu units+table %llu loop %llu wait %llu snapshots %llu colsums %llu initnet %llu latent %llu outer %llu\\n\", (int)blockIdx.x, K, clk[1]-clk[0], clk[2]-clk[1], clk[3]-clk[2], clk[4]-clk[3], clk[5]-clk[4], clk[6]-clk[5], clk[7]-clk[6], clk[8]-clk[7], clk[9]-clk[8]);\n" + s[j:]
tmp = os.path.join(src, "_dp5_clk.hip")
open(tmp, "w").write(s)
others = [f for f in ("slode_api.hip", "ode_kernel.hip", "encoder_kernels.hip", "misc_kernels.hip", "encoder_fused.hip", "aux_kernel.hip")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-shared", "-o", os.path.join(ROOT, "structured_latent_odes_amd", "libslode_clk.so"), "_dp5_clk.hip"] + others
try:
    subprocess.check_call(cmd, cwd=src)
finally:
    os.remove(tmp)
print("built libslode_clk.so")
