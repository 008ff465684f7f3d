// Microbenchmark: what a LONE wave (one wave per SIMD, as in the adaptive-solver kernels: DESIGN 3.3) pays per instruction on gfx950, by
// encoding size (4-byte VOP2, 8-byte VOP2 + literal, 8-byte VOP3), by dependence (one chain / eight independent chains) and by waves per
// workgroup.  Straight-line code of 4096 instructions between two s_memtime reads.
// Build: hipcc --offload-arch=gfx950 -O3 lone_wave_issue.hip -o lone_wave_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
#define R512(x) R8(R64(x))
#define R4096(x) R8(R512(x))
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (KIND == 0) asm volatile(R4096("v_add_f32_e32 %0, %1, %0\n") : "+v"(x0) : "v"(a));
  if (KIND == 1) asm volatile(R512("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                                   "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n")
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
  if (KIND == 2) asm volatile(R4096("v_add_f32_e32 %0, 0x3f800001, %0\n") : "+v"(x0));
  if (KIND == 3) asm volatile(R512("v_add_f32_e32 %0, 0x3f800001, %0\n v_add_f32_e32 %1, 0x3f800001, %1\n v_add_f32_e32 %2, 0x3f800001, %2\n v_add_f32_e32 %3, 0x3f800001, %3\n"
                                   "v_add_f32_e32 %4, 0x3f800001, %4\n v_add_f32_e32 %5, 0x3f800001, %5\n v_add_f32_e32 %6, 0x3f800001, %6\n v_add_f32_e32 %7, 0x3f800001, %7\n")
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  if (KIND == 4) asm volatile(R4096("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
  if (KIND == 5) asm volatile(R512("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  if (KIND == 6) asm volatile(R4096("v_exp_f32_e32 %0, %0\n") : "+v"(x0));
  if (KIND == 7) asm volatile(R512("v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n"
                                   "v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n")
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  if (KIND == 8) asm volatile(R4096("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n") : "+v"(x0));   // (8192 instructions)
  if (KIND == 9) asm volatile(R512("v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %0, %1, %0\n"
                                   "v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %0, %1, %0\n v_add_f32_e32 %0, %1, %0\n s_cmp_eq_u32 s4, s5\n") : "+v"(x0) : "v"(a) : "scc");   // 7 VALU + 1 SALU
  if (KIND == 10) asm volatile(R4096("v_fmac_f32_e32 %0, %1, %0\n") : "+v"(x0) : "v"(a));
  if (KIND == 11) asm volatile(R4096("v_mul_f32_e32 %0, %1, %0\n") : "+v"(x0) : "v"(a));
  if (KIND == 12) asm volatile(R4096("v_cndmask_b32_e32 %0, %1, %0, vcc\n") : "+v"(x0) : "v"(a) : "vcc");
  if (KIND == 13) asm volatile(R4096("v_rcp_f32_e32 %0, %0\n") : "+v"(x0));
  const unsigned long long c1 = __builtin_readcyclecounter();
  asm volatile("s_nop 0" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}
template <int KIND>
void run(const char* name, int n_instr, float* out, unsigned long long* cyc) {
  for (int threads : {64, 128, 256, 512}) {
    unsigned long long h[8] = {0};
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s waves/workgroup %d: %.2f cycles per instruction (wave 0; %d instructions)\n", name, threads / 64, (double)h[0] / n_instr, n_instr);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  run<0>("v_add e32 (4 B), one dependent chain", 4096, out, cyc);
  run<1>("v_add e32 (4 B), eight independent chains", 4096, out, cyc);
  run<2>("v_add e32 + literal (8 B), dependent", 4096, out, cyc);
  run<3>("v_add e32 + literal (8 B), independent", 4096, out, cyc);
  run<4>("v_fma VOP3 (8 B), dependent", 4096, out, cyc);
  run<5>("v_fma VOP3 (8 B), independent", 4096, out, cyc);
  run<10>("v_fmac e32 (4 B), dependent", 4096, out, cyc);
  run<11>("v_mul e32 (4 B), dependent", 4096, out, cyc);
  run<12>("v_cndmask e32 (4 B), dependent", 4096, out, cyc);
  run<6>("v_exp (4 B), dependent", 4096, out, cyc);
  run<7>("v_exp (4 B), independent", 4096, out, cyc);
  run<13>("v_rcp (4 B), dependent", 4096, out, cyc);
  run<8>("v_add dpp + s_nop 1, dependent (per pair)", 4096, out, cyc);
  run<9>("7 dependent v_add + 1 s_cmp (per instruction)", 4096, out, cyc);
  return 0;
}
