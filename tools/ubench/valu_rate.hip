// Microbenchmark: VALU issue rate per SIMD as a function of waves per SIMD and instruction kind (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, const float* __restrict__ w) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  typedef const __attribute__((address_space(4))) float* cptr;
  cptr ws = (cptr)w;
  for (int i = 0; i < N_IT; ++i) {
    if (KIND == 0) {  // v_fma VGPR operands
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (KIND == 1) {  // v_fmac with SGPR weight (s_load) operand
      const float w0 = ws[i & 63], w1 = ws[(i + 1) & 63];
      x0 = fmaf(w0, x1, x0); x1 = fmaf(w1, x2, x1); x2 = fmaf(w0, x3, x2); x3 = fmaf(w1, x4, x3);
      x4 = fmaf(w0, x5, x4); x5 = fmaf(w1, x6, x5); x6 = fmaf(w0, x7, x6); x7 = fmaf(w1, x0, x7);
    } else if (KIND == 2) {  // v_max + v_cndmask mix
      x0 = fmaxf(x0, a); x1 = x1 > b ? x0 : x1; x2 = fmaxf(x2, a); x3 = x3 > b ? x2 : x3;
      x4 = fmaxf(x4, a); x5 = x5 > b ? x4 : x5; x6 = fmaxf(x6, a); x7 = x7 > b ? x6 : x7;
    } else {  // v_exp
      x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
      x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
void run(const char* name, int wgs_per_cu, float* out, float* w) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, w);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, w);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0;  // per launch
  const double instr_per_simd = (double)wgs_per_cu * N_IT * 8;  // wave-instructions per SIMD (1 wave of each WG per SIMD)
  printf("%-28s waves/SIMD=%d  %.1f us  => %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wgs_per_cu, us,
         us * 2400.0 / instr_per_simd);
}
int main() {
  float *out, *w; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&w, 256); hipMemset(w, 0, 256);
  for (int n : {1, 2, 4, 8}) { run<0>("v_fma (vgpr)", n, out, w); run<1>("v_fmac (sgpr weight)", n, out, w); run<2>("v_max/v_cndmask", n, out, w); run<3>("v_exp", n, out, w); }
  return 0;
}
