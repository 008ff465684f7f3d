// Micro-check: DPP wave_shl:1 / wave_shr:1 on gfx950 -- lane i receives lane i+1's (resp. i-1's) value across the whole wave64 in one VALU
// instruction (no LDS).  Build: hipcc --offload-arch=gfx950 -O3 dpp_wave_shl.hip -o dpp_wave_shl ; prints the first mismatching lane or OK.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, float* shl, float* shr) {
  const int i = threadIdx.x;
  const float x = a[i];
  shl[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
  shr[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
int main() {
  float h[128], l[128], r[128], *da, *dl, *dr;
  for (int i = 0; i < 128; ++i) h[i] = 100.f + i;
  hipMalloc(&da, sizeof h); hipMalloc(&dl, sizeof h); hipMalloc(&dr, sizeof h);
  hipMemcpy(da, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, da, dl, dr);
  hipMemcpy(l, dl, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(r, dr, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) {
    const int lane = i & 63;
    const float wl = lane < 63 ? h[i + 1] : -1.f, wr = lane > 0 ? h[i - 1] : -1.f;
    if (l[i] != wl || r[i] != wr) { if (!bad) printf("mismatch at %d: shl %g (want %g) shr %g (want %g)\n", i, l[i], wl, r[i], wr); ++bad; }
  }
  printf(bad ? "FAIL (%d lanes)\n" : "OK: wave_shl:1 gives lane i+1, wave_shr:1 gives lane i-1, the edge lane keeps `old`\n", bad);
  printf("lanes 0,1,62,63: shl %g %g %g %g   shr %g %g %g %g\n", l[0], l[1], l[62], l[63], r[0], r[1], r[62], r[63]);
  return bad != 0;
}
