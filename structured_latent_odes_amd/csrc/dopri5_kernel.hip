// Adaptive Dormand-Prince 5(4) latent-ODE solve (forward) for gfx950: the `solver="dopri5"` string the reference can pass
// through to torchdiffeq.odeint (models/blackbox_ode.py:41-45; BASELINE config[2]).
//
// torchdiffeq's controller uses ONE step size for the whole [B,S] tensor (its error norm is an RMS over the batch), which
// makes results depend on batch composition and cannot shard.  Contract here (SURVEY hard part 3): one controller PER
// TRAJECTORY -- lane = trajectory, every lane runs torchdiffeq's algorithm on its own state (FSAL, Hairer initial step,
// ratio = rms(err / (atol + rtol*max(|y0|,|y1|))), factor clamp [0.2, 10] with safety 0.9, quartic dense output through
// (y0, y_mid, y1, f0, f1)) -- validated at solution level against the oracle's per-trajectory restatement and scipy RK45.
// The dynamics weights arrive as SGPR operands (uniform), the per-trajectory hidden offsets u = W_z z + b live in LDS.
#include "slode_common.h"

typedef const __attribute__((address_space(4))) float* cptr;

namespace {

struct DpK {
  int B, T, L;
  const float *times, *z, *w1, *b1, *w2, *b2, *wh, *bh, *wg, *bg, *wd, *bd;
  float* x;
  float rtol, atol;
  int max_steps;
};

constexpr int DPW = 64;  // lanes (= trajectories) per workgroup

template <int S, int H>
__device__ __forceinline__ void dyn(float t, const float* __restrict__ s_wt, const float* __restrict__ s_ul, cptr wg, cptr bg, cptr wd,
                                    cptr bd, const float (&y)[S], float (&f)[S]) {
  asm volatile("" : "+s"(wg), "+s"(wd), "+s"(bg), "+s"(bd));
  float h[H];
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = fmaxf(fmaf(s_wt[j], t, s_ul[j * DPW]), 0.f);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float xa = bg[s], xd = bd[s];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      xa = fmaf(wg[s * H + j], h[j], xa);
      xd = fmaf(wd[s * H + j], h[j], xd);
    }
    f[s] = sigmoidf_fast(xa) - sigmoidf_fast(xd) * y[s];
  }
}

template <int S>
__device__ __forceinline__ float rms(const float (&v)[S]) {
  float a = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) a = fmaf(v[s], v[s], a);
  return sqrtf(a * (1.0f / S));
}

template <int S, int H>
__global__ void __launch_bounds__(DPW) dopri5_kernel(const DpK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_wt = smem;                 // [32] time column of dynamics_hidden
  float* s_u = s_wt + 32;             // [H][DPW] per-trajectory hidden offsets
  float* s_z = s_u + H * DPW;         // [L][DPW]
  const int lane = threadIdx.x, b = blockIdx.x * DPW + lane, L = k.L, T = k.T;
  const bool live = b < k.B;
  const cptr wg = (cptr)k.wg, bg = (cptr)k.bg, wd = (cptr)k.wd, bd = (cptr)k.bd;
  if (lane < 32) s_wt[lane] = lane < H ? k.wh[lane * (1 + L)] : 0.f;
  for (int l = 0; l < L; ++l) s_z[l * DPW + lane] = live ? k.z[(long long)(live ? b : 0) * L + l] : 0.f;
  __syncthreads();
  // u = W_z z + b_h ; x0 = sigmoid(W2 relu(W1 z + b1) + b2)   (blackbox_ode.py:19-22, 97-101)
  float y[S];
  {
    const cptr wh = (cptr)k.wh, bh = (cptr)k.bh, w1 = (cptr)k.w1, b1 = (cptr)k.b1, w2 = (cptr)k.w2, b2 = (cptr)k.b2;
    float o[S];
#pragma unroll
    for (int s = 0; s < S; ++s) o[s] = b2[s];
    for (int j = 0; j < H; ++j) {
      float uj = bh[j], p0 = b1[j];
      for (int l = 0; l < L; ++l) {
        const float zl = s_z[l * DPW + lane];
        uj = fmaf(wh[j * (1 + L) + 1 + l], zl, uj);
        p0 = fmaf(w1[j * L + l], zl, p0);
      }
      s_u[j * DPW + lane] = uj;
      const float hj = fmaxf(p0, 0.f);
#pragma unroll
      for (int s = 0; s < S; ++s) o[s] = fmaf(w2[s * H + j], hj, o[s]);
    }
#pragma unroll
    for (int s = 0; s < S; ++s) y[s] = sigmoidf_fast(o[s]);
  }
  const float* s_ul = s_u + lane;
  float* xo = k.x + (long long)(live ? b : 0) * T * S;
  if (live) {
#pragma unroll
    for (int s = 0; s < S; ++s) xo[s] = y[s];
  }
  const float rtol = k.rtol, atol = k.atol;
  float t = k.times[0];
  float fcur[S];
  dyn<S, H>(t, s_wt, s_ul, wg, bg, wd, bd, y, fcur);
  // Hairer's initial step (torchdiffeq _select_initial_step, order 4)
  float dt;
  {
    float a0[S], a1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { const float sc = atol + fabsf(y[s]) * rtol; a0[s] = y[s] / sc; a1[s] = fcur[s] / sc; }
    const float d0 = rms<S>(a0), d1 = rms<S>(a1);
    const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    float y1[S], f1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) y1[s] = fmaf(h0, fcur[s], y[s]);
    dyn<S, H>(t + h0, s_wt, s_ul, wg, bg, wd, bd, y1, f1);
#pragma unroll
    for (int s = 0; s < S; ++s) a0[s] = (f1[s] - fcur[s]) / (atol + fabsf(y[s]) * rtol);
    const float d2 = rms<S>(a0) / h0;
    const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
    dt = fminf(100.f * h0, h1);
  }
  int j = 1;
  int steps = 0;
  // every lane leaves the loop: either all outputs written or max_steps reached (outputs then hold the last state)
  while (__any(live && j < T && steps < k.max_steps)) {
    const bool act = live && j < T && steps < k.max_steps;
    ++steps;
    float k2[S], k3[S], k4[S], k5[S], k6[S], k7[S], yi[S];
#pragma unroll
    for (int s = 0; s < S; ++s) yi[s] = fmaf(dt, (1.f / 5) * fcur[s], y[s]);
    dyn<S, H>(t + dt * (1.f / 5), s_wt, s_ul, wg, bg, wd, bd, yi, k2);
#pragma unroll
    for (int s = 0; s < S; ++s) yi[s] = fmaf(dt, (3.f / 40) * fcur[s] + (9.f / 40) * k2[s], y[s]);
    dyn<S, H>(t + dt * (3.f / 10), s_wt, s_ul, wg, bg, wd, bd, yi, k3);
#pragma unroll
    for (int s = 0; s < S; ++s) yi[s] = fmaf(dt, (44.f / 45) * fcur[s] + (-56.f / 15) * k2[s] + (32.f / 9) * k3[s], y[s]);
    dyn<S, H>(t + dt * (4.f / 5), s_wt, s_ul, wg, bg, wd, bd, yi, k4);
#pragma unroll
    for (int s = 0; s < S; ++s)
      yi[s] = fmaf(dt, (19372.f / 6561) * fcur[s] + (-25360.f / 2187) * k2[s] + (64448.f / 6561) * k3[s] + (-212.f / 729) * k4[s], y[s]);
    dyn<S, H>(t + dt * (8.f / 9), s_wt, s_ul, wg, bg, wd, bd, yi, k5);
#pragma unroll
    for (int s = 0; s < S; ++s)
      yi[s] = fmaf(dt, (9017.f / 3168) * fcur[s] + (-355.f / 33) * k2[s] + (46732.f / 5247) * k3[s] + (49.f / 176) * k4[s] + (-5103.f / 18656) * k5[s], y[s]);
    dyn<S, H>(t + dt, s_wt, s_ul, wg, bg, wd, bd, yi, k6);
    float y1[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
      y1[s] = fmaf(dt, (35.f / 384) * fcur[s] + (500.f / 1113) * k3[s] + (125.f / 192) * k4[s] + (-2187.f / 6784) * k5[s] + (11.f / 84) * k6[s], y[s]);
    dyn<S, H>(t + dt, s_wt, s_ul, wg, bg, wd, bd, y1, k7);
    float er[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float e = dt * ((35.f / 384 - 1951.f / 21600) * fcur[s] + (500.f / 1113 - 22642.f / 50085) * k3[s] + (125.f / 192 - 451.f / 720) * k4[s] +
                            (-2187.f / 6784 + 12231.f / 42400) * k5[s] + (11.f / 84 - 649.f / 6300) * k6[s] + (-1.f / 60) * k7[s]);
      er[s] = e / (atol + rtol * fmaxf(fabsf(y[s]), fabsf(y1[s])));
    }
    const float ratio = rms<S>(er);
    // a step at the resolution floor of fp32 time is accepted regardless (torchdiffeq would raise 'underflow in dt')
    const bool accept = act && (ratio <= 1.f || dt <= 16.f * 1.1920929e-7f * fmaxf(fabsf(t), 1.f));
    if (accept) {
      const float t1 = t + dt;
      if (j < T && k.times[j] <= t1) {
        float ymid[S];
#pragma unroll
        for (int s = 0; s < S; ++s)
          ymid[s] = fmaf(dt, (6025192743.f / 30085553152.f / 2) * fcur[s] + (51252292925.f / 65400821598.f / 2) * k3[s] +
                                 (-2691868925.f / 45128329728.f / 2) * k4[s] + (187940372067.f / 1594534317056.f / 2) * k5[s] +
                                 (-1776094331.f / 19743644256.f / 2) * k6[s] + (11237099.f / 235043384.f / 2) * k7[s], y[s]);
        while (j < T && k.times[j] <= t1) {
          const float xq = (k.times[j] - t) / dt;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float ca = 2.f * dt * (k7[s] - fcur[s]) - 8.f * (y1[s] + y[s]) + 16.f * ymid[s];
            const float cb = dt * (5.f * fcur[s] - 3.f * k7[s]) + 18.f * y[s] + 14.f * y1[s] - 32.f * ymid[s];
            const float cc = dt * (k7[s] - 4.f * fcur[s]) - 11.f * y[s] - 5.f * y1[s] + 16.f * ymid[s];
            const float cd = dt * fcur[s];
            xo[j * S + s] = y[s] + xq * (cd + xq * (cc + xq * (cb + xq * ca)));
          }
          ++j;
        }
      }
      t = t1;
#pragma unroll
      for (int s = 0; s < S; ++s) { y[s] = y1[s]; fcur[s] = k7[s]; }
    }
    if (act) {
      float factor;
      if (ratio == 0.f) factor = 10.f;
      else {
        const float safe = 0.9f * powf(ratio, -0.2f);
        factor = fminf(10.f, fmaxf(safe, ratio < 1.f ? 1.f : 0.2f));
      }
      dt *= factor;
    }
  }
  if (live)  // max_steps exhausted: fill the remaining outputs with the last state (finite, and flagged by the host via steps)
    for (; j < T; ++j)
#pragma unroll
      for (int s = 0; s < S; ++s) xo[j * S + s] = y[s];
}

}  // namespace

hipError_t slode_launch_dopri5(const slode_shape& s, const slode_layout& lay, const float* p, const float* times, const float* z,
                               float* x, hipStream_t stream) {
  DpK k;
  k.B = s.B; k.T = s.T; k.L = s.L; k.times = times; k.z = z; k.x = x;
  k.w1 = p + lay.init_w1; k.b1 = p + lay.init_b1; k.w2 = p + lay.init_w2; k.b2 = p + lay.init_b2;
  k.wh = p + lay.dyn_wh; k.bh = p + lay.dyn_bh; k.wg = p + lay.dyn_wg; k.bg = p + lay.dyn_bg; k.wd = p + lay.dyn_wd; k.bd = p + lay.dyn_bd;
  k.rtol = s.rtol > 0.f ? s.rtol : 1e-7f;
  k.atol = s.atol > 0.f ? s.atol : 1e-9f;
  k.max_steps = 20000;
  const int grid = (s.B + DPW - 1) / DPW;
  const size_t lds = sizeof(float) * (32 + (size_t)s.H * DPW + (size_t)s.L * DPW);
  if (s.H == 25 && s.S == 5) hipLaunchKernelGGL((dopri5_kernel<5, 25>), dim3(grid), dim3(DPW), lds, stream, k);
  else if (s.H == 25 && s.S == 8) hipLaunchKernelGGL((dopri5_kernel<8, 25>), dim3(grid), dim3(DPW), lds, stream, k);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
