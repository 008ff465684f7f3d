// Folded encoder path used by slode_elbo_step (gfx950).
//
// The reference's EncoderCONV applies conv1d -> avg_pool1d -> flatten -> Linear with NO nonlinearity in between
// (models/encoder_conv.py:44-47), so for fixed weights the pre-tanh activation is one affine map of the raw observation row:
//     pre[b][m] = b_eff[m] + sum_kappa W_eff[m][kappa] * x[b][kappa],       kappa = memory index of (c, t) inside a trajectory
//     W_eff[m][(c,t)] = sum_f sum_{k'} lin.weight[m][f*n_pool + (t - k')] * w'[f][c][k'],   w' = conv taps convolved with the box filter
//     b_eff[m] = lin.bias[m] + sum_f conv.bias[f] * rowsum[m][f],            rowsum[m][f] = sum_q lin.weight[m][f*n_pool + q]
// Folding costs Hc*C*T*F*(K+P-1) = 4.2 MFLOP once per step (weights change every step) and removes, per trajectory, the
// conv/pool stack and the 374 KB lin.weight stream: the forward becomes a [B x C*T] x [C*T x Hc] product on the raw rows (5x
// fewer FLOPs), the backward a K = B MFMA GEMM with N = C*T followed by the chain rule back to lin.weight / conv.weight.
// Results differ from the layer-by-layer evaluation only by fp32 summation order (~1e-6 relative; tests bound it at 2e-5).
//
// Requires each trajectory's C*T observations to be one dense block (sb == C*T, {sc, st} == {1, C} or {T, 1}); other strides
// use the layer-by-layer kernels of encoder_kernels.hip.
#include "slode_common.h"

typedef const __attribute__((address_space(4))) float* cptr;

#ifdef SLODE_STAMPS
__device__ unsigned long long g_stamps_fold[32];
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps_fold[i] = wall_clock64(); } while (0)
#define STAMP_ANY(i) do { if (threadIdx.x == 0) g_stamps_fold[i] = wall_clock64(); } while (0)
extern "C" int slode_debug_stamps_fold(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_fold), sizeof(unsigned long long) * 32);
}
#else
#define STAMP(i) do { } while (0)
#define STAMP_ANY(i) do { } while (0)
#endif

namespace {

constexpr int TBE = 4;
constexpr int FNT = 1024;  // enc_fwd2 threads
constexpr int RB = 4, IU = 5;

struct FoldK {
  int B, T, C, L, F, K, P, Hc, n_conv, n_pool, FQ, CT, J;
  int t_major;  // 1: kappa = t*C + c ([B,T,C] contiguous); 0: kappa = c*T + t ([B,C,T] contiguous)
  const float *conv_w, *conv_b, *lin_w, *lin_b, *zloc_w, *zloc_b, *zls_w, *zls_b;
  const float* x;       // dense rows [B][CT]
  float *weff, *rowsum, *wprime, *beff;    // [Hc][CT], [Hc][F], [F][C][J], [Hc]
  float *loc, *scale, *hid;                // encoder outputs / saved tanh
  const float *scale_in, *hid_in, *g_loc, *g_scale;
  float *g_pre, *slabs;                    // [B][64], small2 slabs
  int small_stride;
  const float* gslabs; int n_gslabs;       // split-K partials of G = g_pre^T X  [n][Hc*CT]
  float* g_lin_w;                          // final lin.weight gradient (written directly)
  float* conv_slabs;                       // [Hc][F*C*K + F]
  unsigned int* counter;
  const float* cstd; float* sigtab; int gauss;   // likelihood-scale table of this step: [4][CT] (sigtab == nullptr: none)
};

__device__ __forceinline__ int kappa_of(const FoldK& k, int c, int t) { return k.t_major ? t * k.C + c : c * k.T + t; }

// ---- fold: W_eff, rowsum, b_eff, w' ---------------------------------------------------------------------------------
constexpr int WPB = 128;   // W_eff outputs per block of weff_kernel
constexpr int WFG = 4;     // threads per output (filter groups): 8 waves per CU instead of 4 -- the output loop is latency-bound
constexpr int WNT = WPB * WFG;
// One output W_eff[m][kappa(c, t)] by the four lanes of a quad (fh = lane of the quad = group of filters; fixed order: the quad's partial
// sums meet in two DPP adds).  wlm: row m of lin.weight [F][n_pool]; s_wp: w' [F][C][JM].  Shared by weff_kernel and by the in-launch fold
// of enc_chain_kernel: the same operations in the same order, hence the same bits.
template <int JM>
__device__ __forceinline__ float weff_out(const FoldK& k, const float* __restrict__ wlm, const float* __restrict__ s_wp, int c, int t, int fh) {
  const int C = k.C, fper = (k.F + WFG - 1) / WFG;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 2
  for (int f = fh * fper; f < min(k.F, (fh + 1) * fper); ++f) {
    const float* wl = wlm + f * k.n_pool;
    const float* wp = s_wp + (f * C + c) * JM;
    float v[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) v[j] = wl[min(max(t - j, 0), k.n_pool - 1)];   // unconditional, clamped
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      acc0 = fmaf((t - j >= 0 && t - j < k.n_pool) ? v[j] : 0.f, wp[j], acc0);
      if (j + 1 < JM) acc1 = fmaf((t - j - 1 >= 0 && t - j - 1 < k.n_pool) ? v[j + 1] : 0.f, wp[j + 1], acc1);
    }
  }
  float r = acc0 + acc1;
  r += dpp_f<0xB1>(r);   // quad_perm [1,0,3,2]
  r += dpp_f<0x4E>(r);   // quad_perm [2,3,0,1]
  return r;
}
// One wave: rowsum[m][f] for every f, then b_eff[m] (wlm: row m of lin.weight; conv_b / lin_b_m: the biases to fold in).  Shared likewise.
__device__ __forceinline__ void rowsum_beff_wave(const FoldK& k, const float* __restrict__ wlm, int m, int lane, const float* __restrict__ conv_b,
                                                 float lin_b_m) {
  // all F partial sums advance together so F loads are in flight per pass (a serial f loop costs one HBM round trip per filter)
  float sv[SLODE_MAX_F];
#pragma unroll
  for (int f = 0; f < SLODE_MAX_F; ++f) sv[f] = 0.f;
  for (int q = lane; q < k.n_pool; q += 64) {
#pragma unroll
    for (int f = 0; f < SLODE_MAX_F; ++f) {
      const float v = wlm[min(f, k.F - 1) * k.n_pool + q];
      sv[f] += (f < k.F) ? v : 0.f;
    }
  }
  static_assert(SLODE_MAX_F == 16, "wave_sum16 reduces the SLODE_MAX_F = 16 filter sums");
  const float sf = wave_sum16(sv, lane);     // lane holds rowsum[m][(lane >> 2) & 15]
  const int fl = (lane >> 2) & 15;
  if ((lane & 3) == 0 && fl < k.F) k.rowsum[m * k.F + fl] = sf;
  float be = ((lane & 3) == 0 && fl < k.F) ? conv_b[min(fl, k.F - 1)] * sf : 0.f;
  be = wave_sum(be) + lin_b_m;
  if (lane == 0) k.beff[m] = be;
}

template <int JM>  // compile-time bound on J = K + P - 1 (14 for every reference config)
// (pl_*: the pointers of the kernel's first loads as leading arguments -- preloaded into SGPRs at wave launch, see ode_elbo_kernel)
__global__ void __launch_bounds__(WNT) weff_kernel(const float* __restrict__ pl_conv_w, const float* __restrict__ pl_lin_w, const FoldK k,
                                                   const int stage_rows) {
  __shared__ float s_wp[SLODE_MAX_F * SLODE_MAX_C * JM];
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [stage_rows ? the block's lin.weight rows : 0]
  const int tid = threadIdx.x, J = k.J, C = k.C, K = k.K;
  const float fP = (float)k.P;
  // arrival counters of the chain launch, 128 B apart: slots 0..7 one per filter pair; the in-launch fold's: 8 conv pairs done, 9 riders
  // done, 16 + m the blocks of hidden unit m
  if (blockIdx.x == 0 && tid < 16 + SLODE_MAX_HC && k.counter) k.counter[32 * tid] = 0u;
  STAMP(0);
  const int n_w = k.Hc * k.CT;
  const int nb_w = (n_w + WPB - 1) / WPB, nb_rs = (k.Hc + WNT / 64 - 1) / (WNT / 64);
  float* s_lw = smem;
  // Everything this block reads from global memory is fetched in ONE batch: the conv taps, and (W_eff blocks) the lin.weight rows
  // of the hidden units its WPB outputs belong to (2 for C*T >= WPB) -- coalesced, instead of F*JM = 140 strided loads per thread.
  const int e_first = (int)blockIdx.x * WPB, m0 = min(e_first, n_w - 1) / k.CT, m1 = min(e_first + WPB - 1, n_w - 1) / k.CT;
  {
    // Everything the block needs from global memory is requested in one batch and only then stored: the lin.weight rows, and -- per
    // thread -- the (up to P) conv taps under its own w' entry, so that w' needs no staged copy of the taps and no barrier of its own:
    //   w'[f][c][j] = (1/P) * sum_{k + p = j} w[f][c][k]   (rows padded to JM, zero beyond J)
    const bool rows = stage_rows && (int)blockIdx.x < nb_w;
    const int n_st = rows ? (m1 - m0 + 1) * k.FQ : 0;
    const float* src = pl_lin_w + (long long)m0 * k.FQ;
    const int n_wp = k.F * C * JM;
    const int e = min(tid, n_wp - 1), j = e % JM, fc = e / JM;
    float tp[SLODE_MAX_P];
#pragma unroll
    for (int p = 0; p < SLODE_MAX_P; ++p) {
      const int kk = j - p;
      const bool on = p < k.P && j < J && kk >= 0 && kk < K;
      tp[p] = on ? pl_conv_w[fc * K + min(max(kk, 0), K - 1)] : 0.f;
    }
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = rows ? src[min(tid + q * WNT, n_st - 1)] : 0.f;
    if (tid < n_wp) {
      float sw = 0.f;
#pragma unroll
      for (int p = 0; p < SLODE_MAX_P; ++p) sw += tp[p];
      sw = sw / fP;
      s_wp[tid] = sw;
      if (blockIdx.x == 0 && j < J) k.wprime[fc * J + j] = sw;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (tid + q * WNT < n_st) s_lw[tid + q * WNT] = v[q];
    for (int e2 = tid + WNT; e2 < n_wp; e2 += WNT) {   // (more than 512 entries: long-tap shapes)
      const int j2 = e2 % JM, fc2 = e2 / JM;
      float sw = 0.f;
      for (int p = 0; p < k.P; ++p) {
        const int kk = j2 - p;
        if (j2 < J && kk >= 0 && kk < K) sw += pl_conv_w[fc2 * K + kk];
      }
      sw = sw / fP;
      s_wp[e2] = sw;
      if (blockIdx.x == 0 && j2 < J) k.wprime[fc2 * J + j2] = sw;
    }
    for (int i0 = tid + 8 * WNT; i0 < n_st; i0 += 8 * WNT) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[min(i0 + q * WNT, n_st - 1)];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q * WNT < n_st) s_lw[i0 + q * WNT] = v[q];
    }
  }
  __syncthreads();
  STAMP(1);
  if ((int)blockIdx.x < nb_w) {
    // WPB = 128 outputs per block, WFG = 4 threads per output (each a group of the filters): ~250 blocks fill the 256 CUs.  The four
    // threads of an output are one quad: their partial sums meet in two DPP adds (fixed order), no LDS, no barrier.
    static_assert(WFG == 4, "the filter groups of an output are the lanes of a quad");
    const int el = tid >> 2, fh = tid & 3;
    const int e = min((int)blockIdx.x * WPB + el, n_w - 1);
    const int m = e / k.CT, kap = e - m * k.CT;
    int c, t;
    if (k.t_major) { t = kap / C; c = kap - t * C; } else { c = kap / k.T; t = kap - c * k.T; }
    const float* wlm = stage_rows ? s_lw + (m - m0) * k.FQ : k.lin_w + (long long)m * k.FQ;
    const float r = weff_out<JM>(k, wlm, s_wp, c, t, fh);
    if (fh == 0 && (int)blockIdx.x * WPB + el < n_w) k.weff[e] = r;
    STAMP(2);
  } else if ((int)blockIdx.x >= nb_w + nb_rs) {
    // likelihood scales of this step (decoders.py:52-53: softplus(constant_std)) and what the likelihood and its gradient need of them:
    // parameter-only, so once per step here instead of once per trajectory in the ODE/ELBO kernel (same functions: same bits)
    const int i = ((int)blockIdx.x - nb_w - nb_rs) * WNT + tid;
    if (i < k.CT) {
      const float sig = softplusf(k.cstd[i]);
      k.sigtab[i] = sig;
      k.sigtab[k.CT + i] = 1.0f / sig;
      k.sigtab[2 * k.CT + i] = k.gauss ? logf(sig) : logf(2.f * sig);
      k.sigtab[3 * k.CT + i] = 1.f - expf(-sig);
    }
  } else {
    // one wave per hidden unit m: rowsum[m][f] for every f, then b_eff[m]
    const int m = ((int)blockIdx.x - nb_w) * (WNT / 64) + (tid >> 6), lane = tid & 63;
    if (m < k.Hc) rowsum_beff_wave(k, k.lin_w + (long long)m * k.FQ, m, lane, k.conv_b, k.lin_b[m]);
  }
}

// ---- forward: hid = tanh(W_eff x + b_eff), heads --------------------------------------------------------------------
// TB trajectories per 1024-thread block (4, or 8 for large batches: every block streams the whole W_eff -- 120 KB at the metric shape --
// so at B = 4096 the 1024 blocks of the 4-trajectory form pull 160 MB through the L2s: 30 us; twice the trajectories per block halve it)
// VW floats per lane and request: 2 (8-byte requests, any even C*T) or 4 (16-byte requests -- global_load_dwordx4 / ds_read_b128, the width the
// memory pipeline is built for: C*T a multiple of 4, rows 16-byte aligned); a tile of a row is 64 * VW * IUV columns
template <int VW> struct EncVec { typedef __attribute__((ext_vector_type(VW))) float T; static constexpr int IUV = VW == 4 ? 3 : IU; };
// NP: pieces of 64 * VW columns a row has (VW = 4: two for C*T <= 512 -- 400 and 300 among the BASELINE shapes -- else three; a third piece that
// no lane has a column in is a third of the multiply-adds, of the LDS reads and of the W_eff requests for nothing)
template <int NSET, int VW, int NP = EncVec<VW>::IUV>
__device__ __forceinline__ void enc_tile_fma(const typename EncVec<VW>::T (&w)[RB][NP], const float* s_x, int CT, int i0,
                                             float (&acc)[NSET][RB * TBE]) {
  typedef typename EncVec<VW>::T V;
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const bool in = i0 + 64 * VW * u < CT;
    const int i = min(i0 + 64 * VW * u, CT - VW);
#pragma unroll
    for (int st = 0; st < NSET; ++st)
#pragma unroll
      for (int tb = 0; tb < TBE; ++tb) {
        const V pv = *reinterpret_cast<const V*>(s_x + (st * TBE + tb) * CT + i);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          float a = acc[st][r * TBE + tb];
#pragma unroll
          for (int e = 0; e < VW; ++e) a = fmaf(in ? w[r][u][e] : 0.f, pv[e], a);
          acc[st][r * TBE + tb] = a;
        }
      }
  }
}
template <int NSET>
__device__ __forceinline__ void enc_group_finish(float (&acc)[NSET][RB * TBE], int lane, int m0, int b0, const FoldK& k, const float* s_be, float* s_hid) {
  static_assert(RB * TBE == 16, "wave_sum16 reduces RB x TBE = 16 partial sums");
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const float v = wave_sum16(acc[st], lane);
    const int idx = (lane >> 2) & 15, r = idx / TBE, tb = st * TBE + (idx - r * TBE);
    const int mm = min(m0 + r, k.Hc - 1);
    const float hv = tanhf(v + s_be[mm]);
    if ((lane & 3) == 0 && m0 + r < k.Hc) {
      s_hid[tb * 64 + mm] = hv;
      if (b0 + tb < k.B) k.hid[(long long)(b0 + tb) * k.Hc + mm] = hv;
    }
  }
}
// ONE: every wave owns at most one group of RB rows and a row is one tile of 128 * IU columns (C*T <= 640, Hc <= 64: every reference
// shape with T <= 213) -- no loop, and the wave's tile of W_eff is requested BEFORE the observation rows are staged: it does not depend
// on them and both are cold misses (W_eff was written by the fold launch a moment ago, from other CUs): one round trip instead of two.
// BIGL (latent dim >= 32: the proc family's 50): the 2 * L * Hc head weights are staged in one batch of eight loads per thread and TRANSPOSED
// ([which][mm][l]: the head threads of consecutive latent dims read consecutive words), one lane per head output.
template <int TB, bool ONE, bool BIGL, int VW = 2, int NP = EncVec<VW>::IUV>
__global__ void __launch_bounds__(FNT) enc_fwd2_kernel(const float* __restrict__ pl_x, const float* __restrict__ pl_zloc_w, const float* __restrict__ pl_zls_w,
                                                       const float* __restrict__ pl_beff, const float* __restrict__ pl_weff, const FoldK k) {
  static_assert(TB % TBE == 0, "sets of four trajectories (wave_sum16 reduces RB x 4 partial sums)");
  constexpr int NSET = TB / TBE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, NT = blockDim.x, CT = k.CT, Hc = k.Hc, L = k.L;
  float* s_x = smem;                 // [TB][CT]   raw rows in memory order (= kappa order)
  float* s_hid = s_x + TB * CT;      // [TB][64]
  float* s_hw = s_hid + TB * 64;     // [2][L][Hc]
  float* s_be = s_hw + 2 * L * Hc;   // [64] b_eff
  float* s_hb = s_be + 64;           // [2][L] head biases
  const int b0 = blockIdx.x * TB;
  STAMP(8);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = NT >> 6;   // wave-uniform => row pointers in SGPRs
  const int ngroups = (Hc + RB - 1) / RB;
  typedef typename EncVec<VW>::T WV;
  constexpr int IUV = NP;
  static_assert(VW == 2 || ONE, "16-byte requests: the one-tile form only");
  WV w1[RB][IUV];
  if (ONE && wave < ngroups) {
#pragma unroll
    for (int u = 0; u < IUV; ++u) {
      const int ic = min(VW * lane + 64 * VW * u, CT - VW);
#pragma unroll
      for (int r = 0; r < RB; ++r) w1[r][u] = *reinterpret_cast<const WV*>(pl_weff + (long long)min(wave * RB + r, Hc - 1) * CT + ic);
    }
  }
  {
    // raw rows (the block's TB * CT floats are contiguous in memory: dense rows), head weights, b_eff, head biases: every global load of
    // the prologue is requested before the first LDS store (separate load-store loops are separate round trips)
    const long long base = (long long)b0 * CT, lim = (long long)k.B * CT - 1;
    constexpr int NHW = BIGL ? 8 : 2;   // 2 * L * Hc <= 8 * 1024 head weights: one batch (short latents: two loads, then a loop)
    static_assert(2 * SLODE_MAX_L * SLODE_MAX_HC <= 8 * FNT, "the head weights are staged in one batch of loads");
    constexpr int NXB = TB >= 16 ? 8 : 4;   // observation loads in flight per thread (16 trajectories x C*T: one batch of eight)
    float v[NXB], hw[NHW];
#pragma unroll
    for (int q = 0; q < NXB; ++q) v[q] = (q < 4 || q * NT < TB * CT) ? pl_x[min(base + tid + q * NT, lim)] : 0.f;
#pragma unroll
    for (int q = 0; q < NHW; ++q) {
      const int e = min(tid + q * NT, 2 * L * Hc - 1);
      hw[q] = (q * NT < 2 * L * Hc) ? ((e < L * Hc) ? pl_zloc_w[e] : pl_zls_w[e - L * Hc]) : 0.f;   // (kernel-uniform guard)
    }
    const float bev = pl_beff[min(tid, Hc - 1)];
    const int hbi = min(max(tid - 64, 0), 2 * L - 1);
    const float hbv = (hbi < L) ? k.zloc_b[hbi] : k.zls_b[hbi - L];
#pragma unroll
    for (int q = 0; q < NXB; ++q)
      if (tid + q * NT < TB * CT) s_x[tid + q * NT] = v[q];
#pragma unroll
    for (int q = 0; q < NHW; ++q) {
      const int e = tid + q * NT;
      if (e < 2 * L * Hc) {
        if (BIGL) {
          const int which = e / (L * Hc), r = e - which * (L * Hc), l = r / Hc, mm = r - l * Hc;
          s_hw[which * L * Hc + mm * L + l] = hw[q];
        } else {
          s_hw[e] = hw[q];
        }
      }
    }
    if (!BIGL)
      for (int e = tid + NHW * NT; e < 2 * L * Hc; e += NT) s_hw[e] = (e < L * Hc) ? pl_zloc_w[e] : pl_zls_w[e - L * Hc];
    if (tid < Hc) s_be[tid] = bev;
    if (tid >= 64 && tid < 64 + 2 * L) s_hb[tid - 64] = hbv;
    for (int e0 = tid + NXB * NT; e0 < TB * CT; e0 += 4 * NT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = pl_x[min(base + e0 + q * NT, lim)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e0 + q * NT < TB * CT) s_x[e0 + q * NT] = v[q];
    }
  }
  __syncthreads();
  STAMP(9);
  if (ONE) {
    if (wave < ngroups) {
      // the wave's tile of W_eff stays in registers while the block's sets of four trajectories pass by, one set at a time (16 partial
      // sums live, whatever TB is: large batches take 16 trajectories per block and stream W_eff a quarter as often)
#pragma unroll 1
      for (int st = 0; st < NSET; ++st) {
        float acc[1][RB * TBE];   // [r][tb] flattened: wave_sum16 reduces it in place
#pragma unroll
        for (int i = 0; i < RB * TBE; ++i) acc[0][i] = 0.f;
        enc_tile_fma<1, VW, NP>(w1, s_x + st * TBE * CT, CT, VW * lane, acc);
        __builtin_amdgcn_sched_barrier(0);   // keep the reduction's temporaries out of the load/FMA phase (spills otherwise)
        if (st == 0) STAMP(12);
        enc_group_finish<1>(acc, lane, wave * RB, b0 + st * TBE, k, s_be, s_hid + st * TBE * 64);
      }
      STAMP(13);
    }
  } else {
    for (int g = wave; g < ngroups; g += nw) {
      const int m0 = g * RB;
      float acc[NSET][RB * TBE];
#pragma unroll
      for (int st = 0; st < NSET; ++st)
#pragma unroll
        for (int i = 0; i < RB * TBE; ++i) acc[st][i] = 0.f;
      const float* wrow[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) wrow[r] = pl_weff + (long long)min(m0 + r, Hc - 1) * CT;
      // CT is even for C*T of every supported config; odd CT falls back to scalar columns
      if ((CT & 1) == 0) {
        for (int i0 = 2 * lane; i0 < CT; i0 += 128 * IU) {
          EncVec<2>::T w[RB][IU];
#pragma unroll
          for (int u = 0; u < IU; ++u) {
            const int ic = min(i0 + 128 * u, CT - 2);
#pragma unroll
            for (int r = 0; r < RB; ++r) w[r][u] = *reinterpret_cast<const EncVec<2>::T*>(wrow[r] + ic);
          }
          enc_tile_fma<NSET, 2>(w, s_x, CT, i0, acc);
        }
      } else {
        for (int i = lane; i < CT; i += 64) {
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const float w = wrow[r][i];
#pragma unroll
            for (int st = 0; st < NSET; ++st)
#pragma unroll
              for (int tb = 0; tb < TBE; ++tb) acc[st][r * TBE + tb] = fmaf(w, s_x[(st * TBE + tb) * CT + i], acc[st][r * TBE + tb]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reduction's temporaries out of the load/FMA phase (spills otherwise)
      STAMP(12);
      enc_group_finish<NSET>(acc, lane, m0, b0, k, s_be, s_hid);
      STAMP(13);
    }
  }
  __syncthreads();
  STAMP(10);
  // heads: LPO lanes per output (which, trajectory, latent dim; latent dim fastest), each summing every LPO-th hidden unit, then an xor
  // butterfly inside the lane group; LPO = 16 / 4 / 1, the largest that covers all outputs in one pass (short latents: 64 outputs x 16
  // lanes; L = 50: one lane each)
  if (BIGL) {
    // long latents: the two head layers are one [TB <= 16 trajectories x Hc] . [Hc x 2L] product on the f32 matrix cores
    // (v_mfma_f32_16x16x4_f32, exact fp32): wave w takes output columns 16 w .. 16 w + 15 of the 2 L, ceil(Hc / 4) k-steps, one LDS read per
    // operand and k-step (the transposed head weights make the B operand's 16 lanes consecutive words).  A[m][k]: lane (m = lane & 15,
    // k = lane >> 4); B[k][n]: lane (n = lane & 15, k = lane >> 4); D[m][n]: register r of lane l is (m = 4 (l >> 4) + r, n = l & 15).
    static_assert(!BIGL || TB <= 16, "one 16-row tile of trajectories");
    typedef __attribute__((ext_vector_type(4))) float f32x4_t;
    const int n_cols = 2 * L, n_tiles = (n_cols + 15) >> 4;
    for (int tile = wave; tile < n_tiles; tile += nw) {
      const int mrow = lane & 15, kq = lane >> 4;
      const int n = min(tile * 16 + (lane & 15), n_cols - 1), which = n / L, l = n - which * L;
      const float* Wn = s_hw + which * L * Hc + l;
      f32x4_t d = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < (Hc + 3) >> 2; ++ks) {
        const int mm = 4 * ks + kq, mc = min(mm, Hc - 1);
        const float a = (mm < Hc && mrow < TB) ? s_hid[min(mrow, TB - 1) * 64 + mc] : 0.f;
        const float bq = (mm < Hc) ? Wn[mc * L] : 0.f;
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, d, 0, 0, 0);
      }
      const float bias = s_hb[which * L + l];
      if (tile * 16 + (lane & 15) < n_cols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tb = 4 * (lane >> 4) + r;
          if (tb < TB && b0 + tb < k.B) {
            const float acc = d[r] + bias;
            if (which) k.scale[(long long)(b0 + tb) * L + l] = expf(acc);
            else k.loc[(long long)(b0 + tb) * L + l] = acc;
          }
        }
      }
    }
  } else {
    const int n_out = TB * L * 2;
    const int lsh = (n_out * 16 <= NT) ? 4 : ((n_out * 4 <= NT) ? 2 : 0), lpo = 1 << lsh;   // kernel-uniform
    for (int e0 = (tid >> lsh); e0 < ((n_out + (64 >> lsh) - 1) & ~((64 >> lsh) - 1)); e0 += NT >> lsh) {   // (wave-uniform trip count)
      const int e = min(e0, n_out - 1), lg = tid & (lpo - 1);
      const int which = e / (TB * L), r = e - which * (TB * L);
      const int tb = r / L, l = r - tb * L;
      const float* hd = s_hid + tb * 64;
      float acc;
      if (BIGL) {
        const float* W = s_hw + which * L * Hc + l;
        float a0 = 0.f, a1 = 0.f;
        int mm = lg;
        for (; mm + lpo < Hc; mm += 2 * lpo) {
          a0 = fmaf(W[mm * L], hd[mm], a0);
          a1 = fmaf(W[(mm + lpo) * L], hd[mm + lpo], a1);
        }
        if (mm < Hc) a0 = fmaf(W[mm * L], hd[mm], a0);
        acc = a0 + a1;
      } else {
        const float* W = s_hw + which * L * Hc + l * Hc;
        acc = 0.f;
        for (int mm = lg; mm < Hc; mm += lpo) acc = fmaf(W[mm], hd[mm], acc);
      }
      if (lsh >= 4) acc = row16_sum(acc);   // (DPP: the 16 lanes of an output are one row; the quads of the 4-lane form likewise)
      else if (lsh >= 2) { acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); }
      acc += s_hb[which * L + l];
      if (lg == 0 && e0 < n_out && b0 + tb < k.B) {
        if (which) k.scale[(long long)(b0 + tb) * L + l] = expf(acc);
        else k.loc[(long long)(b0 + tb) * L + l] = acc;
      }
    }
  }
  STAMP(11);
}

// ---- backward part 3: chain rule from G = dLoss/dW_eff back to lin.weight (final) and conv.{weight,bias} (per-m partials) ---
// One 640-thread workgroup per (hidden unit m, pair of conv filters): Hc * ceil(F / 2) = 250 workgroups at the reference's shapes -- one
// per CU (round 2: one 1024-thread workgroup per m, 50 of them, whose four LDS-issue-bound stages ran one after the other: 15.6 us).
// G's columns [0, CT) are dLoss/dW_eff[m][:], column CT is g_beff[m] = sum_b g_pre[b][m] (the MFMA GEMM appends a ones-column to X),
// both summed here over the split-K partials in fixed order (every workgroup of an m sums the row again: L2 hits).
constexpr int CNT = 640;    // threads of a chain / rider block
constexpr int CNT1 = 384;   // ... of which run contraction (i); the other 256 run (ii)
constexpr int FPC = 2;      // conv filters per chain block
constexpr int QCH = 32;     // q-chunks of (ii): the half-wave of an (f, c) pair
constexpr int PERM = 8;     // weights a lane of (ii) holds in registers at a time
constexpr int NPA = 2;      // Adam passes of a block's lin.weight piece whose state is fetched in the prologue
static_assert(4 * (FPC * SLODE_MAX_C * SLODE_MAX_K + FPC) <= CNT, "four lanes per conv element in the last block of a filter pair");
static_assert(SLODE_MAX_HC <= 64, "the last block holds ceil(Hc / 4) <= 16 row values per lane");
// The same launch finishes the whole flat gradient: rider blocks (blockIdx >= Hc * NFC) reduce everything that does not depend on this
// kernel (ODE half, lin.bias, head layers, loss), the chain blocks apply Adam to their own piece of the lin.weight row, and -- per filter
// pair -- the LAST chain block to arrive (agent-scope counter, one per pair, 128 B apart) sums the Hc conv rows of the pair's taps.
// Fixed order => reproducible.
// FN: the in-launch fold form (FOLD-NEXT, a measured arm: SLODE_FOLD_NEXT=1) -- a separate instantiation, so that the shipped form carries
// none of its registers (with the fold section compiled into the one kernel: 87 -> 104 VGPRs, 87 -> 122 SGPR spills, +2 us at T = 300)
template <int C, int JM, bool FN = false>
__global__ void __launch_bounds__(CNT) enc_chain_kernel(const float* __restrict__ pl_gslabs, const float* __restrict__ pl_lin_w, const float* __restrict__ pl_wprime,
                                                        const FoldK k, const TailK tl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int NFC = (k.F + FPC - 1) / FPC, n_chain = k.Hc * NFC;
  if ((int)blockIdx.x >= n_chain) {   // rider block
    // (normally one element per rider thread; the in-launch fold form may run fewer, looping riders so that the whole grid is resident)
    const int r = (int)blockIdx.x - n_chain, n_rid = (int)gridDim.x - n_chain;
    if (r == 0) STAMP_ANY(26);
    for (int i = tl.lin_b + r * CNT + tid; i <= tl.n_total; i += n_rid * CNT) {
      if (i < tl.n_total) tail_element<FN>(tl, i);
      else tail_loss(tl);
    }
    if (r == 0) STAMP_ANY(27);
    if (FN) {   // this block's new parameter values (lin.bias: stored agent-scope) have reached L2; count it in
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(tl.done + 32 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int m = (int)blockIdx.x % k.Hc, fci = (int)blockIdx.x / k.Hc, f0 = fci * FPC, nf = min(FPC, k.F - f0);
  const int CT = k.CT, J = k.J, K = k.K, F = k.F, n_pool = k.n_pool, FQ = k.FQ, T = k.T, GN = CT + 1;
  const int n_lw = nf * n_pool;                          // this block's piece of the lin.weight row: [f0 * n_pool, f0 * n_pool + n_lw)
  float* s_G = smem;                                     // [CT + 1]
  float* s_wl = s_G + ((GN + 3) & ~3);                   // [FPC * n_pool]   lin.weight[m][f0 * n_pool ...]
  float* s_wp = s_wl + ((FPC * n_pool + 3) & ~3);        // [FPC][C][JM]     w' (zero padded)
  float* s_pw = s_wp + FPC * C * JM;                     // [FPC][C][JM]     dLoss/dw' of this m
  float* s_glw = s_pw + FPC * C * JM;                    // [FPC * n_pool]   this piece's lin.weight gradient (for the fused Adam pass)
  STAMP(16);
  // Adam state of this block's lin.weight elements: fetched first, used last (a cold miss off the block's tail)
  const int lw0 = tl.lin_w + m * FQ + f0 * n_pool;
  float am[NPA], av[NPA], ap[NPA];
#pragma unroll
  for (int u = 0; u < NPA; ++u) {
    const int i = lw0 + min(tid + u * CNT, n_lw - 1);
    const bool on = tl.ad.p != nullptr && tid + u * CNT < n_lw;
    am[u] = on ? tl.ad.m[i] : 0.f; av[u] = on ? tl.ad.v[i] : 0.f; ap[u] = on ? tl.ad.p[i] : 0.f;
  }
  const float rs = k.rowsum[m * F + f0 + min(tid, nf - 1)];   // (for the conv.bias column: requested here, used behind the contractions)
  // split-K partials of row m: all of a column's loads in flight at once (a loop of load-then-add costs one L2 / HBM round trip per
  // partial: the GEMM launch wrote them from other CUs), summed in fixed order
  {
    // (row sums, the lin.weight piece and w' are requested together and only then stored: three loops would be three round trips)
    const float g0 = strided_sum(pl_gslabs + (long long)m * GN + min(tid, GN - 1), k.Hc * GN, k.n_gslabs);
    const float wl0 = pl_lin_w[(long long)m * FQ + f0 * n_pool + min(tid, n_lw - 1)];
    const int i = min(tid, nf * C * JM - 1), j = i % JM, fc = i / JM;
    const float wp0 = (j < J) ? pl_wprime[(f0 * C + fc) * J + min(j, J - 1)] : 0.f;
    if (tid < GN) s_G[tid] = g0;
    if (tid < n_lw) s_wl[tid] = wl0;
    if (tid < nf * C * JM) s_wp[tid] = wp0;
  }
  for (int i = tid + CNT; i < GN; i += CNT) s_G[i] = strided_sum(pl_gslabs + (long long)m * GN + i, k.Hc * GN, k.n_gslabs);
  for (int i = tid + CNT; i < n_lw; i += CNT) s_wl[i] = pl_lin_w[(long long)m * FQ + f0 * n_pool + i];
  static_assert(FPC * SLODE_MAX_C * (SLODE_MAX_K + SLODE_MAX_P) <= CNT, "w' of a filter pair is staged in one pass");
  __syncthreads();
  STAMP(17);
  const float gb = s_G[CT];
  const int sC = k.t_major ? 1 : T, sT = k.t_major ? C : 1;   // kappa(c, t) = c*sC + t*sT
  // The two contractions are independent: waves 0..5 run (i) while waves 6..9 run (ii).
  if (tid < CNT1) {
    // (i) dLoss/d lin.weight[m][f*n_pool + q] = g_beff[m]*conv_b[f] + sum_{c,j} G[kappa(c, q+j)] * w'[f][c][j]; thread (f, q)
    for (int e = tid; e < n_lw; e += CNT1) {
      const int fl = e / n_pool, q = e - fl * n_pool;
      float acc = gb * k.conv_b[f0 + fl];
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int j = 0; j < JM; ++j)   // taps beyond J meet w' == 0
          acc = fmaf(s_G[c * sC + min(q + j, T - 1) * sT], s_wp[(fl * C + c) * JM + j], acc);
      k.g_lin_w[(long long)m * FQ + f0 * n_pool + e] = acc;
      s_glw[e] = acc;
    }
    STAMP(18);
  } else {
    // (ii) dLoss/dw'[f][c][j] (this m) = sum_q G[kappa(c, q+j)] * lin.weight[m][f*n_pool + q]: the 32 lanes of a half-wave share (f, c) and
    // split q; the half-wave's partial sums meet in a fixed-order reduction
    const int per = (n_pool + QCH - 1) / QCH;
    for (int e0 = tid - CNT1; e0 < ((nf * C * QCH + 63) & ~63); e0 += CNT - CNT1) {   // wave-uniform trip count
      const bool valid = e0 < nf * C * QCH;
      const int e = valid ? e0 : 0;
      const int fc = e / QCH, ch = e - fc * QCH, fl = fc / C, c = fc - fl * C;
      const int q0 = min(ch * per, n_pool), q1 = valid ? min(n_pool, q0 + per) : q0;
      const float* wl = s_wl + fl * n_pool;
      const float* gp = s_G + c * sC;
      float acc[JM];
#pragma unroll
      for (int j = 0; j < JM; ++j) acc[j] = 0.f;
      // PERM weights and the PERM + JM - 1 samples of G they meet, in registers: PERM * JM FMAs for 2 * PERM + JM - 1 LDS reads, no window to shift
      for (int qq = q0; qq < q1; qq += PERM) {
        float wv[PERM], gv[PERM + JM - 1];
#pragma unroll
        for (int i = 0; i < PERM; ++i) wv[i] = wl[min(qq + i, n_pool - 1)];
#pragma unroll
        for (int i = 0; i < PERM + JM - 1; ++i) gv[i] = gp[min(qq + i, T - 1) * sT];
#pragma unroll
        for (int i = 0; i < PERM; ++i) wv[i] = (qq + i < q1) ? wv[i] : 0.f;
#pragma unroll
        for (int j = 0; j < JM; ++j)
#pragma unroll
          for (int i = 0; i < PERM; ++i) acc[j] = fmaf(gv[i + j], wv[i], acc[j]);
      }
      // half-wave sum: four DPP adds inside the 16-lane rows, then the two rows of the half-wave (v_permlane16_swap)
#pragma unroll
      for (int j = 0; j < JM; ++j) acc[j] = half_wave_sum(acc[j]);
      if (valid && ch == 0) {
#pragma unroll
        for (int j = 0; j < JM; ++j) s_pw[fc * JM + j] = (j < J) ? acc[j] : 0.f;
      }
    }
  }
  __syncthreads();
  STAMP(19);
  // w' -> conv taps (adjoint of the box filter) and conv.bias; this block's columns of slab row m, stored agent-scope (sc1, write-through)
  float* row = k.conv_slabs + (long long)m * (F * C * K + F);
  const float fP = (float)k.P;
  for (int e = tid; e < nf * C * K; e += CNT) {
    const int kk = e % K, fc = e / K;
    float s = 0.f;
    for (int p = 0; p < k.P; ++p) s += s_pw[fc * JM + kk + p];
    __hip_atomic_store(row + f0 * C * K + e, s / fP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the pair's last block
  }
  if (tid < nf) __hip_atomic_store(row + F * C * K + f0 + tid, gb * rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  STAMP(20);
  // Hand-off of the conv rows to the pair's last block to arrive, without a cache-wide release: the rows are stored with agent-scope (sc1,
  // write-through) stores; EVERY storing wave drains them (explicit s_waitcnt vmcnt(0) below: the compiler does not emit one for a
  // store it has no later use for) ahead of the barrier behind which one lane bumps the pair's agent-scope counter; the block that sees
  // the final count reads the rows with sc1 (L1-bypassing) loads behind the next barrier -- every byte was stored sc1 and drained
  // before its block's counter add, and the add's returned value is what told this block it is last (MI355X_MICROARCH.md, hand-offs
  // measured with sc1 loads in place of the acquire: 4-byte stores and loads, one counter; tests/test_host_cpu.py pins the sc1 bits).
  __shared__ int s_last;
  // conv element of this lane group as part of the last block (4 lanes per element): its Adam state is fetched now, off the critical path
  const int n_el = nf * C * K + nf, el = tid >> 2, part = tid & 3;
  const bool el_on = el < n_el;
  const int col = el < nf * C * K ? f0 * C * K + el : F * C * K + f0 + (el - nf * C * K);   // column of the conv slab rows
  const int ci = tl.conv_w + min(col, F * C * K + F - 1);
  const bool c_on = el_on && part == 0 && tl.ad.p != nullptr;
  const float cm = c_on ? tl.ad.m[ci] : 0.f, cv = c_on ? tl.ad.v[ci] : 0.f, cp = c_on ? tl.ad.p[ci] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 row stores have reached L2 before the counter can move
  __syncthreads();
  if (tid == 0)
    s_last = (__hip_atomic_fetch_add(tl.counter + 32 * fci, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(k.Hc - 1)) ? 1 : 0;
  STAMP(21);
  __syncthreads();
  const bool last = s_last != 0;
  if (last) STAMP_ANY(24);
  // the pair's counter goes back to zero for the next launch on this workspace (the last block is the last to touch it: a step that finds
  // the fold of the current weights already in the workspace has no fold launch to zero it)
  if (last && tid == 0) tl.counter[32 * fci] = 0u;
  // the last block to arrive is the one every other block's work waits behind: its loads of the Hc conv rows (4 lanes per element, each
  // summing every 4th row in fixed order; 16 sc1 loads in flight per lane) go out now and fly during its own piece's Adam pass
  constexpr int NRV = 16;
  float rv[NRV];
  const int nrow = (k.Hc - part + 3) / 4;
  if (last && el_on) {
    const float* src = tl.conv_slabs + (long long)part * tl.n_cv + col;
#pragma unroll
    for (int q = 0; q < NRV; ++q)
      rv[q] = __hip_atomic_load(src + (long long)min(q, nrow - 1) * 4 * tl.n_cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tl.ad.p != nullptr) {   // Adam on this block's piece of the lin.weight row
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
      const int e = tid + u * CNT;
      if (e < n_lw) {
        const float g = s_glw[e];
        const float mi = am[u] + tl.ad.one_minus_b1 * (g - am[u]);
        const float vi = av[u] * tl.ad.b2 + tl.ad.one_minus_b2 * g * g;
        const float pn = ap[u] - tl.ad.step_size * (mi / (sqrtf(vi) / tl.ad.sqrt_bc2 + tl.ad.eps));
        if (FN) __hip_atomic_store(tl.ad.p + lw0 + e, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by the row's other blocks)
        else tl.ad.p[lw0 + e] = pn;
        tl.ad.m[lw0 + e] = mi;
        tl.ad.v[lw0 + e] = vi;
      }
    }
    for (int e = tid + NPA * CNT; e < n_lw; e += CNT) {   // (rows longer than the prologue fetched state for)
      const int i = lw0 + e;
      const float g = s_glw[e];
      float mi = tl.ad.m[i], vi = tl.ad.v[i];
      mi = mi + tl.ad.one_minus_b1 * (g - mi);
      vi = vi * tl.ad.b2 + tl.ad.one_minus_b2 * g * g;
      const float pn = tl.ad.p[i] - tl.ad.step_size * (mi / (sqrtf(vi) / tl.ad.sqrt_bc2 + tl.ad.eps));
      if (FN) __hip_atomic_store(tl.ad.p + i, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else tl.ad.p[i] = pn;
      tl.ad.m[i] = mi;
      tl.ad.v[i] = vi;
    }
  }
  STAMP(22);
  if (FN) {   // this block's piece of the new lin.weight row (stored agent-scope above) has reached L2: count it in for row m
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(tl.done + 32 * (16 + m), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (last) {   // conv.weight, conv.bias of this filter pair
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (el_on) {
#pragma unroll
      for (int q = 0; q < NRV; q += 4) {
        a0 += (q < nrow) ? rv[q] : 0.f;
        a1 += (q + 1 < nrow) ? rv[q + 1] : 0.f;
        a2 += (q + 2 < nrow) ? rv[q + 2] : 0.f;
        a3 += (q + 3 < nrow) ? rv[q + 3] : 0.f;
      }
    }
    float g = (a0 + a1) + (a2 + a3);
    g += dpp_f<0xB1>(g);   // the element's four lanes are one quad: quad_perm [1,0,3,2], [2,3,0,1]
    g += dpp_f<0x4E>(g);
    if (el_on && part == 0) {
      tl.grads[ci] = g;
      if (tl.ad.p) {   // adam_apply with the prefetched state
        const float mi = cm + tl.ad.one_minus_b1 * (g - cm);
        const float vi = cv * tl.ad.b2 + tl.ad.one_minus_b2 * g * g;
        const float pn = cp - tl.ad.step_size * (mi / (sqrtf(vi) / tl.ad.sqrt_bc2 + tl.ad.eps));
        if (FN) __hip_atomic_store(tl.ad.p + ci, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by every block's fold)
        else tl.ad.p[ci] = pn;
        tl.ad.m[ci] = mi;
        tl.ad.v[ci] = vi;
      }
    }
    STAMP_ANY(25);
  }
  if (!FN) return;
  // ---- FOLD-NEXT: W_eff, rowsum, b_eff, w' of the UPDATED weights, for the next step (which then has no fold launch) ---------------------
  // Every parameter the fold reads was stored agent-scope (sc1) by its owner in this launch: this block's lin.weight piece above, the conv
  // taps / biases by the pairs' last blocks, lin.bias by the rider blocks.  Hand-off as for the conv rows: every storing wave drains
  // (s_waitcnt vmcnt(0)), barrier, ONE lane adds to the launch's arrival counter; ONE wave polls it with sc1 loads until every block of
  // the launch has arrived (all of them are resident: the launcher enables this form only then), barrier, sc1 loads of the data.
  // Arrivals (all generation-based, never reset: gen = the handle's count of such launches on this workspace): row m's NFC blocks on
  // slot 16 + m (above), the NFC last blocks of the filter pairs on slot 8 (here, behind their conv Adam), the rider blocks on slot 9.
  // No counter sees more than a handful of adds: a single launch-wide counter put 255 read-modify-writes behind 250 pollers of the same
  // line (chain launch 8.5 -> 17.4 us, profiles/r04_c_ab7_fold_next_single_counter.log).
  if (last) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the pair whose add completes the count raises a FLAG on a line of its own, slot 10: the 250 waiting blocks poll that line, which is
    //  written once -- polling the counter itself puts the five adds behind the pollers' reads of the same line)
    if (tid == 0 && __hip_atomic_fetch_add(tl.done + 32 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == tl.done_target * (unsigned int)NFC)
      __hip_atomic_store(tl.done + 32 * 10, tl.done_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ int s_fold_ok;
  float* s_row = s_glw + ((FPC * n_pool + 3) & ~3);   // [F][n_pool] row m of the new lin.weight
  float* s_wpf = s_row + ((FQ + 3) & ~3);              // [F][C][JM]  new w'
  const unsigned int gen = tl.done_target, t_row = gen * (unsigned int)NFC, t_rid = gen * (unsigned int)tl.n_riders;
  // bounded waits: the launcher only enables this form when the occupancy query says every block of the launch is resident at once; should
  // that ever not hold, a wait ends after ~0.3 s and the row is written as NaN (a loud NaN loss in the next step, not a hung GPU)
  if (tid == 0) {   // (1) the row's other blocks: their pieces of the new lin.weight row
    int ok = 0;
    for (int it = 0; it < (1 << 18); ++it) {
      if ((int)(__hip_atomic_load(tl.done + 32 * (16 + m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - t_row) >= 0) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    s_fold_ok = ok;
  }
  __syncthreads();
  // the row is requested now (sc1) and flies while the block waits for the conv taps of the slowest filter pair
  const float* src = tl.ad.p + tl.lin_w + (long long)m * FQ;
  constexpr int NRQ = 4;
  float rv4[NRQ];
#pragma unroll
  for (int q = 0; q < NRQ; ++q) rv4[q] = __hip_atomic_load(src + min(tid + q * CNT, FQ - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0 && s_fold_ok) {   // (2) every pair's conv taps / biases, and -- the row's first block -- the riders' lin.bias
    int ok = 0;
    for (int it = 0; it < (1 << 18); ++it) {
      const unsigned int b = __hip_atomic_load(tl.done + 32 * 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int c = fci == 0 ? __hip_atomic_load(tl.done + 32 * 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t_rid;
      if ((int)(b - gen) >= 0 && (int)(c - t_rid) >= 0) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    s_fold_ok = ok;
  }
  __syncthreads();
  if (!s_fold_ok) {
    const int per = (CT + NFC - 1) / NFC, k0 = fci * per, k1 = min(CT, k0 + per);
    for (int kap = k0 + tid; kap < k1; kap += CNT) k.weff[(long long)m * CT + kap] = __builtin_nanf("");
    return;
  }
  STAMP(28);
  {
    // the conv taps under this thread's w' entry (sc1), then the stores of both
    const int n_wp = F * C * JM;
    const int e = min(tid, n_wp - 1), j = e % JM, fc = e / JM;
    float tp[SLODE_MAX_P];
#pragma unroll
    for (int p = 0; p < SLODE_MAX_P; ++p) {
      const int kk = j - p;
      const bool on = p < k.P && j < J && kk >= 0 && kk < K;
      tp[p] = on ? __hip_atomic_load(tl.ad.p + tl.conv_w + fc * K + min(max(kk, 0), K - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NRQ; ++q)
      if (tid + q * CNT < FQ) s_row[tid + q * CNT] = rv4[q];
    for (int i = tid + NRQ * CNT; i < FQ; i += CNT) s_row[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < n_wp) {
      float sw = 0.f;
#pragma unroll
      for (int p = 0; p < SLODE_MAX_P; ++p) sw += tp[p];
      sw = sw / fP;
      s_wpf[tid] = sw;
      if (blockIdx.x == 0 && j < J) k.wprime[fc * J + j] = sw;
    }
    for (int e2 = tid + CNT; e2 < n_wp; e2 += CNT) {   // (more entries than threads: long-tap shapes)
      const int j2 = e2 % JM, fc2 = e2 / JM;
      float sw = 0.f;
      for (int p = 0; p < k.P; ++p) {
        const int kk = j2 - p;
        if (j2 < J && kk >= 0 && kk < K) sw += __hip_atomic_load(tl.ad.p + tl.conv_w + fc2 * K + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sw = sw / fP;
      s_wpf[e2] = sw;
      if (blockIdx.x == 0 && j2 < J) k.wprime[fc2 * J + j2] = sw;
    }
  }
  __syncthreads();
  STAMP(29);
  {   // this block's share of row m: kappas [fci * per, (fci + 1) * per), four lanes per output exactly as weff_kernel
    const int per = (CT + NFC - 1) / NFC, k0 = fci * per, k1 = min(CT, k0 + per);
    for (int el = tid >> 2; k0 + el < k1; el += CNT / 4) {
      const int kap = k0 + el, fh = tid & 3;
      int c, t;
      if (k.t_major) { t = kap / C; c = kap - t * C; } else { c = kap / T; t = kap - c * T; }
      const float r = weff_out<JM>(k, s_row, s_wpf, c, t, fh);
      if (fh == 0) k.weff[(long long)m * CT + kap] = r;
    }
  }
  if (fci == 0 && tid < 64) {   // the row's first block: rowsum[m][:] and b_eff[m] with the new conv.bias / lin.bias (sc1: other blocks' stores)
    float* s_cb = s_wpf + F * C * JM;   // [F] new conv.bias
    if (tid < F) s_cb[tid] = __hip_atomic_load(tl.ad.p + tl.conv_w + F * C * K + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float lb = __hip_atomic_load(tl.ad.p + tl.lin_b + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rowsum_beff_wave(k, s_row, m, tid, s_cb, lb);   // (same wave: its LDS stores above are read in program order)
  }
  STAMP(30);
}

FoldK make_foldk(const FoldLaunch& a) {
  const slode_shape& s = a.s;
  const slode_layout& lay = a.lay;
  const float* p = a.params;
  FoldK k{};
  k.B = s.B; k.T = s.T; k.C = s.C; k.L = s.L; k.F = s.F; k.K = s.K; k.P = s.P; k.Hc = s.Hc;
  k.n_conv = s.T - s.K + 1; k.n_pool = k.n_conv - s.P + 1; k.FQ = s.F * k.n_pool; k.CT = s.C * s.T; k.J = s.K + s.P - 1;
  k.t_major = a.t_major;
  k.conv_w = p + lay.conv_w; k.conv_b = p + lay.conv_b; k.lin_w = p + lay.lin_w; k.lin_b = p + lay.lin_b;
  k.zloc_w = p + lay.zloc_w; k.zloc_b = p + lay.zloc_b; k.zls_w = p + lay.zls_w; k.zls_b = p + lay.zls_b;
  k.x = a.x; k.weff = a.weff; k.rowsum = a.rowsum; k.wprime = a.wprime; k.beff = a.beff;
  k.loc = a.loc; k.scale = a.scale; k.hid = a.hid;
  k.scale_in = a.scale; k.hid_in = a.hid; k.g_loc = a.g_loc; k.g_scale = a.g_scale;
  k.g_pre = a.g_pre; k.slabs = a.small_slabs; k.small_stride = a.small_stride;
  k.gslabs = a.gslabs; k.n_gslabs = a.n_gslabs; k.g_lin_w = a.g_lin_w; k.conv_slabs = a.conv_slabs; k.counter = a.counter;
  k.cstd = p + lay.cstd; k.sigtab = a.sigtab; k.gauss = s.likelihood == SLODE_GAUSS ? 1 : 0;
  return k;
}

}  // namespace

int slode_fold_small_count(const slode_shape& s) { return s.Hc + 2 * (s.L * s.Hc + s.L); }
int slode_chain_blocks(const slode_shape& s, int n_total, int lin_b, int* n_chain) {
  *n_chain = s.Hc * ((s.F + FPC - 1) / FPC);
  return *n_chain + (n_total + 1 - lin_b + CNT - 1) / CNT;
}
static size_t chain_lds(const slode_shape& s, bool fold_next) {
  const int n_pool = s.T - s.K + 1 - s.P + 1, J = s.K + s.P - 1, JM = J <= 14 ? 14 : SLODE_MAX_K + SLODE_MAX_P;
  return sizeof(float) * ((size_t)s.C * s.T + 8 + 2 * ((size_t)FPC * n_pool + 4) + 2 * (size_t)FPC * s.C * JM +
                          (fold_next ? (size_t)s.F * n_pool + 4 + (size_t)s.F * s.C * JM + SLODE_MAX_F : 0));
}
// how many blocks of the chain launch (in-launch fold form) the device holds AT ONCE: the blocks of that form wait for each other, so the
// form is only used when this covers the whole grid.  The runtime's occupancy query for the instantiation the launch would take.
int slode_chain_resident_blocks(const slode_shape& s, int num_cu) {
  const int J = s.K + s.P - 1, JM = J <= 14 ? 14 : SLODE_MAX_K + SLODE_MAX_P;
  const size_t lds = chain_lds(s, true);
  int per_cu = 0;
  hipError_t e = hipErrorInvalidValue;
  if (s.C == 3 && JM == 14) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (enc_chain_kernel<3, 14, true>), CNT, lds);
  else if (s.C == 4 && JM == 14) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (enc_chain_kernel<4, 14, true>), CNT, lds);
  else if (s.C == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (enc_chain_kernel<3, SLODE_MAX_K + SLODE_MAX_P, true>), CNT, lds);
  else if (s.C == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (enc_chain_kernel<4, SLODE_MAX_K + SLODE_MAX_P, true>), CNT, lds);
  if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; }
  return per_cu * num_cu;
}

hipError_t slode_launch_fold_fwd(const FoldLaunch& a, hipStream_t stream) {
  FoldK k = make_foldk(a);
  const int nb_w = (k.Hc * k.CT + WPB - 1) / WPB, nb_r = (k.Hc + WNT / 64 - 1) / (WNT / 64) + (k.sigtab ? (k.CT + WNT - 1) / WNT : 0);
  // dynamic LDS: (when they fit) the lin.weight rows a block's WPB consecutive outputs can touch
  const size_t max_rows = (size_t)(WPB - 1) / k.CT + 2;   // WPB outputs starting anywhere inside a row of CT
  const int stage_rows = (max_rows * (size_t)k.FQ) * sizeof(float) <= 96 * 1024 ? 1 : 0;
  const size_t wlds = sizeof(float) * (stage_rows ? max_rows * (size_t)k.FQ : 4);
  if (a.fold_skip) {
    // (the previous step's chain launch folded the current weights: encoder forward only)
  } else if (k.J <= 14) {
    if (wlds > 48 * 1024) (void)hipFuncSetAttribute((const void*)weff_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
    SLODE_LAUNCH("weff", (weff_kernel<14>), dim3(nb_w + nb_r), dim3(WNT), wlds, stream, k.conv_w, k.lin_w, k, stage_rows);
  } else {
    if (wlds > 48 * 1024) (void)hipFuncSetAttribute((const void*)weff_kernel<SLODE_MAX_K + SLODE_MAX_P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
    SLODE_LAUNCH("weff", (weff_kernel<SLODE_MAX_K + SLODE_MAX_P>), dim3(nb_w + nb_r), dim3(WNT), wlds, stream, k.conv_w, k.lin_w, k, stage_rows);
  }
  if (a.skip_enc) return hipGetLastError();
  const bool one = (k.CT & 1) == 0 && k.CT <= 128 * IU && (k.Hc + RB - 1) / RB <= FNT / 64;
  // 16-byte requests where the rows allow them (C*T a multiple of 4, at most 3 x 256 columns; W_eff and the dense observation rows are 16-byte aligned then)
  const bool wide = one && (k.CT & 3) == 0 && k.CT <= 256 * EncVec<4>::IUV && (reinterpret_cast<uintptr_t>(k.weff) & 15) == 0;
  // (a batch that fills the chip several times over: fewer, fatter blocks -- one block per CU at B = 4096)
  const int TB = (one && k.B >= 4096) ? 4 * TBE : (k.B >= 2048 ? 2 * TBE : TBE);
  const size_t lds = sizeof(float) * ((size_t)TB * k.CT + TB * 64 + 2 * (size_t)k.L * k.Hc + 64 + 2 * (size_t)k.L);
#define SLODE_ENC_FWD2(TT, OO, LL)                                                                                                      \
  do {                                                                                                                                \
    (void)hipFuncSetAttribute((const void*)enc_fwd2_kernel<TT, OO, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    SLODE_LAUNCH("enc_fwd2", (enc_fwd2_kernel<TT, OO, LL>), dim3((k.B + TB - 1) / TB), dim3(FNT), lds, stream, k.x, k.zloc_w, k.zls_w, k.beff, k.weff, k); \
  } while (0)
#define SLODE_ENC_FWD2_L(TT, OO) do { if (k.L >= 32) SLODE_ENC_FWD2(TT, OO, true); else SLODE_ENC_FWD2(TT, OO, false); } while (0)
#define SLODE_ENC_FWD2_W(TT, LL, PP)                                                                                                    \
  do {                                                                                                                                \
    (void)hipFuncSetAttribute((const void*)enc_fwd2_kernel<TT, true, LL, 4, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    SLODE_LAUNCH("enc_fwd2", (enc_fwd2_kernel<TT, true, LL, 4, PP>), dim3((k.B + TB - 1) / TB), dim3(FNT), lds, stream, k.x, k.zloc_w, k.zls_w, k.beff, k.weff, k); \
  } while (0)
#define SLODE_ENC_FWD2_WP(TT, LL) do { if (k.CT <= 512) SLODE_ENC_FWD2_W(TT, LL, 2); else SLODE_ENC_FWD2_W(TT, LL, 3); } while (0)
#define SLODE_ENC_FWD2_WL(TT) do { if (k.L >= 32) SLODE_ENC_FWD2_WP(TT, true); else SLODE_ENC_FWD2_WP(TT, false); } while (0)
  if (wide && TB == 4 * TBE) SLODE_ENC_FWD2_WL(4 * TBE);
  else if (wide && TB == TBE) SLODE_ENC_FWD2_WL(TBE);
  else if (wide) SLODE_ENC_FWD2_WL(2 * TBE);
  else if (TB == 4 * TBE) SLODE_ENC_FWD2_L(4 * TBE, true);
  else if (TB == TBE) { if (one) SLODE_ENC_FWD2_L(TBE, true); else SLODE_ENC_FWD2_L(TBE, false); }
  else { if (one) SLODE_ENC_FWD2_L(2 * TBE, true); else SLODE_ENC_FWD2_L(2 * TBE, false); }
  return hipGetLastError();
}

hipError_t slode_launch_fold_chain(const FoldLaunch& a, hipStream_t stream) {
  if (!a.tail) return hipErrorInvalidValue;   // the chain launch always finishes the flat gradient (its only caller is the fused tail)
  FoldK k = make_foldk(a);
  const int JM = k.J <= 14 ? 14 : SLODE_MAX_K + SLODE_MAX_P;
  const TailK tl = *a.tail;
  const size_t lds = chain_lds(a.s, tl.fold_next != 0);
  const int riders = tl.n_riders > 0 ? tl.n_riders : (tl.n_total + 1 - tl.lin_b + CNT - 1) / CNT;
  const dim3 grid(k.Hc * ((k.F + FPC - 1) / FPC) + riders);
#define SLODE_CHAIN(CC, JJ)                                                                                          \
  do {                                                                                                               \
    if (tl.fold_next) {                                                                                             \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)enc_chain_kernel<CC, JJ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      SLODE_LAUNCH("enc_chain", (enc_chain_kernel<CC, JJ, true>), grid, dim3(CNT), lds, stream, k.gslabs, k.lin_w, (const float*)k.wprime, k, tl);          \
    } else {                                                                                                         \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)enc_chain_kernel<CC, JJ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      SLODE_LAUNCH("enc_chain", (enc_chain_kernel<CC, JJ, false>), grid, dim3(CNT), lds, stream, k.gslabs, k.lin_w, (const float*)k.wprime, k, tl);         \
    }                                                                                                                \
  } while (0)
  if (k.C == 3 && JM == 14) SLODE_CHAIN(3, 14);
  else if (k.C == 4 && JM == 14) SLODE_CHAIN(4, 14);
  else if (k.C == 3) SLODE_CHAIN(3, SLODE_MAX_K + SLODE_MAX_P);
  else if (k.C == 4) SLODE_CHAIN(4, SLODE_MAX_K + SLODE_MAX_P);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
