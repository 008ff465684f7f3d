// EncoderCONV forward / backward for gfx950 (MI355X).
//
// Replaces models/encoder_conv.py:43-51 (forward) and autograd's backward through it:
//   conv1d(C->F, k=K, valid) -> avg_pool1d(P, stride 1) -> flatten (filter-major) -> Linear(F*n_pool -> Hc) -> tanh
//   -> z_loc Linear(Hc -> L), z_scale = exp(Linear(Hc -> L)).
//
// Layout: observations are read through element strides of the logical [B,C,T] tensor, so the reference's
// permuted view of a contiguous [B,T,C] batch (training_cvs.py:25) is consumed in place.  One workgroup owns a
// tile of TBE trajectories; the observation tile, the conv output and the pooled features live in LDS; the
// 374 KB lin.weight is streamed once per tile from L2 with coalesced row reads and shared by the TBE trajectories.
// The backward's lin.weight gradient (the only real GEMM on the path: [Hc x B] x [B x F*n_pool]) runs on the
// f32 MFMA (v_mfma_f32_32x32x2_f32) with operands loaded straight from HBM in their natural row-major layout.
#include "slode_common.h"

typedef const __attribute__((address_space(4))) float* cptr;

#ifdef SLODE_STAMPS
__device__ unsigned long long g_stamps_enc[32];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps_enc[i] = wall_clock64(); } while (0)
extern "C" int slode_debug_stamps_enc(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_enc), sizeof(unsigned long long) * 32);
}
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

constexpr int TBE = 4;       // trajectories per workgroup
constexpr int ENC_NT = 1024; // threads per workgroup (1 workgroup per CU: 16 waves keep ~100 KB of lin.weight loads in flight)
constexpr int RB = 4;        // lin rows per wave pass
constexpr int ENC_NT_BWD = 512;  // backward: 2 waves/SIMD => 256-VGPR budget for its register-blocked phases
constexpr int IU = 5;        // float2 column chunks a lane keeps in flight per row

struct EncK {
  int B, T, C, L, F, K, P, Hc, n_conv, n_pool, FQ;
  const float *conv_w, *conv_b, *lin_w, *lin_b, *zloc_w, *zloc_b, *zls_w, *zls_b;
  const float* obs;
  long long sb, sc, st;
  float *loc, *scale, *pooled, *hid;
  // backward
  const float *scale_in, *pooled_in, *hid_in, *g_loc, *g_scale;
  float *g_pre, *slabs;
  int small_stride;
};

__device__ __forceinline__ void load_obs_tile(const EncK& k, int b0, float* s_x) {
  // s_x[tb][c][t]; iterate in memory order of the source for coalescing
  const int CT = k.C * k.T;
  for (int e = threadIdx.x; e < TBE * CT; e += blockDim.x) {
    const int tb = e / CT, r = e - tb * CT;
    int c, t;
    if (k.sc < k.st) { t = r / k.C; c = r - t * k.C; } else { c = r / k.T; t = r - c * k.T; }
    const int b = b0 + tb;
    s_x[(tb * k.C + c) * k.T + t] = (b < k.B) ? k.obs[(long long)b * k.sb + (long long)c * k.sc + (long long)t * k.st] : 0.f;
  }
}

// ---- forward ---------------------------------------------------------------------------------------------
template <int C, int K>
__global__ void __launch_bounds__(ENC_NT) enc_fwd_kernel(const EncK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CKP = ((C * K + 1) + 3) & ~3;  // taps + bias, padded to a multiple of 4 floats
  const int tid = threadIdx.x, NT = blockDim.x;
  const int T = k.T, F = k.F, n_conv = k.n_conv, n_pool = k.n_pool, FQ = k.FQ, Hc = k.Hc, L = k.L;
  float* s_x = smem;                               // [TBE][C][T]
  float* s_conv = s_x + TBE * C * T;               // [TBE][F][n_conv]
  float* s_pool = s_conv + TBE * F * n_conv;       // [TBE][FQ]
  float* s_hid = s_pool + TBE * FQ;                // [TBE][64]
  float* s_cw = s_hid + TBE * 64;                  // [F][CKP] conv taps (+ bias in the last padded slot), rows 16-B aligned
  float* s_hw = s_cw + F * CKP;                    // [2][L][Hc] z_loc / z_scale head weights
  const int b0 = blockIdx.x * TBE;
  STAMP(0);

  load_obs_tile(k, b0, s_x);
  for (int e = tid; e < F * CKP; e += NT) {
    const int f = e / CKP, r = e - f * CKP;
    s_cw[e] = (r < C * K) ? k.conv_w[f * C * K + r] : ((r == CKP - 1) ? k.conv_b[f] : 0.f);
  }
  for (int e = tid; e < 2 * L * Hc; e += NT) s_hw[e] = (e < L * Hc) ? k.zloc_w[e] : k.zls_w[e - L * Hc];
  __syncthreads();
  STAMP(1);
  // conv: one thread per (tb, p); the K-window of every channel in registers, filter taps broadcast from LDS (b128 reads)
  for (int e = tid; e < TBE * n_conv; e += NT) {
    const int tb = e / n_conv, p = e - tb * n_conv;
    float xw[C][K];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) xw[c][kk] = s_x[(tb * C + c) * T + p + kk];
    for (int f = 0; f < F; ++f) {
      const float4* w4 = reinterpret_cast<const float4*>(s_cw + f * CKP);
      float wr[CKP];
#pragma unroll
      for (int q = 0; q < CKP / 4; ++q) {
        const float4 v = w4[q];
        wr[4 * q] = v.x; wr[4 * q + 1] = v.y; wr[4 * q + 2] = v.z; wr[4 * q + 3] = v.w;
      }
      float acc = wr[CKP - 1];  // bias
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kk = 0; kk < K; ++kk) acc = fmaf(wr[c * K + kk], xw[c][kk], acc);
      s_conv[(tb * F + f) * n_conv + p] = acc;
    }
  }
  __syncthreads();
  STAMP(2);
  // average pool, stride 1 (sum / P as ATen's avg_pool does), filter-major flatten; one wave per (tb, f) row
  {
    const float fP = (float)k.P;
    const int wave = tid >> 6, lane = tid & 63, nw = NT >> 6;
    for (int row = wave; row < TBE * F; row += nw) {
      const int tb = row / F, f = row - tb * F;
      const float* cr = s_conv + row * n_conv;
      for (int q = lane; q < n_pool; q += 64) {
        float tap[SLODE_MAX_P];
#pragma unroll
        for (int j = 0; j < SLODE_MAX_P; ++j) tap[j] = cr[min(q + j, n_conv - 1)];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < SLODE_MAX_P; ++j) sum += (j < k.P) ? tap[j] : 0.f;
        const float v = sum / fP;
        s_pool[tb * FQ + f * n_pool + q] = v;
        if (k.pooled && b0 + tb < k.B) k.pooled[(long long)(b0 + tb) * FQ + f * n_pool + q] = v;
      }
    }
  }
  __syncthreads();
  STAMP(3);
  // lin + tanh: each wave streams RB rows of lin.weight (coalesced) against the TBE pooled vectors in LDS
  {
    const int wave = tid >> 6, lane = tid & 63, nw = NT >> 6;
    const int ngroups = (Hc + RB - 1) / RB;
    for (int g = wave; g < ngroups; g += nw) {
      const int m0 = g * RB;
      float acc[RB][TBE];
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int tb = 0; tb < TBE; ++tb) acc[r][tb] = 0.f;
      if ((FQ & 1) == 0) {
        // each lane owns column pairs i = 2*lane + 128*it; all RB x IU float2 loads of a macro-iteration are issued
        // before the first FMA so ~10 KB per wave are in flight (the phase is L2-latency bound otherwise)
        const float* wrow[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) wrow[r] = k.lin_w + (long long)min(m0 + r, Hc - 1) * FQ;
        for (int i0 = 2 * lane; i0 < FQ; i0 += 128 * IU) {
          float2 w[RB][IU];
#pragma unroll
          for (int u = 0; u < IU; ++u) {
            const int i = i0 + 128 * u;
            const int ic = min(i, FQ - 2);  // unconditional load from a clamped address, zeroed by a select below:
#pragma unroll                          // a predicated load would become a branch + s_waitcnt per load
            for (int r = 0; r < RB; ++r) w[r][u] = *reinterpret_cast<const float2*>(wrow[r] + ic);
          }
#pragma unroll
          for (int u = 0; u < IU; ++u) {
            const bool in = i0 + 128 * u < FQ;
#pragma unroll
            for (int r = 0; r < RB; ++r) { w[r][u].x = in ? w[r][u].x : 0.f; w[r][u].y = in ? w[r][u].y : 0.f; }
          }
#pragma unroll
          for (int u = 0; u < IU; ++u) {
            const int i = min(i0 + 128 * u, FQ - 2);  // out-of-range chunks carry zero weights
#pragma unroll
            for (int tb = 0; tb < TBE; ++tb) {
              const float2 pv = *reinterpret_cast<const float2*>(s_pool + tb * FQ + i);
#pragma unroll
              for (int r = 0; r < RB; ++r) acc[r][tb] = fmaf(w[r][u].y, pv.y, fmaf(w[r][u].x, pv.x, acc[r][tb]));
            }
          }
        }
      } else {
#pragma unroll 4
        for (int i = lane; i < FQ; i += 64) {
          float pv[TBE];
#pragma unroll
          for (int tb = 0; tb < TBE; ++tb) pv[tb] = s_pool[tb * FQ + i];
#pragma unroll
          for (int r = 0; r < RB; ++r) {
            const int m = min(m0 + r, Hc - 1);
            const float w = k.lin_w[(long long)m * FQ + i];
#pragma unroll
            for (int tb = 0; tb < TBE; ++tb) acc[r][tb] = fmaf(w, pv[tb], acc[r][tb]);
          }
        }
      }
      static_assert(RB * TBE == 16, "wave_sum16 reduces RB x TBE = 16 partial sums");
      float flat[16];
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int tb = 0; tb < TBE; ++tb) flat[r * TBE + tb] = acc[r][tb];
      const float v = wave_sum16(flat, lane);
      {
        const int idx = (lane >> 2) & 15, r = idx / TBE, tb = idx - r * TBE;
        const int mm = min(m0 + r, Hc - 1);
        const float hv = tanhf(v + k.lin_b[mm]);
        if ((lane & 3) == 0 && m0 + r < Hc) {
          s_hid[tb * 64 + mm] = hv;
          if (k.hid && b0 + tb < k.B) k.hid[(long long)(b0 + tb) * Hc + mm] = hv;
        }
      }
    }
  }
  __syncthreads();
  STAMP(4);
  // heads: z_loc, z_scale = exp(.)
  for (int e = tid; e < TBE * L * 2; e += NT) {
    const int which = e / (TBE * L), r = e - which * (TBE * L);
    const int tb = r / L, l = r - tb * L;
    const float* W = s_hw + which * L * Hc;
    float acc = which ? k.zls_b[l] : k.zloc_b[l];
#pragma unroll 10
    for (int mm = 0; mm < Hc; ++mm) acc = fmaf(W[l * Hc + mm], s_hid[tb * 64 + mm], acc);
    if (b0 + tb < k.B) {
      if (which) k.scale[(long long)(b0 + tb) * L + l] = expf(acc);
      else k.loc[(long long)(b0 + tb) * L + l] = acc;
    }
  }
  STAMP(5);
}

// ---- backward, part 1: heads, tanh, lin^T, pool^T, conv weight gradient -----------------------------------
// small slab layout: [conv_w F*C*K][conv_b F][lin_b Hc][zloc_w L*Hc][zloc_b L][zls_w L*Hc][zls_b L]
template <int C, int K>
__global__ void __launch_bounds__(ENC_NT_BWD) enc_bwd_kernel(const EncK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int T = k.T, F = k.F, n_conv = k.n_conv, n_pool = k.n_pool, FQ = k.FQ, Hc = k.Hc, L = k.L;
  float* s_x = smem;                            // [TBE][C][T]
  float* s_gpool = s_x + TBE * C * T;           // [TBE][FQ]     (later: cross-wave reduction scratch)
  float* s_gconv = s_gpool + TBE * FQ;          // [TBE][F][n_conv]
  float* s_gpre = s_gconv + TBE * F * n_conv;   // [64][TBE]     (m-major so one b128 read feeds the TBE FMAs)
  float* s_hid = s_gpre + 64 * TBE;             // [TBE][64]
  float* s_gl = s_hid + TBE * 64;               // [TBE][L]  g_loc
  float* s_gs = s_gl + TBE * L;                 // [TBE][L]  g_scale * scale
  float* s_hw = s_gs + TBE * L;                 // [2][L][Hc] z_loc / z_scale head weights
  const int b0 = blockIdx.x * TBE;
  float* slab = k.slabs + (long long)blockIdx.x * k.small_stride;
  const int o_convw = 0, o_convb = F * C * K, o_linb = o_convb + F, o_zlw = o_linb + Hc, o_zlb = o_zlw + L * Hc,
            o_zsw = o_zlb + L, o_zsb = o_zsw + L * Hc;

  STAMP(8);
  load_obs_tile(k, b0, s_x);
  for (int e = tid; e < 2 * L * Hc; e += NT) s_hw[e] = (e < L * Hc) ? k.zloc_w[e] : k.zls_w[e - L * Hc];
  for (int e = tid; e < TBE * L; e += NT) {
    const int tb = e / L, l = e - tb * L, b = b0 + tb;
    const bool ok = b < k.B;
    s_gl[e] = ok ? k.g_loc[(long long)b * L + l] : 0.f;
    s_gs[e] = ok ? k.g_scale[(long long)b * L + l] * k.scale_in[(long long)b * L + l] : 0.f;
  }
  for (int e = tid; e < TBE * 64; e += NT) {
    const int tb = e >> 6, mm = e & 63, b = b0 + tb;
    s_hid[e] = (b < k.B && mm < Hc) ? k.hid_in[(long long)b * Hc + mm] : 0.f;
  }
  __syncthreads();
  STAMP(9);
  // through the heads and tanh
  for (int e = tid; e < TBE * 64; e += NT) {
    const int tb = e >> 6, mm = e & 63;
    float g = 0.f;
    if (mm < Hc) {
#pragma unroll 4
      for (int l = 0; l < L; ++l) {
        g = fmaf(s_hw[l * Hc + mm], s_gl[tb * L + l], g);
        g = fmaf(s_hw[(L + l) * Hc + mm], s_gs[tb * L + l], g);
      }
      const float hv = s_hid[tb * 64 + mm];
      g *= (1.f - hv * hv);
    }
    s_gpre[mm * TBE + tb] = g;
    if (b0 + tb < k.B) k.g_pre[(long long)(b0 + tb) * 64 + mm] = g;
  }
  // head weight / bias partials (sum over the tile's trajectories)
  for (int e = tid; e < L * Hc; e += NT) {
    const int l = e / Hc, mm = e - l * Hc;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int tb = 0; tb < TBE; ++tb) {
      a1 = fmaf(s_gl[tb * L + l], s_hid[tb * 64 + mm], a1);
      a2 = fmaf(s_gs[tb * L + l], s_hid[tb * 64 + mm], a2);
    }
    slab[o_zlw + e] = a1;
    slab[o_zsw + e] = a2;
  }
  for (int l = tid; l < L; l += NT) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int tb = 0; tb < TBE; ++tb) { a1 += s_gl[tb * L + l]; a2 += s_gs[tb * L + l]; }
    slab[o_zlb + l] = a1;
    slab[o_zsb + l] = a2;
  }
  __syncthreads();
  STAMP(10);
  for (int mm = tid; mm < Hc; mm += NT) {
    float a = 0.f;
#pragma unroll
    for (int tb = 0; tb < TBE; ++tb) a += s_gpre[mm * TBE + tb];
    slab[o_linb + mm] = a;
  }
  // g_pooled[tb][i] = sum_m g_pre[tb][m] * lin_w[m][i]   (thread <-> column pair, coalesced rows of lin_w, MB loads in flight)
  if ((FQ & 1) == 0) {
    constexpr int MB = 25;
    for (int i = 2 * tid; i < FQ; i += 2 * NT) {
      float acc[TBE][2];
#pragma unroll
      for (int tb = 0; tb < TBE; ++tb) acc[tb][0] = acc[tb][1] = 0.f;
      for (int mb = 0; mb < Hc; mb += MB) {
        float2 w[MB];
#pragma unroll
        for (int q = 0; q < MB; ++q)  // unconditional loads (clamped row); rows >= Hc meet g_pre == 0 below
          w[q] = *reinterpret_cast<const float2*>(k.lin_w + (long long)min(mb + q, Hc - 1) * FQ + i);
#pragma unroll
        for (int q = 0; q < MB; ++q) {
          const float4 g = *reinterpret_cast<const float4*>(s_gpre + min(mb + q, 63) * TBE);
          acc[0][0] = fmaf(g.x, w[q].x, acc[0][0]); acc[0][1] = fmaf(g.x, w[q].y, acc[0][1]);
          acc[1][0] = fmaf(g.y, w[q].x, acc[1][0]); acc[1][1] = fmaf(g.y, w[q].y, acc[1][1]);
          acc[2][0] = fmaf(g.z, w[q].x, acc[2][0]); acc[2][1] = fmaf(g.z, w[q].y, acc[2][1]);
          acc[3][0] = fmaf(g.w, w[q].x, acc[3][0]); acc[3][1] = fmaf(g.w, w[q].y, acc[3][1]);
        }
      }
#pragma unroll
      for (int tb = 0; tb < TBE; ++tb) { s_gpool[tb * FQ + i] = acc[tb][0]; s_gpool[tb * FQ + i + 1] = acc[tb][1]; }
    }
  } else {
    for (int i = tid; i < FQ; i += NT) {
      float acc[TBE];
#pragma unroll
      for (int tb = 0; tb < TBE; ++tb) acc[tb] = 0.f;
#pragma unroll 10
      for (int mm = 0; mm < Hc; ++mm) {
        const float w = k.lin_w[(long long)mm * FQ + i];
        const float4 g = *reinterpret_cast<const float4*>(s_gpre + mm * TBE);
        acc[0] = fmaf(g.x, w, acc[0]); acc[1] = fmaf(g.y, w, acc[1]);
        acc[2] = fmaf(g.z, w, acc[2]); acc[3] = fmaf(g.w, w, acc[3]);
      }
#pragma unroll
      for (int tb = 0; tb < TBE; ++tb) s_gpool[tb * FQ + i] = acc[tb];
    }
  }
  __syncthreads();
  STAMP(11);
  // pool^T: g_conv[p] = (1/P) * sum_{q in [p-P+1, p] ∩ [0, n_pool)} g_pooled[q]; one wave per (tb, f) row
  {
    const float fP = (float)k.P;
    const int wave = tid >> 6, lane = tid & 63, nw = NT >> 6;
    for (int row = wave; row < TBE * F; row += nw) {
      const int tb = row / F, f = row - tb * F;
      const float* gr = s_gpool + tb * FQ + f * n_pool;
      for (int p = lane; p < n_conv; p += 64) {
        float tap[SLODE_MAX_P];
#pragma unroll
        for (int j = 0; j < SLODE_MAX_P; ++j) tap[j] = gr[min(max(p - j, 0), n_pool - 1)];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < SLODE_MAX_P; ++j) sum += (j < k.P && p - j >= 0 && p - j < n_pool) ? tap[j] : 0.f;
        s_gconv[row * n_conv + p] = sum / fP;
      }
    }
  }
  __syncthreads();
  STAMP(12);
  // conv weight gradient: lane f = tid % 16, slot = tid / 16 walks (tb, block of 8 output positions)
  {
    constexpr int PB = 8;
    const int f = tid & 15, slot = tid >> 4, nslot = NT >> 4;
    const int nblk = (n_conv + PB - 1) / PB;
    float acc[C][K], accb = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) acc[c][kk] = 0.f;
    if (f < F) {
      for (int it = slot; it < TBE * nblk; it += nslot) {
        const int tb = it / nblk, p0 = (it - tb * nblk) * PB;
        float G[PB];
#pragma unroll
        for (int dp = 0; dp < PB; ++dp) {
          const float gv = s_gconv[(tb * F + f) * n_conv + min(p0 + dp, n_conv - 1)];
          G[dp] = (p0 + dp < n_conv) ? gv : 0.f;
          accb += G[dp];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float X[PB + K - 1];
#pragma unroll
          for (int j = 0; j < PB + K - 1; ++j) {
            const float xv = s_x[(tb * C + c) * T + min(p0 + j, T - 1)];
            X[j] = (p0 + j < T) ? xv : 0.f;
          }
#pragma unroll
          for (int kk = 0; kk < K; ++kk)
#pragma unroll
            for (int dp = 0; dp < PB; ++dp) acc[c][kk] = fmaf(G[dp], X[dp + kk], acc[c][kk]);
        }
      }
    }
    // lanes l, l^16, l^32 share f inside a wave; then a fixed-order sum over waves through LDS
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        float v = acc[c][kk];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[c][kk] = v;
      }
    accb += __shfl_xor(accb, 16, 64);
    accb += __shfl_xor(accb, 32, 64);
    __syncthreads();  // s_gpool is free now
    STAMP(13);
    float* s_red = s_gpool;  // [nw][16][C*K+1]
    const int wave = tid >> 6, lane = tid & 63, nw = NT >> 6;
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kk = 0; kk < K; ++kk) s_red[(wave * 16 + lane) * (C * K + 1) + c * K + kk] = acc[c][kk];
      s_red[(wave * 16 + lane) * (C * K + 1) + C * K] = accb;
    }
    __syncthreads();
    for (int e = tid; e < F * (C * K + 1); e += NT) {
      const int ff = e / (C * K + 1), r = e - ff * (C * K + 1);
      float v = 0.f;
      for (int w = 0; w < nw; ++w) v += s_red[(w * 16 + ff) * (C * K + 1) + r];
      if (r < C * K) slab[o_convw + ff * C * K + r] = v;
      else slab[o_convb + ff] = v;
    }
  }
  STAMP(14);
}

// ---- backward, part 2: lin.weight gradient on the f32 matrix cores -----------------------------------------
// g_W[m][i] = sum_b g_pre[b][m] * pooled[b][i].  v_mfma_f32_32x32x2_f32: A[i=m][k=b], B[k=b][j=i]; both operands are
// rows of row-major HBM arrays, so lane (l&31, l>>5) loads element [b0 + (l>>5)][base + (l&31)] directly (coalesced).
// Workgroup = 4 waves = 4 K-splits of one 64(m) x 32(i) output tile, reduced in LDS; grid = (i-tiles, splitk_grid).
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int KP = 16;  // batch-pairs (K = 2 each) whose operands a wave loads before issuing the MFMAs

// With ones_col != 0 the right-hand matrix gets one extra all-ones column (index FQ): output column FQ = sum_b g_pre[b][m], and the
// slabs have FQ + 1 columns per row (folded encoder path: g_beff comes for free).
struct GemmProb { const float* A; int lda; const float* X; float* slabs; int M, N, ntiles; };   // A: [B][lda], rows m < 64 used
struct GemmProbs { GemmProb p[3]; int n_gemm_x; Stage1 rider; int rider_bx; int splitk_grid; };   // rider: stage-1 slab reduction in extra blocks; splitk_grid > 0: one-dimensional grid [products (tile, split) | riders]
// pl_A0 / pl_X0 = ps.p[0].A / .X (the big product's operands) as leading, SGPR-preloaded arguments (see ode_elbo_kernel)
__global__ void __launch_bounds__(256) enc_bwd_lin_kernel(const float* __restrict__ pl_A0, const float* __restrict__ pl_X0, const GemmProbs ps, int B,
                                                          int per_wave, int ones_col) {
  __shared__ __attribute__((aligned(16))) float s_part[4 * 32 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // One-dimensional grid, the products' blocks FIRST: they are dispatched to CUs of their own before the rider blocks fill in (with the
  // riders interleaved in dispatch order the two kinds delayed each other: products alone 5.3 us, riders alone 4.7, together 6.6).
  // (The layer-by-layer path launches the same kernel two-dimensional without riders: both index forms are handled.)
  const int n_prod = ps.n_gemm_x * ps.splitk_grid;
  const int blk = ps.splitk_grid ? (int)blockIdx.x : -1;
  if (blk >= n_prod) {   // rider blocks: (bx, group) = stage 1 of the ODE-slab reduction, independent of the GEMMs
    const int r = blk - n_prod;
    const int bx = r % ps.rider_bx, g = r / ps.rider_bx;
    if (g < SLODE_REDUCE_GROUPS) slab_stage1_block(ps.rider, bx, g, s_part);
    return;
  }
  const int by = ps.splitk_grid ? blk / ps.n_gemm_x : (int)blockIdx.y;
  // up to three independent products share the launch (column tiles laid end to end)
  int tile = ps.splitk_grid ? blk - by * ps.n_gemm_x : (int)blockIdx.x;
  const int which = tile < ps.p[0].ntiles ? 0 : (tile < ps.p[0].ntiles + ps.p[1].ntiles ? 1 : 2);
  tile -= which == 0 ? 0 : (which == 1 ? ps.p[0].ntiles : ps.p[0].ntiles + ps.p[1].ntiles);
  const GemmProb p = which == 0 ? ps.p[0] : (which == 1 ? ps.p[1] : ps.p[2]);
  const float* __restrict__ g_pre = which == 0 ? pl_A0 : p.A;
  const float* __restrict__ pooled = which == 0 ? pl_X0 : p.X;
  float* __restrict__ slabs = p.slabs;
  const int Hc = p.M, FQ = p.N, lda = p.lda;
  const int i0 = tile * 32;
  const int ks = by * 4 + wave;
  const int bbeg = ks * per_wave, bend = min(B, bbeg + per_wave);
  const int col = lane & 31, kh = lane >> 5;
  const bool col_ok = i0 + col < FQ;
  const bool col_one = ones_col && (i0 + col == FQ);
  const int icol = min(i0 + col, FQ - 1);
  const int NO = FQ + (ones_col ? 1 : 0);   // output columns per slab row
  f32x16 acc0 = {0}, acc1 = {0};
  for (int bb = bbeg; bb < bend; bb += 2 * KP) {
    float a0[KP], a1[KP], bv[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const int b = min(bb + 2 * q + kh, B - 1);  // unconditional clamped loads, masked by selects below
      a0[q] = g_pre[(long long)b * lda + col];
      a1[q] = g_pre[(long long)b * lda + 32 + col];
      bv[q] = pooled[(long long)b * FQ + icol];
    }
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const bool ok = bb + 2 * q + kh < bend;
      a0[q] = ok ? a0[q] : 0.f;
      a1[q] = ok ? a1[q] : 0.f;
      bv[q] = ok ? (col_ok ? bv[q] : (col_one ? 1.f : 0.f)) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bv[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bv[q], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s_part[(wave * 32 + r) * 64 + lane] = acc0[r];
    s_part[(wave * 32 + 16 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  float* slab = slabs + (long long)by * Hc * NO;
  for (int e = tid; e < 32 * 64; e += 256) {
    const int reg = e >> 6, ln = e & 63;
    const float v = (s_part[(0 * 32 + reg) * 64 + ln] + s_part[(1 * 32 + reg) * 64 + ln]) +
                    (s_part[(2 * 32 + reg) * 64 + ln] + s_part[(3 * 32 + reg) * 64 + ln]);
    const int r16 = reg & 15, tile = reg >> 4;
    const int m = tile * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (ln >> 5);  // C/D map of the 32x32 MFMA
    const int i = i0 + (ln & 31);
    if (m < Hc && i < NO) slab[(long long)m * NO + i] = v;
  }
}

EncK make_enck(const slode_shape& s, const slode_layout& lay, const float* p) {
  EncK k{};
  k.B = s.B; k.T = s.T; k.C = s.C; k.L = s.L; k.F = s.F; k.K = s.K; k.P = s.P; k.Hc = s.Hc;
  k.n_conv = s.T - s.K + 1; k.n_pool = k.n_conv - s.P + 1; k.FQ = s.F * k.n_pool;
  k.conv_w = p + lay.conv_w; k.conv_b = p + lay.conv_b; k.lin_w = p + lay.lin_w; k.lin_b = p + lay.lin_b;
  k.zloc_w = p + lay.zloc_w; k.zloc_b = p + lay.zloc_b; k.zls_w = p + lay.zls_w; k.zls_b = p + lay.zls_b;
  return k;
}

size_t enc_fwd_lds(const EncK& k) {
  const int ckp = ((k.C * k.K + 1) + 3) & ~3;
  return sizeof(float) * ((size_t)TBE * k.C * k.T + (size_t)TBE * k.F * k.n_conv + (size_t)TBE * k.FQ + TBE * 64 +
                          (size_t)k.F * ckp + 2 * (size_t)k.L * k.Hc);
}
size_t enc_bwd_lds(const EncK& k) {
  size_t gpool = (size_t)TBE * k.FQ;
  const size_t red = (size_t)(ENC_NT_BWD / 64) * 16 * (k.C * k.K + 1);
  if (red > gpool) gpool = red;
  return sizeof(float) * ((size_t)TBE * k.C * k.T + gpool + (size_t)TBE * k.F * k.n_conv + 64 * TBE + TBE * 64 +
                          2 * (size_t)TBE * k.L + 2 * (size_t)k.L * k.Hc);
}

}  // namespace

hipError_t slode_launch_gemm_tail(const float* g_pre, const float* x, float* gslabs, int Hc, int CT, const float* glat, const float* hid,
                                  float* gslabs_loc, float* gslabs_ls, int L, int B, int splitk, const float* ode_slabs, int ode_stride,
                                  int ode_n, int ode_count, float* ode_part, const float** ode_part_out, int* ode_n_out,
                                  hipStream_t stream, int zr_rows, int zr_lo, int zr_hi) {
  const int total_splits = splitk * 4;
  int per_wave = (B + total_splits - 1) / total_splits;
  per_wave = (per_wave + 1) & ~1;
  GemmProbs ps{};
  ps.p[0] = GemmProb{g_pre, 64, x, gslabs, Hc, CT, (CT + 1 + 31) / 32};
  ps.p[1] = GemmProb{glat, 128, hid, gslabs_loc, L, Hc, (Hc + 1 + 31) / 32};
  ps.p[2] = GemmProb{glat + 64, 128, hid, gslabs_ls, L, Hc, (Hc + 1 + 31) / 32};
  ps.n_gemm_x = ps.p[0].ntiles + ps.p[1].ntiles + ps.p[2].ntiles;
  int rider_x = 0;
  *ode_part_out = ode_slabs; *ode_n_out = ode_n;
  if (ode_part && ode_n > 2 * SLODE_REDUCE_GROUPS) {
    const int per = (ode_n + SLODE_REDUCE_GROUPS - 1) / SLODE_REDUCE_GROUPS;
    ps.rider = Stage1{ode_slabs, ode_stride, ode_n, ode_count, per, ode_part, zr_rows, zr_lo, zr_hi};
    ps.rider_bx = (ode_count + SLODE_S1_COLS - 1) / SLODE_S1_COLS;
    rider_x = (ps.rider_bx * SLODE_REDUCE_GROUPS + splitk - 1) / splitk;
    *ode_part_out = ode_part; *ode_n_out = (ode_n + per - 1) / per;
  }
  ps.splitk_grid = splitk;
  const int n_rider = ps.rider_bx * (rider_x ? SLODE_REDUCE_GROUPS : 0);
  SLODE_LAUNCH("enc_bwd_lin", enc_bwd_lin_kernel, dim3(ps.n_gemm_x * splitk + n_rider), dim3(256), 0, stream, ps.p[0].A, ps.p[0].X, ps, B, per_wave, 1);
  return hipGetLastError();
}

int slode_enc_small_count(const slode_shape& s) { return s.F * s.C * s.K + s.F + s.Hc + 2 * (s.L * s.Hc + s.L); }
int slode_enc_bwd_grid(const slode_shape& s) { return (s.B + TBE - 1) / TBE; }
int slode_enc_lin_splitk(const slode_shape& s) {
  int g = (s.B + 127) / 128;  // 4 waves per workgroup => <= 32 trajectories (one prefetch batch) per wave up to B = 2048
  if (g < 1) g = 1;
  if (g > 16) g = 16;
  return g;
}

hipError_t slode_launch_enc_fwd(const EncLaunch& a, hipStream_t stream) {
  EncK k = make_enck(a.s, a.lay, a.params);
  k.obs = a.obs; k.sb = a.sb; k.sc = a.sc; k.st = a.st;
  k.loc = a.loc; k.scale = a.scale; k.pooled = a.pooled; k.hid = a.hid;
  const size_t lds = enc_fwd_lds(k);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int grid = (a.s.B + TBE - 1) / TBE;
  if (a.s.C == 3 && a.s.K == 10) {
    (void)hipFuncSetAttribute((const void*)enc_fwd_kernel<3, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SLODE_LAUNCH("enc_fwd", (enc_fwd_kernel<3, 10>), dim3(grid), dim3(ENC_NT), lds, stream, k);
  } else if (a.s.C == 4 && a.s.K == 10) {
    (void)hipFuncSetAttribute((const void*)enc_fwd_kernel<4, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SLODE_LAUNCH("enc_fwd", (enc_fwd_kernel<4, 10>), dim3(grid), dim3(ENC_NT), lds, stream, k);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t slode_launch_enc_bwd(const EncBwdLaunch& a, hipStream_t stream) {
  EncK k = make_enck(a.s, a.lay, a.params);
  k.obs = a.obs; k.sb = a.sb; k.sc = a.sc; k.st = a.st;
  k.scale_in = a.scale; k.pooled_in = a.pooled; k.hid_in = a.hid; k.g_loc = a.g_loc; k.g_scale = a.g_scale;
  k.g_pre = a.g_pre; k.slabs = a.slabs_small; k.small_stride = a.small_stride;
  const size_t lds = enc_bwd_lds(k);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (a.s.C == 3 && a.s.K == 10) {
    (void)hipFuncSetAttribute((const void*)enc_bwd_kernel<3, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SLODE_LAUNCH("enc_bwd", (enc_bwd_kernel<3, 10>), dim3(a.grid_small), dim3(ENC_NT_BWD), lds, stream, k);
  } else if (a.s.C == 4 && a.s.K == 10) {
    (void)hipFuncSetAttribute((const void*)enc_bwd_kernel<4, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SLODE_LAUNCH("enc_bwd", (enc_bwd_kernel<4, 10>), dim3(a.grid_small), dim3(ENC_NT_BWD), lds, stream, k);
  } else {
    return hipErrorInvalidValue;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // lin.weight gradient (MFMA split-K); consumes g_pre written by the kernel above (same stream => ordered)
  const int total_splits = a.splitk * 4;
  int per_wave = (a.s.B + total_splits - 1) / total_splits;
  per_wave = (per_wave + 1) & ~1;
  GemmProbs ps{};
  ps.p[0] = GemmProb{a.g_pre, 64, a.pooled, a.slabs_lin, a.s.Hc, k.FQ, (k.FQ + 31) / 32};
  ps.n_gemm_x = ps.p[0].ntiles;
  SLODE_LAUNCH("enc_bwd_lin", enc_bwd_lin_kernel, dim3(ps.p[0].ntiles, a.splitk), dim3(256), 0, stream, ps.p[0].A, ps.p[0].X, ps, a.s.B, per_wave, 0);
  return hipGetLastError();
}
