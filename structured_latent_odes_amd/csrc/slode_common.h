// Shared declarations for libslode.so (gfx950 only).  Internal; the public ABI is include/slode.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <math.h>
#include "../../include/slode.h"

#define SLODE_MAX_T 1024
#define SLODE_MAX_S 8
#define SLODE_MAX_H 32
#define SLODE_MAX_L 64
#define SLODE_MAX_C 4
#define SLODE_MAX_NU 16
#define SLODE_MAX_F 16
#define SLODE_MAX_K 16
#define SLODE_MAX_P 8
#define SLODE_MAX_HC 64

// Measurement aid (slode_profile_enable): while a profiled entry point runs, every kernel it launches goes out through
// hipExtLaunchKernelGGL with its OWN start / stop event pair -- the begin and end timestamps of that dispatch, i.e. the quantity
// rocprofv3 --kernel-trace reports -- and no extra packet enters the stream.  One table per handle; the launching thread finds it
// through a thread-local pointer that is non-null only inside that entry point.
#define SLODE_CLOCK_MAX SLODE_PROFILE_MAX_KERNELS
struct ClockTable { int n; const char* name[SLODE_CLOCK_MAX]; hipEvent_t ev[SLODE_CLOCK_MAX][2]; };
extern thread_local ClockTable* g_slode_clock;
#define SLODE_LAUNCH(NAME, fn, grid, block, lds, stream, ...)                                                                        \
  do {                                                                                                                               \
    ClockTable* ct_ = g_slode_clock;                                                                                                 \
    if (ct_ && ct_->n < SLODE_CLOCK_MAX) {                                                                                           \
      const int i_ = ct_->n++;                                                                                                       \
      ct_->name[i_] = NAME;                                                                                                          \
      hipExtLaunchKernelGGL(fn, grid, block, lds, stream, ct_->ev[i_][0], ct_->ev[i_][1], 0, __VA_ARGS__);                            \
    } else {                                                                                                                         \
      hipLaunchKernelGGL(fn, grid, block, lds, stream, __VA_ARGS__);                                                                 \
    }                                                                                                                                \
  } while (0)

struct slode_ctx {
  int device;
  int num_cu;
  char err[512];
  int profile;            // slode_profile_enable: kernels of the step entry points carry their own timestamps
  ClockTable clk;
  int ev_ready;           // events created
  int no_fold;            // env SLODE_NO_FOLD: use the layer-by-layer encoder kernels inside slode_elbo_step
  int adam_lo2, adam_hi2; // slode_adam_region: elements with their own Adam step count = step + adam_delta2
  int64_t adam_delta2;
  // diagnostics / test hooks, read from the environment ONCE in slode_create (never at launch time):
  int ode_loop;           // SLODE_ODE_LOOP: persistent-loop grid even when every trajectory could have its own workgroup
  int ode_generic;        // SLODE_ODE_GENERIC: skip the shape-specialised instantiations
  int ode_alg;            // SLODE_ODE_ALG = 1 / 2: measured A/B arms of the fused kernel (metric shape only; ode_kernel.hip)
  int enc_fuse;           // SLODE_ENC_FUSE (default 1): encoder forward inside the ODE kernel where an instantiation exists
  int ode_pack;           // SLODE_ODE_PACK = 4: four trajectories per ODE workgroup (metric shape)
  int ode_grid_cap;       // SLODE_ODE_GRID = n: at most n workgroups in the persistent-loop grid (tests: several trajectories per workgroup at small B)
  // slode_rng_seed: the Philox stream of the calls that draw their own noise (eps == NULL); rng_counter = drawing calls made so far
  uint64_t rng_seed, rng_counter; int64_t rng_b0;
  // in-launch fold (SLODE_FOLD_NEXT, default on): the chain launch of a step that updates the weights leaves W_eff & co. of the NEW weights
  // in the workspace; the next step on the same (workspace, params) skips its fold launch.  fold_gen: launches that counted on `done`.
  int fold_on, fold_valid, fold_tmajor;
  const void* fold_ws; const void* fold_params;
  unsigned int fold_gen;
  int dp5_w64;            // SLODE_DP5_LPT: lanes per trajectory of the forward adaptive solve (8, 16, 32, 64; 0 = by batch size)
  int chain_resident; int chain_resident_sig[8];   // cached occupancy answer for the shape (T, C, F, K, P, Hc, L, n_params)
};

// ---- reparameterisation noise drawn in the kernels (slode_rng_seed; eps == NULL) and labels as the loader yields them -------------
// Philox-4x32-10 (Salmon et al., SC'11), counter-based: the draw for (trajectory b, latent index l) of the handle's n-th drawing call is
//   block = philox(counter = [b + b0 (low 32) | l >> 2 | n (low 32) | n (high 32)], key = seed),  eps = Box-Muller(block)[l & 3]
// -- no state, no ordering between threads, and independent of how the batch is sharded over GPUs (b0 = the shard's first global
// trajectory).  Latent indices run in the guide's site order (mechanistic_cvs.py:225-237: z_iext, z_rtpr, z_epsilon are consecutive
// ranges of the concatenated latent), so l IS the site order.  tests/test_rng_cpu.py restates it in numpy.
struct RngK { unsigned int k0, k1, c2, c3; long long b0; int on; };
// label columns of u[B, n_u] as separate dense [B, width] tensors (n == 0: one dense matrix, the `u` pointer)
struct LabelSrc { const float* p[SLODE_MAX_LABELS]; int off[SLODE_MAX_LABELS + 1]; int n; };

__host__ __device__ inline void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0, unsigned int k1,
                                              unsigned int (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned int)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#ifdef __HIPCC__
// standard normal for (trajectory b of this launch, latent index l): words (0, 1) of the block give the pair of l & 2 == 0, words (2, 3)
// the other; u = ((x >> 9) + 0.5) 2^-23 in (0, 1) (exact in fp32), r = sqrt(-2 ln u_a), angle 2 pi u_b, cos for even l, sin for odd l
__device__ __forceinline__ float slode_rng_normal(const RngK& r, long long b, int l) {
  unsigned int x[4];
  philox4x32_10((unsigned int)(b + r.b0), (unsigned int)(l >> 2), r.c2, r.c3, r.k0, r.k1, x);
  const unsigned int xa = (l & 2) ? x[2] : x[0], xb = (l & 2) ? x[3] : x[1];
  const float ua = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-7f, ub = ((float)(xb >> 9) + 0.5f) * 1.1920928955078125e-7f;   // exact in fp32
  const float rad = sqrtf(-2.0f * logf(ua));
  float sn, cs;
  sincospif(2.0f * ub, &sn, &cs);
  return rad * ((l & 1) ? sn : cs);
}
__device__ __forceinline__ float slode_eps_at(const RngK& r, const float* eps, long long b, int L, int l) {
  return r.on ? slode_rng_normal(r, b, l) : eps[b * L + l];
}
__device__ __forceinline__ float slode_label_at(const LabelSrc& ls, const float* u, int nu, long long b, int col) {
  if (ls.n == 0) return u[b * nu + col];
  int i = 0;
#pragma unroll
  for (int q = 1; q < SLODE_MAX_LABELS; ++q) i = (q < ls.n && col >= ls.off[q]) ? q : i;
  const int w = ls.off[i + 1] - ls.off[i];
  return ls.p[i][b * w + (col - ls.off[i])];
}
#endif

// ---- device helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// sigmoid via v_exp_f32 / v_rcp_f32 (abs error < 1e-7 on the range the dynamics net produces)
__device__ __forceinline__ float sigmoidf_fast(float x) {
  return fast_rcp(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
// torch.nn.Softplus(beta=1, threshold=20): models/decoders.py:52
__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }


// One DPP move of a float (quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140): a VALU
// operand modifier, no LDS-pipe round trip (ds_bpermute)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, valid in every lane of the row (fixed order: xor 1, xor 2, half mirror, mirror)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}
// sum over each half-wave (lanes 0..31, lanes 32..63), valid in every lane of the half: the row's four DPP adds, then v_permlane16_swap
// hands every row the sum of the other row of its half
__device__ __forceinline__ float half_wave_sum(float v) {
  v = row16_sum(v);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the wave, valid in every lane, without the LDS pipe (six ds_bpermute until round 3): the four DPP adds of a row, then gfx950's
// v_permlane16_swap / v_permlane32_swap hand every row the sum of its neighbour row, every half the sum of the other half.  Fixed order.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Sixteen wave-wide sums for the price of 15 exchanges (instead of 96 shuffles): a halving butterfly over lane bits 5..2 leaves ONE of
// the sixteen values per lane, two more exchanges finish it.  Returns sum over the wave of a[(lane >> 2) & 15] (fixed order).  No
// exchange touches the LDS pipe (round 3; until then 17 ds_bpermute): the two cross-row levels are gfx950's v_permlane32_swap /
// v_permlane16_swap -- one instruction hands each half (each row parity) the other's operand -- the levels inside a 16-lane row are
// DPP moves (row_ror:8; row_half_mirror, which pairs lane i with 7 - i: any pairing of the low with the high half of a group serves a
// sum), the last two quad_perm adds.
__device__ __forceinline__ float wave_sum16(float (&a)[16], int lane) {
#pragma unroll
  for (int v = 0; v < 8; ++v) {   // lanes 0..31 keep a[v], lanes 32..63 keep a[v + 8]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[v]), __float_as_uint(a[v + 8]), false, false);
    a[v] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {   // even rows keep a[v], odd rows a[v + 4]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[v]), __float_as_uint(a[v + 4]), false, false);
    a[v] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float keep = hi ? a[v + 2] : a[v], send = hi ? a[v] : a[v + 2];
      a[v] = keep + dpp_f<0x128>(send);   // row_ror:8
    }
  }
  {
    const bool hi = (lane & 4) != 0;
    const float keep = hi ? a[1] : a[0], send = hi ? a[0] : a[1];
    a[0] = keep + dpp_f<0x141>(send);     // row_half_mirror
  }
  float r = a[0];
  r += dpp_f<0x4E>(r);   // quad_perm [2,3,0,1]
  r += dpp_f<0xB1>(r);   // quad_perm [1,0,3,2]
  return r;
}

// sum_{w < n} p[w * stride] with 16 loads in flight; fixed order (four round-robin partial sums, combined pairwise)
__device__ __forceinline__ float strided_sum(const float* p, int stride, int n) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int w = 0; w < n; w += 16) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = p[(long long)min(w + q, n - 1) * stride];   // clamped, unconditional
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
      a0 += (w + q < n) ? v[q] : 0.f;
      a1 += (w + q + 1 < n) ? v[q + 1] : 0.f;
      a2 += (w + q + 2 < n) ? v[q + 2] : 0.f;
      a3 += (w + q + 3 < n) ? v[q + 3] : 0.f;
    }
  }
  return (a0 + a1) + (a2 + a3);
}

// ---- fused tail of the folded-encoder ELBO step (runs inside the chain-rule launch, encoder_fused.hip) -------------------------
// Elements of [lo2, hi2) carry their OWN step count (pyro.optim keeps one torch.optim.Adam per parameter: the label heads of the
// cvs / challenge families see their first non-None gradient one SVI step later than everything else, training_cvs.py:236-249):
// step_size2 / sqrt_bc2_2 are their bias corrections, skip2 != 0 leaves them untouched (their step count is still 0).
struct AdamK { float *p, *m, *v; float step_size, one_minus_b1, b2, one_minus_b2, sqrt_bc2, eps; int lo2, hi2, skip2; float step_size2, sqrt_bc2_2; };
// torch.optim.Adam single-tensor formulas (see adam_kernel, misc_kernels.hip)
// Returns the parameter's value after the step.  through != 0: the new value is stored agent-scope (sc1, write-through) -- it is handed to
// other workgroups of the same launch (the in-launch fold of enc_chain_kernel reads lin.bias that way).
__device__ __forceinline__ float adam_apply(const AdamK& a, int i, float g, int through = 0) {
  const bool r2 = i >= a.lo2 && i < a.hi2;
  if (r2 && a.skip2) return a.p[i];
  float mi = a.m[i], vi = a.v[i];
  mi = mi + a.one_minus_b1 * (g - mi);
  vi = vi * a.b2 + a.one_minus_b2 * g * g;
  const float denom = sqrtf(vi) / (r2 ? a.sqrt_bc2_2 : a.sqrt_bc2) + a.eps;
  const float pn = a.p[i] - (r2 ? a.step_size2 : a.step_size) * (mi / denom);
  if (through) __hip_atomic_store(a.p + i, pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else a.p[i] = pn;
  a.m[i] = mi;
  a.v[i] = vi;
  return pn;
}
struct AdamHost { float *p, *m, *v; float lr, b1, b2, eps; int64_t step, n; int lo2 = 0, hi2 = 0; int64_t delta2 = 0; };
inline AdamK make_adamk(const AdamHost* a) {
  AdamK k{};
  if (a && a->p) {
    const double bc1 = 1.0 - pow((double)a->b1, (double)a->step), bc2 = 1.0 - pow((double)a->b2, (double)a->step);
    k.p = a->p; k.m = a->m; k.v = a->v;
    k.step_size = (float)((double)a->lr / bc1); k.one_minus_b1 = 1.0f - a->b1; k.b2 = a->b2;
    k.one_minus_b2 = 1.0f - a->b2; k.sqrt_bc2 = (float)sqrt(bc2); k.eps = a->eps;
    k.lo2 = a->lo2; k.hi2 = a->hi2; k.step_size2 = k.step_size; k.sqrt_bc2_2 = k.sqrt_bc2;
    if (a->hi2 > a->lo2) {
      const int64_t s2 = a->step + a->delta2;
      k.skip2 = s2 < 1 ? 1 : 0;
      if (s2 >= 1) {
        k.step_size2 = (float)((double)a->lr / (1.0 - pow((double)a->b1, (double)s2)));
        k.sqrt_bc2_2 = (float)sqrt(1.0 - pow((double)a->b2, (double)s2));
      }
    }
  }
  return k;
}
// Where every flat-gradient element outside lin.weight comes from.  ODE half: partial slabs of the stage-1 rider blocks of the GEMM
// launch (element 0 = loss).  conv taps: Hc per-m rows of the chain kernel.  lin.bias = ones-column of G = g_pre^T [x|1].  Head layers =
// rows of glat[:, 0:L]^T [hid|1] (z_loc) and glat[:, 64:64+L]^T [hid|1] (z_log_scale).  Elements [n_params, n_total) (parameters the
// caller appended) have zero gradient and only take the Adam step.
struct TailK {
  const float *gslabs, *gslabs_loc, *gslabs_ls, *conv_slabs;
  const float* ode_part; int ode_stride, ode_n;
  int part_lo, part_hi;   // the part rows hold [loss | flat elements [part_lo, part_hi)]; [ode_begin, n_params) outside it: zero gradient
  float* loss_out;
  int gsplit, Hc, L, CT, n_cv;                       // n_cv = F*C*K + F
  int conv_w, lin_w, lin_b, zloc_w, zloc_b, zls_w, zls_b, ode_begin, n_params, n_total;
  float* grads;
  AdamK ad;
  unsigned int* counter;   // arrivals of the chain blocks (zeroed by the fold kernel at the start of the step)
  // FOLD-NEXT (enc_chain_kernel): the launch also folds W_eff / b_eff / rowsum / w' / the likelihood-scale table of the UPDATED weights
  // for the next step, so that step needs no fold launch.  done: arrival counter of all blocks of the launch (never reset: the launch's
  // target is done_target); cstd_off: flat offset of constant_std (its rider threads write the table), sigtab [4][CT], gauss
  int fold_next;
  int n_riders;            // > 0: rider blocks of the launch (fewer than one thread per element: they loop)
  unsigned int* done;
  unsigned int done_target;
  int cstd_off, gauss;
  float* sigtab;
};
__device__ __forceinline__ void tail_loss(const TailK& k) {
  if (!k.loss_out) return;
  double acc = 0.0;
  for (int w = 0; w < k.ode_n; ++w) acc += (double)k.ode_part[(long long)w * k.ode_stride];   // fixed order
  k.loss_out[0] = (float)acc;
}
// element i of [0, lin_w) (conv) or [lin_b, n_total): gradient, write, optional Adam
template <bool FN = false>   // FN: the in-launch fold form of the chain launch (its riders publish what the fold reads, and write the scale table)
__device__ __forceinline__ void tail_element(const TailK& k, int i) {
  float g = 0.f;
  if (i >= k.ode_begin && i < k.n_params) {
    if (i >= k.part_lo && i < k.part_hi) g = strided_sum(k.ode_part + 1 + (i - k.part_lo), k.ode_stride, k.ode_n);
  } else if (i < k.lin_w) {
    g = strided_sum(k.conv_slabs + (i - k.conv_w), k.n_cv, k.Hc);
  } else if (i < k.n_params) {
    const int Hc = k.Hc, GN = k.CT + 1, HN = Hc + 1;
    if (i < k.zloc_w) {
      g = strided_sum(k.gslabs + (long long)(i - k.lin_b) * GN + k.CT, Hc * GN, k.gsplit);
    } else {
      int row, col;
      const float* src = k.gslabs_loc;
      if (i < k.zloc_b) { const int e = i - k.zloc_w; row = e / Hc; col = e - row * Hc; }
      else if (i < k.zls_w) { row = i - k.zloc_b; col = Hc; }
      else if (i < k.zls_b) { const int e = i - k.zls_w; row = e / Hc; col = e - row * Hc; src = k.gslabs_ls; }
      else { row = i - k.zls_b; col = Hc; src = k.gslabs_ls; }
      g = strided_sum(src + (long long)row * HN + col, k.L * HN, k.gsplit);
    }
  }
  if (i < k.n_params) k.grads[i] = g;
  if (k.ad.p) {
    const float pn = adam_apply(k.ad, i, g, FN ? 1 : 0);
    if (FN && k.sigtab && i >= k.cstd_off && i < k.cstd_off + k.CT) {
      // likelihood scales of the NEXT step (what weff_kernel's extra blocks compute at the start of a step: same functions, same bits)
      const int e = i - k.cstd_off;
      const float sig = softplusf(pn);
      k.sigtab[e] = sig;
      k.sigtab[k.CT + e] = 1.0f / sig;
      k.sigtab[2 * k.CT + e] = k.gauss ? logf(sig) : logf(2.f * sig);
      k.sigtab[3 * k.CT + e] = 1.f - expf(-sig);
    }
  }
}

// Stage 1 of the slab reduction: [n][stride] -> [G][stride] partial sums.  One 256-thread block = 64 elements x 4 sub-groups; every
// thread keeps 4 independent loads in flight; summation order is fixed (slab index) => bitwise reproducible.  s_p: 256 floats.
// zr_*: rows [0, zr_rows) carry nothing in columns [zr_lo, zr_hi) (dopri5 training: the scorer's rows and the solver-side range, which only
// the reverse sweep's rows fill) -- those reads are skipped (4096 x 12.8 KB of zeros at BASELINE config[2])
struct Stage1 { const float* slabs; int stride, n, count, per; float* out; int zr_rows, zr_lo, zr_hi; };
// One block = SLODE_S1_COLS columns x one group of rows; wave q takes rows w0 + q, w0 + q + 4, ...; a lane reads FOUR consecutive columns per
// row (16 bytes: a wave's request is 1 KB of one row -- with 4 bytes per lane the 52 MB of config[2]'s scorer rows were 203 k requests of
// 256 B; the sums per column run in the same order as before).  s_p: 4 x SLODE_S1_COLS floats.  stride: a multiple of 4, rows 16-byte aligned.
#define SLODE_S1_COLS 256
__device__ __forceinline__ void slab_stage1_block(const Stage1& f, int bx, int g, float* s_p) {
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const int lane64 = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = bx * SLODE_S1_COLS + lane64 * 4;
  const int w0 = g * f.per, w1 = min(f.n, w0 + f.per);
  f4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (e < f.count) {
    const float* p = f.slabs + e;
    // first row of this sub-group that carries column e + c: rows below zr_rows hold nothing (not even zeros) in columns [zr_lo, zr_hi)
    int st[4], w = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      st[c] = w0 + q;
      if (e + c >= f.zr_lo && e + c < f.zr_hi && st[c] < f.zr_rows) st[c] += ((f.zr_rows - st[c] + 3) >> 2) << 2;
      w = min(w, st[c]);
    }
    auto row = [&](int r) __attribute__((always_inline)) {
      f4_t v = *reinterpret_cast<const f4_t*>(p + (long long)r * f.stride);
      v.x = r >= st[0] ? v.x : 0.f; v.y = r >= st[1] ? v.y : 0.f; v.z = r >= st[2] ? v.z : 0.f; v.w = r >= st[3] ? v.w : 0.f;   // (selects: what is masked may be anything)
      return v;
    };
    for (; w + 12 < w1; w += 16) {
      const f4_t v0 = row(w), v1 = row(w + 4), v2 = row(w + 8), v3 = row(w + 12);
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; w < w1; w += 4) a0 += row(w);
  }
  const f4_t t = (a0 + a1) + (a2 + a3);
  *reinterpret_cast<f4_t*>(s_p + q * SLODE_S1_COLS + lane64 * 4) = t;
  __syncthreads();
  for (int c = threadIdx.x; c < SLODE_S1_COLS; c += 256) {
    const int ec = bx * SLODE_S1_COLS + c;
    if (ec < f.count)
      f.out[(long long)g * f.stride + ec] = (s_p[c] + s_p[SLODE_S1_COLS + c]) + (s_p[2 * SLODE_S1_COLS + c] + s_p[3 * SLODE_S1_COLS + c]);
  }
}

// Deterministic block-wide sum (fixed tree).  `scratch` needs blockDim.x/64 floats.  Result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < nw; ++w) r += scratch[w];
  return r;
}

// ---- launch wrappers implemented in the kernel translation units ----------------------------------------
struct OdeLaunch {
  slode_shape s;
  slode_layout lay;
  const float* params;
  const float* times;    // [T]
  const float* stage_t;  // [R*(T-1)+1]
  // fused-ELBO inputs
  const float* obs;
  int64_t sb, sc, st;    // element strides of the logical [B,C,T] observations
  const float* u;        // [B,n_u]
  const float* eps;      // [B,L]
  const float* loc;      // [B,L]  (encoder output)
  const float* scale;    // [B,L]
  // solve-only inputs
  const float* z_in;     // [B,L]  (when loc == nullptr)
  const float* gx_in;    // [B,T,S] upstream gradient (solve-bwd); nullptr => likelihood gradient
  // outputs
  float* x_out;          // [B,T,S] or nullptr
  float* z_out;          // [B,L] or nullptr
  float* g_loc;          // [B,L]  (dLoss/dloc)   | solve-bwd: g_z
  float* g_scale;        // [B,L]  or nullptr
  float* slabs;          // [grid][slab_stride] per-workgroup partial results (slot 0 = loss, then ode segment)
  int slab_stride;
  int grid;
  int backward;          // 0: forward only
  int with_ll;           // 1: likelihood + latent terms (ELBO); 0: pure ODE solve
  const float* sigtab = nullptr;   // optional [4][C*T]: softplus(constant_std) | its reciprocal | log(scale) (Gauss) or log(2 scale) (ALD) |
                                   // 1 - exp(-scale) = softplus'; parameter-only, computed once per step instead of once per trajectory
  // folded-encoder ELBO step: the kernel also runs the encoder heads + tanh backward and writes g_pre / glat ([B][64] each)
  const float* enc_hid = nullptr;
  float* g_pre = nullptr;
  float* glat = nullptr;   // [B][128] = [g_loc (L) | pad to 64 | g_scale * scale (L) | pad]
  // externally solved trajectories (dopri5 training; generic instantiation only): the kernel scores x_ext instead of its own solve,
  // writes dLoss/dx to gx_out and back-propagates nothing through a solver of its own (the adaptive solver's reverse sweep adds its
  // share of the latent gradient to g_loc / g_scale afterwards)
  const float* x_ext = nullptr;   // [B][T][S]
  float* gx_out = nullptr;        // [B][T][S]
  int ext_skip = 0;               // 1: the scorer does not write the zeros of the slab row's solver-side range [init net | dynamics] --
                                  // only when the consumer of the rows (stage 1 of the fused tail, Stage1::zr_*) does not read them
  int force_loop = 0, force_generic = 0, alg = 0;   // handle flags (slode_ctx)
  int enc_fuse = 0;      // 1: the kernel runs the encoder forward of its trajectories itself (ENCF; slode_ode_can_fuse_encoder), from
  const float *enc_weff = nullptr, *enc_beff = nullptr;   // the fold launch's W_eff [Hc][C*T] / b_eff [Hc]; obs must be dense rows (sb == C*T)
  float* enc_hid_out = nullptr;                          // saved tanh [B][Hc]
  int pack = 0;          // 4: four trajectories per workgroup where the shape has such an instantiation (ode_kernel.hip, PK)
  RngK rng{};            // on: eps is drawn in the kernel (eps may be NULL)
  LabelSrc lab{};        // n > 0: the label columns come from separate tensors (u may be NULL)
};
hipError_t slode_launch_ode(const OdeLaunch& a, hipStream_t stream, char* err, size_t errlen);
size_t slode_ode_lds_bytes(const slode_shape& s, int nthreads, bool one = false);
int slode_ode_threads(const slode_shape& s);
bool slode_ode_can_fuse_encoder(const slode_shape& s, bool bwd, int grid);

struct EncLaunch {
  slode_shape s;
  slode_layout lay;
  const float* params;
  const float* obs;
  int64_t sb, sc, st;
  float* loc;
  float* scale;
  float* pooled;  // [B, F*n_pool]
  float* hid;     // [B, Hc]
};
hipError_t slode_launch_enc_fwd(const EncLaunch& a, hipStream_t stream);

struct EncBwdLaunch {
  slode_shape s;
  slode_layout lay;
  const float* params;
  const float* obs;
  int64_t sb, sc, st;
  const float* scale;
  const float* pooled;
  const float* hid;
  const float* g_loc;
  const float* g_scale;
  float* g_pre;        // [B, 64]   workspace: dLoss/d(lin pre-activation)
  float* slabs_small;  // [grid_small][small_stride]: conv_w, conv_b, lin_b, zloc_w, zloc_b, zls_w, zls_b partials
  int small_stride;
  int grid_small;
  float* slabs_lin;    // [splitk][Hc*FQ] partial lin.weight gradients
  int splitk;
};
hipError_t slode_launch_enc_bwd(const EncBwdLaunch& a, hipStream_t stream);
int slode_enc_small_count(const slode_shape& s);   // floats per small slab
int slode_enc_bwd_grid(const slode_shape& s);
int slode_enc_lin_splitk(const slode_shape& s);

// Folded encoder path (encoder_fused.hip)
struct FoldLaunch {
  slode_shape s;
  slode_layout lay;
  const float* params;
  const float* x;        // dense observation rows [B][C*T]
  int t_major;           // 1: [B,T,C] contiguous, 0: [B,C,T] contiguous
  float *weff, *rowsum, *wprime, *beff;
  float *loc, *scale, *hid;
  const float *g_loc, *g_scale;
  float *g_pre, *small_slabs; int small_stride;
  const float* gslabs; int n_gslabs;
  float *g_lin_w, *conv_slabs;
  unsigned int* counter = nullptr;   // zeroed by the fold kernel; arrivals of the chain blocks
  float* sigtab = nullptr;           // [4][C*T] likelihood-scale table of this step (see OdeLaunch::sigtab), written by extra fold blocks
  const TailK* tail = nullptr;       // chain launch also finishes the whole flat gradient (+ loss, + optional Adam)
  int skip_enc = 0;                  // 1: fold only (the ODE kernel runs the encoder forward itself)
  int fold_skip = 0;                 // 1: encoder forward only (the workspace already holds the fold of the current weights)
};
// blocks of the chain launch (chain blocks + riders): what the in-launch fold's arrival target counts, and whether they are all resident
int slode_chain_blocks(const slode_shape& s, int n_total, int lin_b, int* n_chain);
int slode_chain_resident_blocks(const slode_shape& s, int num_cu);
hipError_t slode_launch_fold_fwd(const FoldLaunch& a, hipStream_t stream);
hipError_t slode_launch_fold_chain(const FoldLaunch& a, hipStream_t stream);
// The three split-K MFMA products of the fused tail in one launch, each with a ones-column appended to its right-hand matrix:
//   gslabs[s][m < Hc][CT + 1] = g_pre^T [x | 1];  gslabs_loc[s][l < L][Hc + 1] = glat[:, 0:L]^T [hid | 1];  gslabs_ls likewise from
//   glat[:, 64:64+L]   (g_pre: [B][64], glat: [B][128] = [g_loc | pad | g_scale * scale | pad])
// Extra blocks of the same launch run stage 1 of the ODE-slab reduction (ode_n slabs -> *ode_n_out partial slabs in ode_part;
// with few slabs stage 1 is skipped and *ode_part_out = ode_slabs).
hipError_t slode_launch_gemm_tail(const float* g_pre, const float* x, float* gslabs, int Hc, int CT, const float* glat, const float* hid,
                                  float* gslabs_loc, float* gslabs_ls, int L, int B, int splitk, const float* ode_slabs, int ode_stride,
                                  int ode_n, int ode_count, float* ode_part, const float** ode_part_out, int* ode_n_out,
                                  hipStream_t stream, int zr_rows = 0, int zr_lo = 0, int zr_hi = 0);
int slode_fold_small_count(const slode_shape& s);

struct AuxLaunch {
  slode_shape s;
  slode_layout lay;
  const float* params;
  const float *loc, *scale, *eps, *u;
  float *g_loc, *g_scale, *slabs;
  int slab_stride, grid, backward;
  int compact = 0;                   // slab row = [loss | flat range [lay.aux_w1[0], lay.cstd)] instead of the whole ODE segment
  const float* enc_hid = nullptr;    // folded encoder path: the kernel also runs the encoder heads + tanh backward (g_pre, glat)
  float *g_pre = nullptr, *glat = nullptr;
  RngK rng{};
  LabelSrc lab{};
};
hipError_t slode_launch_aux(const AuxLaunch& a, hipStream_t stream);

#define SLODE_REDUCE_GROUPS 16
struct ReduceLaunch {
  slode_shape s;
  slode_layout lay;
  const float* ode_slabs; int ode_stride; int ode_n;       // may be null
  const float* small_slabs; int small_stride; int small_n; // may be null
  const float* lin_slabs; int lin_n;                       // may be null
  float* grads;       // flat gradient (may be null when only the loss is wanted)
  float* loss_out;    // may be null
  int zero_rest;      // also zero grads outside the written segments [0, n_params)
  float* ode_part;    // [SLODE_REDUCE_GROUPS][ode_stride] scratch for the two-stage reduction (may be null)
  float* small_part;  // [SLODE_REDUCE_GROUPS][small_stride] likewise
  int folded;         // 1: folded-encoder families (small = [lin_b..zls_b]; `lin` family = per-m conv slabs at flat offset conv_w)
  float *adam_p = nullptr, *adam_m = nullptr, *adam_v = nullptr;  // optional fused Adam (after the positional members)
  float adam_lr = 0.f, adam_b1 = 0.f, adam_b2 = 0.f, adam_eps = 0.f;
  int64_t adam_step = 0, adam_n = 0;
  int adam_lo2 = 0, adam_hi2 = 0; int64_t adam_delta2 = 0;
};
hipError_t slode_launch_reduce(const ReduceLaunch& a, hipStream_t stream);


hipError_t slode_launch_stage_times(const slode_shape& s, const float* times, float* stage_t, hipStream_t stream);
hipError_t slode_launch_decode_heads(const slode_shape& s, const slode_layout& lay, const float* params,
                                     const float* x, float* mu, float* std_ct, hipStream_t stream);
hipError_t slode_launch_decode_heads_bwd(const slode_shape& s, const slode_layout& lay, const float* params, const float* x, const float* g_mu,
                                         const float* g_std, float* g_x, float* g_heads, float* g_cstd, hipStream_t stream);
hipError_t slode_launch_init_state(const slode_shape& s, const slode_layout& lay, const float* params, const float* z, float* x0, hipStream_t stream);
hipError_t slode_launch_prior_nets(const slode_shape& s, const slode_layout& lay, const float* params, const float* u, float* loc, float* scale,
                                   hipStream_t stream);
hipError_t slode_launch_label_heads(const slode_shape& s, const slode_layout& lay, const float* params, const float* z, float* out, hipStream_t stream);
hipError_t slode_launch_dynamics_eval(const slode_shape& s, const slode_layout& lay, const float* params, float t,
                                      const float* state, const float* z, float* out, hipStream_t stream);
// adaptive solve with step records (training): z = loc + scale * eps is formed in the kernel and written to z_out
// rng.on: eps is drawn by the forward kernel and written to eps_out, which the scorer and the reverse sweep then read as `eps`
struct DopriRec { const float *loc, *scale, *eps; float* z_out; float* rec; int* nrec; int kmax; RngK rng{}; float* eps_out = nullptr;
                  int w64 = 8;   // lanes per trajectory of the forward solve: 8 (dopri5_kernel), 16 / 32 / 64 (dopri5_lpt_kernel); SLODE_DP5_LPT
                  float* tabs = nullptr;   // [rows][slode_dopri5_tab_floats]: the forward kernel's per-workgroup tables, handed to the reverse sweep
};
size_t slode_dopri5_tab_floats(const slode_shape& s);
int slode_dopri5_kmax(const slode_shape& s);
int slode_dopri5_rows(const slode_shape& s);
hipError_t slode_launch_dopri5(const slode_shape& s, const slode_layout& lay, const float* params, const float* times, const float* z,
                               float* x, hipStream_t stream, const DopriRec* rec = nullptr);
hipError_t slode_launch_dopri5_bwd(const slode_shape& s, const slode_layout& lay, const float* params, const float* times, const DopriRec& rec,
                                   const float* gx, float* g_loc, float* g_scale, float* slabs, int slab_stride, int drop_z, float* snap, hipStream_t stream,
                                   const float* enc_hid = nullptr, float* g_pre = nullptr, float* glat = nullptr);   // (rec.tabs: tables of the forward kernel)
hipError_t slode_launch_adam_k(int64_t n, const float* g, const AdamHost& a, hipStream_t stream);
hipError_t slode_launch_adam(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2,
                             float eps, int64_t step, hipStream_t stream);
// data-parallel payload: out[0, total) = [sum of the gsplit split-K partials of G | of G_loc | of G_ls | sum of the ode_n partial rows],
// every sum in fixed order (slode_api.hip: PayloadMap); the gaps between the pieces are written as zeros
hipError_t slode_launch_pack_payload(const float* gslabs, const float* gslabs_loc, const float* gslabs_ls, int gsplit, int Hc, int CT, int L,
                                     const float* ode_part, int ode_stride, int ode_n, int ode_count, float* out, int o_loc, int o_ls, int o_ode,
                                     int total, hipStream_t stream);
// eps_out[B][L] (may be NULL) and / or raw[B][ceil(L/4)][4] Philox words (may be NULL) of one drawing call: misc_kernels.hip
hipError_t slode_launch_rng_fill(const RngK& r, int B, int L, float* eps_out, unsigned int* raw, hipStream_t stream,
                                 const float* loc = nullptr, const float* scale = nullptr);   // loc given: eps_out = loc + scale * eps
