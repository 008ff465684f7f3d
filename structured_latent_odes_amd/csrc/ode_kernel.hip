// Fused latent-ODE solve + ELBO kernel for gfx950 (MI355X).
//
// Replaces, for one minibatch shard, the work the reference does in
//   models/mechanistic_cvs.py:105-238 (model + guide: latent sample, priors, likelihood sites),
//   models/decoders.py:42-54 / 84-91   (Decoder / GaussianDecoder forward),
//   models/blackbox_ode.py:36-47,97-109 (OdeModel.solve_ODE over Dynamics.forward via torchdiffeq.odeint),
// and autograd's backward through all of it (== adjoint_solver=False gradients).
//
// Design: ONE workgroup integrates ONE trajectory (loop-free form, B <= 65,536; a persistent loop beyond), with
// NT = roundup64(T) threads (>= 64 + roundup64(Q*C*S)).
//   * f(t,x) = a(t,z) - d(t,z) * x never reads the state inside the net (blackbox_ode.py:97-109), and the hidden layer is
//     relu(w_t t + u_j(z)): along the monotone stage-time table every unit switches on over a prefix or a suffix.  The 2S head
//     pre-activations are therefore piecewise linear in t with <= H+1 segments: wave 0 finds the switching indices, ranks them and
//     builds a table [segment][value | slope] (P0b, P0c); a stage evaluation is a 5-step search, S ds_read_b128 and 2S fma + sigmoid (P1).
//   * every fixed-grid step collapses to the affine map x' = A x + b; the forward recurrence runs as an affine scan on all waves
//     (P2, block_affine_scan), its adjoint on wave 0 (P4, wave_affine_scan) beside the head-weight gradients of the other waves.
//   * likelihood (asymmetric Laplace x3 quantile heads, or Gaussian), latent log-probs, prior nets and -- proc family -- label heads
//     are fused in (P0a, P0b, P3); the reverse mode of the step coefficients is thread <-> step again (P5) and leaves per-sample
//     gradient rows G[nt][2S] in LDS; the weight-gradient contraction (P6) needs no per-sample x per-unit work: chunked prefix /
//     suffix sums of G evaluated at each unit's switching index.
//   * every gradient element has ONE owner thread and goes straight to the workgroup's slab row (no LDS accumulators, nothing but the
//     loss partial carried in registers across trajectories); slab rows are summed in fixed order by the tail => bitwise reproducible.
//   * folded ELBO step: the encoder heads + tanh backward (g_pre, glat) run here too (P7 / last block of the trajectory), and -- metric
//     shape, loop-free form (ENCF) -- the encoder FORWARD of the workgroup's own trajectory in the set-up: the step has no encoder launch.
//   * no cross-lane exchange touches the LDS pipe: DPP moves inside 16-lane rows (the affine scans put a state component's eight chunks
//     into half a row), v_permlane16_swap / v_permlane32_swap across rows and halves (slode_common.h).
//   * long latents (L >= 32, shape-specialised): only the parameters the solver phases re-read live in LDS; the rest is staged in the
//     idle work block for P0 and read from global memory in P7 (ode_cold_global) -- 4 workgroups per CU instead of 2.
#include "slode_common.h"
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(4))) float* cptr;  // uniform loads => s_load + SGPR operands
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifdef SLODE_STAMPS  // diagnostic build only: phase boundaries of workgroup 0 in 10 ns ticks (s_memrealtime)
__device__ unsigned long long g_stamps_ode[32];
__device__ unsigned long long g_stamps_ode_late[32];   // the same phase boundaries for workgroup 1000 (last residency slot of its CU)
__device__ unsigned long long g_wg_span[2 * 4096];   // [start, end] of every workgroup
__device__ unsigned int g_wg_hw[2 * 4096];            // [HW_ID, XCC_ID] of every workgroup's wave 0
#define STAMP(i) do { if (tid_outer == 0) { const unsigned long long t_ = wall_clock64(); if (vblk == 0) g_stamps_ode[i] = t_; if (vblk == 1000) g_stamps_ode_late[i] = t_; \
    if ((i) == 0 && vblk < 4096) { g_wg_span[2 * vblk] = t_; g_wg_hw[2 * vblk] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g_wg_hw[2 * vblk + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } if ((i) == 11 && vblk < 4096) g_wg_span[2 * vblk + 1] = t_; } } while (0)
extern "C" int slode_debug_stamps_ode(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_ode), sizeof(unsigned long long) * 32);
}
extern "C" int slode_debug_stamps_ode_late(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_ode_late), sizeof(unsigned long long) * 32);
}
extern "C" int slode_debug_wg_hw(unsigned int* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_hw), sizeof(unsigned int) * 2 * n);
}
extern "C" int slode_debug_wg_span(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_span), sizeof(unsigned long long) * 2 * n);
}
#else
// Product build: the phase boundaries rotate the issue priority between the workgroups that share a CU.  The CU arbitrates
// oldest-first, so of the (up to four) co-resident trajectories the youngest only gets the issue slots the others leave and finishes
// last (round 1: 27 / 31 / 36 / 41 us); alternating high / low priority by (phase + residency slot) parity evens their progress.  Wave 0
// (every serial stretch) stays one level above its workgroup's bulk waves.  Round 3 measured two more schemes against this one: a rotation
// over four levels by (phase + slot) mod 4 -- 27.6 us against 27.5, no better, and the four-way branch of the macro itself costs 0.4 us --
// a static level = residency slot (youngest highest): 28.4 us; and a FEEDBACK scheme (level = how far the workgroup lags the solo
// timeline of the phase boundary, s_memrealtime against a nominal table): 28.6 - 29.9 us over the thresholds tried.  The stamps of a
// last-slot workgroup (tools/stamps.py, second column) show where it loses its 7 us -- P1, P2 and P4, the phases in which every wave of
// every co-resident workgroup is busy -- so what it lacks is issue slots and LDS cycles, not priority.  Staggering the START of the
// co-resident workgroups (slot s idles s x 0.4 .. 2.4 us behind its set-up loads, so that their all-wave phases coincide less) does not
// help either: the kernel gets longer by about 0.6 x the last slot's delay (27.3 -> 27.2 / 29.0 / 28.6 / 30.1 / 31.8 us).
#define STAMP(i) do { if (((i) + prio_slot) & 1) { if (tid_outer < 64) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); } \
                      else { if (tid_outer < 64) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } } while (0)
#endif

namespace {


struct OdeK {
  int B, T, C, L, nu, ng, method, R, nt, Q, gauss;
  int uses_next;  // rk4: k4 is evaluated at t1 == next step's first stage
  slode_group grp[SLODE_MAX_GROUPS];
  float tau[SLODE_MAX_HEADS];
  const float *ploc_w[SLODE_MAX_GROUPS], *ploc_b[SLODE_MAX_GROUPS], *pls_w[SLODE_MAX_GROUPS], *pls_b[SLODE_MAX_GROUPS];
  const float *w1, *b1, *w2, *b2, *wh, *bh, *wg, *bg, *wd, *bd, *head[SLODE_MAX_HEADS], *cstd;
  const float* sigtab;   // optional [4][C*T]: softplus(constant_std) | 1/scale | log(scale) or log(2 scale) | softplus' (OdeLaunch::sigtab)
  // offsets relative to lay.ode_begin (accumulator / slab index = 1 + offset)
  int o_ploc_w[SLODE_MAX_GROUPS], o_ploc_b[SLODE_MAX_GROUPS], o_pls_w[SLODE_MAX_GROUPS], o_pls_b[SLODE_MAX_GROUPS];
  int o_w1, o_b1, o_w2, o_b2, o_wh, o_bh, o_wg, o_bg, o_wd, o_bd, o_head[SLODE_MAX_HEADS], o_cstd;
  int nseg;
  int n_aux, U;        // label heads scored inside the main loss (proc family)
  int n_aux_lds;       // LDS rows reserved for them (same value on host and device)
  int stage_encw;      // 1: P7 stages zloc_w | zls_w into LDS for the fused encoder-head backward
  float aux_mult;
  slode_aux aux[SLODE_MAX_AUX];
  int o_aux_w1[SLODE_MAX_AUX], o_aux_b1[SLODE_MAX_AUX], o_aux_w2[SLODE_MAX_AUX], o_aux_b2[SLODE_MAX_AUX], o_aux_c[SLODE_MAX_AUX];
  int npar;            // floats of the segment staged in LDS: [ode_begin, cstd) = priors | init net | dynamics | heads
  const float* pseg;   // params + ode_begin
  const float *times, *stage_t, *obs, *u, *eps, *loc, *scale, *z_in, *gx_in;
  long long sb, sc, st;
  float *x_out, *z_out, *g_loc, *g_scale, *slabs;
  int slab_stride, backward, with_ll;
  // fused encoder-head backward (folded encoder path): g_pre[b][m] = (1 - hid^2) * (zloc_w^T g_loc + zls_w^T (g_scale * scale))
  const float *enc_hid, *enc_zloc_w, *enc_zls_w;
  const float *enc_weff, *enc_beff, *enc_zloc_b, *enc_zls_b;   // ENCF: folded encoder weights of this step (fold launch), head biases
  float* enc_hid_out;                                          // ENCF: saved tanh [B][Hc]
  float *g_pre, *glat;   // g_pre [B][64]; glat [B][128], row = [g_loc (L) | pad to 64 | g_scale * scale (L) | pad]
  // externally solved trajectories (dopri5 training, generic instantiation only; see OdeLaunch)
  const float* x_ext;
  float* gx_out;
  int ext_skip;   // scorer: leave the solver-side range [init net | dynamics] of the slab row unwritten (OdeLaunch::ext_skip)
  int Hc;
  // gradient-segment elements no phase of this launch owns (zero gradient: written as zeros once per workgroup), relative to ode_begin
  int nz, zlo[8], zhi[8];
  RngK rng;       // on: the guide's noise is drawn here (Philox, slode_common.h) instead of read from eps
  LabelSrc lab;   // n > 0: label columns from the loader's separate tensors instead of the dense u matrix
};

struct LdsMap {  // offsets in floats
  int ts, sig, A, x, lam, st, stn, ax, ct, gm, hp, tau, ps, ord, sgs, ewt, epre, epre0, esw, ms, sf, par, uu, auxh, auxd, auxgo, z, gzl, gpl, gls, u, wt, pre0,
      hid0, x0, go, gp0, gu, red, pf, meta, total;
};

__host__ __device__ constexpr int pad4(int n) { return (n + 3) & ~3; }
// samples by which NQ = nthreads / 2S chunks of ceil(nt / NQ) samples overhang the stage-time table; 0 when that is more than 8 rows
__host__ __device__ constexpr int ode_pad_rows(int nt, int nthreads, int S) {
  const int nq = nthreads / (2 * S), cl = (nt + nq - 1) / nq, over = nq * cl - nt;
  return over <= 8 ? over : 0;
}
__host__ __device__ constexpr bool ode_full_chunks(int nt, int nthreads, int S) {
  const int nq = nthreads / (2 * S), cl = (nt + nq - 1) / nq;
  return nq * cl - nt <= 8;
}
__host__ __device__ inline int imax2(int a, int b) { return a > b ? a : b; }
// Long latents (L >= 32: the proc family's 50): only the parameters the solver phases re-read stay in LDS -- [init b1 | W2 | b2] and
// [dyn b_h | W_g | b_g | W_d | b_d] -- the prior nets, the L-wide rows of the init net and of the hidden layer and the label heads are
// read from global memory (L2 / L1 hits: every workgroup reads the same 20 KB) by the few phases that touch them once per trajectory.
// 22 KB of LDS per workgroup less: 4 workgroups per CU instead of 2 for BASELINE config[2]'s shapes.
__host__ __device__ constexpr bool ode_cold_global(int L_) { return L_ >= 32; }
__host__ __device__ constexpr int ode_hot_r1(int S, int H) { return H + S * H + S; }            // b1 | W2 | b2
__host__ __device__ constexpr int ode_hot_r2(int S, int H) { return H + 2 * (S * H + S); }      // b_h | W_g | b_g | W_d | b_d
__host__ __device__ constexpr int ode_hot_floats(int S, int H) { return ode_hot_r1(S, H) + ode_hot_r2(S, H); }

// `one`: loop-free form (one workgroup per trajectory): softplus(constant_std) stays in registers, no s_sig.
// The block A | x | lam | st is one work region: besides the scan operands it holds, at different times, the piecewise-linear table of
// the dynamics heads (P1), the stage-0 exchange rows / dLoss/dmu (P1, P3), the per-sample gradient rows G[nt][2S] (P5 -> P6) and the
// staged encoder head weights (P7).
// `scorer`: the shape-specialised solver-free scorer (ALG 3) touches neither the scan operands A / lam, nor the sample rows, the chunk
// sums, GM | GT or the event arrays: they get no space (their offsets alias what follows; nothing reads or writes them).
__host__ __device__ inline LdsMap lds_map(int T, int S, int H, int C, int L, int Q, int nt, int npar, int nthreads, int naux, bool one,
                                          bool scorer = false) {
  LdsMap m;
  int o = 0;
  if (scorer) {
    m.ts = o; o += pad4(nt);
    m.sig = o; o += one ? 0 : pad4(C * T);
    m.ax = pad4(T * S);
    m.A = o; m.lam = o;                 // (unused)
    m.x = o; o += m.ax;
    m.st = o; m.stn = pad4(Q * C * T); o += m.stn;   // dLoss/dmu [Q*C][T]
    m.ct = o; m.gm = o;                 // (unused)
    m.hp = o; o += pad4(4 * Q * C * S);
    m.tau = m.ps = m.ord = m.sgs = m.ewt = m.epre = m.epre0 = m.esw = m.ms = m.sf = o;   // (unused)
  } else {
  // P6 reads the sample rows in chunks of CL = ceil(nt / NQ) samples: when the last chunk overhangs the table by a few samples, zero
  // pad rows (and pad stage times) make every chunk full, and the contraction needs no bounds at all (ode_pad_rows)
  const int padr = ode_pad_rows(nt, nthreads, S);
  m.ts = o; o += pad4(nt + padr);
  m.sig = o; o += one ? 0 : pad4(C * T);
  const int tabn = (H + 1) * 4 * S;   // table rows [H+1][V (2S) | slope (2S)]
  m.ax = imax2(pad4(T * S), pad4((tabn + 2) / 3));
  m.A = o; o += m.ax;
  m.x = o; o += m.ax;
  m.lam = o; o += m.ax;
  int stn = imax2(((2 * S + 4) & ~3) * T, Q * C * T);   // stage-0 exchange rows [T][SP]; dLoss/dmu [Q*C][T]
  stn = imax2(stn, (nt + padr) * 2 * S - 3 * m.ax);     // G rows (+ pad rows) overlay the whole block
  m.st = o; m.stn = pad4(stn); o += m.stn;
  m.ct = o; o += pad4((nthreads / (2 * S) + 1) * 4 * S);   // chunk sums [NQ (+1: total)][sum g (2S) | sum g t (2S)]
  m.gm = o; o += 2 * 2 * S * 32;                        // GM | GT: [2][2S][32] (unit H = the constant-1 unit)
  m.hp = o; o += pad4(4 * Q * C * S);                   // head-weight partials [hsplit][Q*C*S]
  m.tau = o; o += 32;
  m.ps = o; o += 32;
  m.ord = o; o += 32;
  m.sgs = o; o += 32;
  m.ewt = o; o += 32;
  m.epre = o; o += 32;
  m.epre0 = o; o += 32;
  m.esw = o; o += 32 * 2 * S;                           // events in table order: sign * W[c][unit], [32][2S]
  m.ms = o; o += 32;
  m.sf = o; o += 32;
  }
  m.uu = o; o += SLODE_MAX_NU;
  m.z = o; o += pad4(L);
  m.gzl = o; o += pad4(L);
  m.gpl = o; o += pad4(2 * L);  // [d(-log p)/d prior loc | eps]
  m.gls = o; o += pad4(2 * L);  // [d(-log p)/d prior log-scale | guide scale]
  m.u = o; o += 32;
  m.wt = o; o += 32;
  m.pre0 = o; o += 32;
  m.hid0 = o; o += 32;
  m.x0 = o; o += 8;
  m.go = o; o += 8;
  m.gp0 = o; o += 32;
  m.gu = o; o += 32;
  m.red = o; o += 16;
  m.pf = o; o += 3 * pad4(L);
  m.meta = o; o += pad4(L) * 8;   // per latent dim: prior-net offsets (ints), see setup
  // everything above depends on (T, S, C, L, Q, method, nthreads) only: compile-time offsets in the shape-specialised instantiations
  m.auxh = o; o += naux * 32;   // label heads scored in the main loss only
  m.auxd = o; o += naux * 32;
  m.auxgo = o; o += pad4(naux * 12);
  m.par = o; o += pad4(npar);      // [priors | init net | dynamics | heads]: the gradient of every element goes straight to the slab
  m.total = o;
  return m;
}

// a(t), d(t): models/blackbox_ode.py:97-109 with the z-part of the hidden pre-activation (s_u) hoisted.
// The 2*S*H head weights arrive as SGPR operands (s_load through the constant address space), ONE ROW AHEAD of the row being
// accumulated: with all 2*S rows in flight at once the scheduler overflowed the ~100 SGPRs and spilled them through VGPR lanes
// (v_writelane / v_readlane: +22 % vector instructions in this phase).
template <int S, int H>
__device__ __forceinline__ void eval_ad(float t, const float* __restrict__ s_wt, const float* __restrict__ s_u,
                                        cptr wg, cptr bg, cptr wd, cptr bd, float (&a)[S], float (&d)[S]) {
  // keep the weight s_loads local to this call: hoisting them across the trajectory loop costs >100 SGPRs
  asm volatile("" : "+s"(wg), "+s"(wd), "+s"(bg), "+s"(bd));
  const float* wt = (const float*)__builtin_assume_aligned(s_wt, 16);
  const float* uu = (const float*)__builtin_assume_aligned(s_u, 16);
  float h[H];
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = fmaxf(fmaf(wt[j], t, uu[j]), 0.f);
  float bias[2 * S];
#pragma unroll
  for (int s = 0; s < S; ++s) { bias[s] = bg[s]; bias[S + s] = bd[s]; }
  float wr[2][H];
#pragma unroll
  for (int j = 0; j < H; ++j) wr[0][j] = wg[j];
  float out[2 * S];
#pragma unroll
  for (int r = 0; r < 2 * S; ++r) {   // row r of [W_g; W_d]
    if (r + 1 < 2 * S) {
      cptr nxt = (r + 1 < S) ? wg + (r + 1) * H : wd + (r + 1 - S) * H;
      asm volatile("" : "+s"(nxt));   // the row's loads cannot be issued before this point
#pragma unroll
      for (int j = 0; j < H; ++j) wr[(r + 1) & 1][j] = nxt[j];
    }
    float acc = bias[r];
#pragma unroll
    for (int j = 0; j < H; ++j) acc = fmaf(wr[r & 1][j], h[j], acc);
    asm volatile("" : "+v"(acc));   // row r is finished before row r+2's pointer is laundered: at most two rows of SGPRs live
    out[r] = acc;
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    a[s] = sigmoidf_fast(out[s]);
    d[s] = sigmoidf_fast(out[S + s]);
  }
}

// Step coefficients x' = A x + b for one state component (tests/kernel_math.py::step_coeffs).
// a*/d*: stage values; for rk4 index 3 is the next step's first stage.
__device__ __forceinline__ void step_fwd(int method, float h, const float a[4], const float d[4], float& A, float& b) {
  if (method == SLODE_EULER) {
    A = 1.f - h * d[0];
    b = h * a[0];
  } else if (method == SLODE_MIDPOINT) {
    const float m = 1.f - 0.5f * h * d[0], c = 0.5f * h * a[0];
    A = 1.f - h * d[1] * m;
    b = h * (a[1] - d[1] * c);
  } else {
    const float h3 = h * (1.0f / 3.0f);
    const float p1 = a[0], q1 = -d[0];
    const float c2 = h3 * p1, m2 = 1.f + h3 * q1;
    const float p2 = a[1] - d[1] * c2, q2 = -d[1] * m2;
    const float c3 = h * (p2 - p1 * (1.0f / 3.0f)), m3 = 1.f + h * (q2 - q1 * (1.0f / 3.0f));
    const float p3 = a[2] - d[2] * c3, q3 = -d[2] * m3;
    const float c4 = h * (p1 - p2 + p3), m4 = 1.f + h * (q1 - q2 + q3);
    const float p4 = a[3] - d[3] * c4, q4 = -d[3] * m4;
    const float G = h * 0.125f;
    A = 1.f + G * (q1 + 3.f * (q2 + q3) + q4);
    b = G * (p1 + 3.f * (p2 + p3) + p4);
  }
}

// Reverse mode of step_fwd: (gA, gb) -> ga[r], gd[r] (tests/kernel_math.py::step_coeffs_bwd).
__device__ __forceinline__ void step_bwd(int method, float h, const float a[4], const float d[4], float gA, float gb,
                                         float ga[4], float gd[4]) {
  if (method == SLODE_EULER) {
    gd[0] = -h * gA;
    ga[0] = h * gb;
    ga[1] = gd[1] = ga[2] = gd[2] = ga[3] = gd[3] = 0.f;
  } else if (method == SLODE_MIDPOINT) {
    const float m = 1.f - 0.5f * h * d[0], c = 0.5f * h * a[0];
    ga[1] = h * gb;
    gd[1] = -h * (m * gA + c * gb);
    const float gm = -h * d[1] * gA, gc = -h * d[1] * gb;
    gd[0] = -0.5f * h * gm;
    ga[0] = 0.5f * h * gc;
    ga[2] = gd[2] = ga[3] = gd[3] = 0.f;
  } else {
    const float h3 = h * (1.0f / 3.0f), G = h * 0.125f;
    const float p1 = a[0], q1 = -d[0];
    const float c2 = h3 * p1, m2 = 1.f + h3 * q1;
    const float p2 = a[1] - d[1] * c2, q2 = -d[1] * m2;
    const float c3 = h * (p2 - p1 * (1.0f / 3.0f)), m3 = 1.f + h * (q2 - q1 * (1.0f / 3.0f));
    const float p3 = a[2] - d[2] * c3, q3 = -d[2] * m3;
    const float c4 = h * (p1 - p2 + p3), m4 = 1.f + h * (q1 - q2 + q3);
    float gp1 = G * gb, gq1 = G * gA;
    const float gp4 = G * gb, gq4 = G * gA;
    float gp2 = 3.f * G * gb, gp3 = 3.f * G * gb, gq2 = 3.f * G * gA, gq3 = 3.f * G * gA;
    ga[3] = gp4;
    gd[3] = -c4 * gp4 - m4 * gq4;
    const float gc4 = -d[3] * gp4, gm4 = -d[3] * gq4;
    gp1 += h * gc4; gp2 -= h * gc4; gp3 += h * gc4;
    gq1 += h * gm4; gq2 -= h * gm4; gq3 += h * gm4;
    ga[2] = gp3;
    gd[2] = -c3 * gp3 - m3 * gq3;
    const float gc3 = -d[2] * gp3, gm3 = -d[2] * gq3;
    gp2 += h * gc3; gp1 -= h3 * gc3;
    gq2 += h * gm3; gq1 -= h3 * gm3;
    ga[1] = gp2;
    gd[1] = -c2 * gp2 - m2 * gq2;
    const float gc2 = -d[1] * gp2, gm2 = -d[1] * gq2;
    gp1 += h3 * gc2; gq1 += h3 * gm2;
    ga[0] = gp1;
    gd[0] = -gq1;
  }
}

// ---- grad_mode = reference_adjoint: torchdiffeq.odeint_adjoint's backward for the fixed-grid solvers (blackbox_ode.py:40-42, the
// reference default; oracle: _OdeintAdjoint).  From node n+1 back to node n (hb = t_n - t_{n+1} < 0) the augmented state
// (y, lam, g_theta) takes ONE step of the same method, y restarting from the stored forward value:
//   dlam/dt = +d(t) lam  =>  lam_n = M lam_{n+1} (+ dLoss/dx_n),      dg_theta/dt = -lam (a' dxa/dtheta - y d' dxd/dtheta).
// D[j] = d at the j-th BACKWARD stage: node n+1, then the forward stages of step n in reverse order (rk4: r = 2, 1, 0).
__device__ __forceinline__ float radj_M(int method, float hb, const float D[4]) {
  if (method == SLODE_EULER) return 1.f + hb * D[0];
  if (method == SLODE_MIDPOINT) return 1.f + hb * D[1] * (1.f + 0.5f * hb * D[0]);
  const float h3 = hb * (1.0f / 3.0f);
  const float m2 = 1.f + h3 * D[0];
  const float m3 = 1.f + hb * (D[1] * m2 - D[0] * (1.0f / 3.0f));
  const float m4 = 1.f + hb * (D[0] - D[1] * m2 + D[2] * m3);
  return 1.f + hb * 0.125f * (D[0] + 3.f * (D[1] * m2 + D[2] * m3) + D[3] * m4);
}
// Quadrature of the parameter adjoint over one backward step: per backward stage j the weights of dxa/dtheta and dxd/dtheta
// BEFORE the sigmoid derivatives: gaq[j] = -w_j lam_j, gdq[j] = +w_j lam_j y_j  (w = hb * RK weights).
__device__ __forceinline__ void radj_stage_grads(int method, float hb, const float A[4], const float D[4], float lam, float y,
                                                 float gaq[4], float gdq[4]) {
  gaq[0] = gaq[1] = gaq[2] = gaq[3] = 0.f;
  gdq[0] = gdq[1] = gdq[2] = gdq[3] = 0.f;
  if (method == SLODE_EULER) {
    gaq[0] = -hb * lam;
    gdq[0] = hb * lam * y;
  } else if (method == SLODE_MIDPOINT) {
    const float ym = y + 0.5f * hb * (A[0] - D[0] * y), lm = (1.f + 0.5f * hb * D[0]) * lam;
    gaq[1] = -hb * lm;
    gdq[1] = hb * lm * ym;
  } else {
    const float h3 = hb * (1.0f / 3.0f), w1 = hb * 0.125f, w3 = hb * 0.375f;
    const float k1 = A[0] - D[0] * y, m2 = 1.f + h3 * D[0];
    const float y2 = y + h3 * k1, k2 = A[1] - D[1] * y2;
    const float m3 = 1.f + hb * (D[1] * m2 - D[0] * (1.0f / 3.0f));
    const float y3 = y + hb * (k2 - k1 * (1.0f / 3.0f)), k3 = A[2] - D[2] * y3;
    const float m4 = 1.f + hb * (D[0] - D[1] * m2 + D[2] * m3);
    const float y4 = y + hb * (k1 - k2 + k3);
    gaq[0] = -w1 * lam;            gdq[0] = w1 * lam * y;
    gaq[1] = -w3 * m2 * lam;       gdq[1] = w3 * m2 * lam * y2;
    gaq[2] = -w3 * m3 * lam;       gdq[2] = w3 * m3 * lam * y3;
    gaq[3] = -w1 * m4 * lam;       gdq[3] = w1 * m4 * lam * y4;
  }
}

// Affine recurrence y_{k+1} = A[i(k)] * y_k + v[j(k)], k = 0..T-2, results stored back over v, executed by ONE wave.
//   forward (REV=false): i = k,       j = k+1, y_0 = v[0]      (x_{n+1} = A_n x_n + b_n, b_n pre-stored in v[n+1])
//   reverse (REV=true):  i = T-2-k,   j = i,   y_0 = v[T-1]    (lambda_i = A_i lambda_{i+1} + g_i)
// Affine maps compose, so the T-1 long dependency chain is cut into chunks handled by lanes (component s, chunk c): compose the chunk's
// maps (registers), scan the composed maps across a component's chunks, replay the chunk.  Lane = 8 s + c: the EIGHT chunks of a
// component are half a DPP row, so the Kogge-Stone steps are row_shr:1 / 2 / 4 moves on the VALU (until round 3: lane = c S + s, twelve
// chunks, and every step a pair of ds_bpermute at lane distance d S -- ten dependent LDS-pipe round trips per scan); what a move drags
// in from the neighbouring component's half-row lands on lanes c < d, which ignore it.
constexpr int SCAN_NCH = 8;   // chunks per component and wave
template <int D>
__device__ __forceinline__ float scan_shr(float v) {   // value of the lane D places down the row (0 past the row's start)
  return dpp_f<0x110 + D>(v);
}
// steps per lane and pass of the one-wave scan (2 x this many registers): a grid length known at compile time gets the fewest passes
// of <= 26 steps and no more unrolled iterations than it needs (T = 200: one pass of 25; T = 300: two of 19; T = 100: one of 13)
__host__ __device__ constexpr int wave_scan_cl(int T_) {
  if (T_ <= 0) return 17;
  const int n = T_ - 1, passes = (n + SCAN_NCH * 26 - 1) / (SCAN_NCH * 26);
  return (n + SCAN_NCH * passes - 1) / (SCAN_NCH * passes);
}
template <int S, bool REV, int CLMAX = 17>
__device__ __forceinline__ void wave_affine_scan(const float* __restrict__ s_A, float* __restrict__ s_v, int T, int lane) {
  static_assert(S <= 8, "one half-row of eight chunks per state component");
  constexpr int NC = SCAN_NCH;   // (CLMAX steps per lane per pass: one pass up to T = NC * CLMAX + 1; longer grids take several)
  const int nsteps = T - 1;
  const int c = lane & (NC - 1), s = min(lane >> 3, S - 1);
  const bool lane_on = (lane >> 3) < S;
  float carry = lane_on ? s_v[(REV ? (T - 1) : 0) * S + s] : 0.f;
  for (int base = 0; base < nsteps; base += NC * CLMAX) {
    const int left = nsteps - base;
    int CL = (left + NC - 1) / NC;
    if (CL > CLMAX) CL = CLMAX;
    const int k0 = base + c * CL;
    float Ar[CLMAX], vr[CLMAX];
#pragma unroll
    for (int q = 0; q < CLMAX; ++q) {
      const int kk = k0 + q;
      const bool on = lane_on && q < CL && kk < nsteps;
      const int kc = min(kk, nsteps - 1), i = REV ? (T - 2 - kc) : kc;  // unconditional LDS reads (clamped), then selects
      const float Aq = s_A[i * S + s], vq = s_v[(REV ? i : i + 1) * S + s];
      Ar[q] = on ? Aq : 1.f;
      vr[q] = on ? vq : 0.f;
    }
    float P = 1.f, Q = 0.f;   // this chunk's composed map y -> P y + Q
#pragma unroll
    for (int q = 0; q < CLMAX; ++q) {
      Q = fmaf(Ar[q], Q, vr[q]);
      P *= Ar[q];
    }
    // inclusive scan of the composed maps over the component's chunks (Kogge-Stone): afterwards (P, Q) maps the pass's start state to
    // the END of chunk c
    { const float Pp = scan_shr<1>(P), Qp = scan_shr<1>(Q); if (c >= 1) { Q = fmaf(P, Qp, Q); P *= Pp; } }
    { const float Pp = scan_shr<2>(P), Qp = scan_shr<2>(Q); if (c >= 2) { Q = fmaf(P, Qp, Q); P *= Pp; } }
    { const float Pp = scan_shr<4>(P), Qp = scan_shr<4>(Q); if (c >= 4) { Q = fmaf(P, Qp, Q); P *= Pp; } }
    const float Pe = scan_shr<1>(P), Qe = scan_shr<1>(Q);
    float y = (c == 0) ? carry : fmaf(Pe, carry, Qe);   // state at the start of this lane's chunk
#pragma unroll
    for (int q = 0; q < CLMAX; ++q) {
      const int kk = k0 + q;
      const bool on = lane_on && q < CL && kk < nsteps;
      y = fmaf(Ar[q], y, vr[q]);  // masked entries are the identity map
      if (on) {
        const int i = REV ? (T - 2 - kk) : kk;
        s_v[(REV ? i : i + 1) * S + s] = y;
      }
    }
    carry = __shfl(y, (lane & ~(NC - 1)) + NC - 1, 64);   // the component's last chunk
  }
}

// Barrier among the NW waves of ONE trajectory of a packed workgroup (PK > 1, SOFTB): gfx950's s_barrier always joins the whole workgroup,
// which would lock the packed trajectories' phases together (the PK = 4 lockstep arm: +3.7 us).  One LDS word per trajectory counts
// arrivals (ds_add_rtn by lane 0 behind the wave's own LDS / memory operations); the wave whose add completes the count wakes the others
// with s_wakeup, which meanwhile sleep and poll (one broadcast ds_read per poll) until the count reaches their own arrival number --
// `gen`, a per-wave register that every wave of the trajectory advances identically.  Workgroup-scope release / acquire fences give the compiler and the hardware the same
// ordering a __syncthreads() would.
#ifndef SLODE_SB_SLEEP
#define SLODE_SB_SLEEP 8
#endif
__device__ __forceinline__ void soft_barrier(unsigned int* cnt, unsigned int& gen, int nwaves) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  gen += (unsigned int)nwaves;
  unsigned int old = 0;
  if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  old = (unsigned int)__builtin_amdgcn_readfirstlane((int)old);
  if (old + 1u == gen) {
    asm volatile("s_wakeup");   // the last arriver pings every sleeping wave of the workgroup: the waiters below leave their s_sleep at once
  } else {
    // a waiter parks in s_sleep (64 x SLODE_SB_SLEEP cycles at most: a ping that arrives between its poll and its sleep is lost) and looks
    // at the counter again when woken -- by its own trajectory's last arriver or by another trajectory's (then it goes back to sleep)
    while ((int)(__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - (int)gen) < 0)
      __builtin_amdgcn_s_sleep(SLODE_SB_SLEEP);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The same recurrence on ALL waves of the workgroup (forward scan: nothing else runs beside it).  NW * 8 chunks of <= 8 steps per
// component (the block has at least roundup64(T) threads), lanes (s, chunk): compose the chunk's maps, Kogge-Stone over the wave's
// chunks of the component (DPP, as above), the waves' total maps through LDS (s_xw[NW][2][S]) and one barrier, then every lane applies
// the totals of the waves before its own and replays its chunk.  Contains a barrier: every thread of the workgroup calls it.  The
// caller's next barrier publishes the results.
template <int S, bool REV, int CLMAX = 8, class Bar>
__device__ __forceinline__ void block_affine_scan(const float* __restrict__ s_A, float* __restrict__ s_v, int T, int tid, int NT,
                                                  float* __restrict__ s_xw, Bar&& bar) {
  static_assert(S <= 8, "one half-row of eight chunks per state component");
  constexpr int NC = SCAN_NCH;   // (CLMAX: registers per lane; the shape-specialised kernels pass their exact chunk length)
  const int NW = NT >> 6, wave = tid >> 6, lane = tid & 63;
  const int nsteps = T - 1, nch = NW * NC;
  const int CL = (nsteps + nch - 1) / nch;   // <= 8 <= CLMAX (NT >= roundup64(T): nch = NT / 8 >= nsteps / 8)
  const int c = lane & (NC - 1), s = min(lane >> 3, S - 1);
  const bool lane_on = (lane >> 3) < S;
  const int k0 = (wave * NC + c) * CL;
  const float y0 = s_v[(REV ? (T - 1) : 0) * S + s];
  float Ar[CLMAX], vr[CLMAX];
#pragma unroll
  for (int q = 0; q < CLMAX; ++q) {
    const int kk = k0 + q;
    const bool on = lane_on && q < CL && kk < nsteps;
    const int kc = min(kk, nsteps - 1), i = REV ? (T - 2 - kc) : kc;  // unconditional LDS reads (clamped), then selects
    const float Aq = s_A[i * S + s], vq = s_v[(REV ? i : i + 1) * S + s];
    Ar[q] = on ? Aq : 1.f;
    vr[q] = on ? vq : 0.f;
  }
  float P = 1.f, Q = 0.f;   // this chunk's composed map y -> P y + Q
#pragma unroll
  for (int q = 0; q < CLMAX; ++q) {
    Q = fmaf(Ar[q], Q, vr[q]);
    P *= Ar[q];
  }
  // inclusive scan over the wave's chunks of the component
  { const float Pp = scan_shr<1>(P), Qp = scan_shr<1>(Q); if (c >= 1) { Q = fmaf(P, Qp, Q); P *= Pp; } }
  { const float Pp = scan_shr<2>(P), Qp = scan_shr<2>(Q); if (c >= 2) { Q = fmaf(P, Qp, Q); P *= Pp; } }
  { const float Pp = scan_shr<4>(P), Qp = scan_shr<4>(Q); if (c >= 4) { Q = fmaf(P, Qp, Q); P *= Pp; } }
  if (lane_on && c == NC - 1) {   // the wave's total map
    s_xw[(wave * 2 + 0) * S + s] = P;
    s_xw[(wave * 2 + 1) * S + s] = Q;
  }
  const float Pe = scan_shr<1>(P), Qe = scan_shr<1>(Q);
  bar();
  float y = y0;   // state at the start of this wave's stretch
  for (int w = 0; w < wave; ++w) y = fmaf(s_xw[(w * 2 + 0) * S + s], y, s_xw[(w * 2 + 1) * S + s]);
  y = (c == 0) ? y : fmaf(Pe, y, Qe);   // state at the start of this lane's chunk
#pragma unroll
  for (int q = 0; q < CLMAX; ++q) {
    const int kk = k0 + q;
    const bool on = lane_on && q < CL && kk < nsteps;
    y = fmaf(Ar[q], y, vr[q]);  // masked entries are the identity map
    if (on) {
      const int i = REV ? (T - 2 - kk) : kk;
      s_v[(REV ? i : i + 1) * S + s] = y;
    }
  }
}

__host__ __device__ constexpr int ode_threads_for(int T, int Q, int C, int S) {
  // waves >= 1 carry the head-gradient role (one (q,c,s) entry per thread) next to the adjoint scan on wave 0
  const int nt = ((T + 63) / 64) * 64, need = 64 + ((Q * C * S + 63) / 64) * 64;
  return nt < need ? need : nt;
}

// Register budget = 512 / (waves per SIMD the declared block size forces).  Loop-free forms fit 128 (S = 5) / 168-256 (S = 8, short
// grids) without spilling.  The persistent-loop forms hoist kernel-argument loads and need more: the specialised ones declare their
// exact block size, the generic ones cap the grid length they accept (512 threads) -- NO instantiation may spill a VGPR
// (tools/check_spills.py).
__host__ __device__ constexpr int ode_max_threads(int S, int T_, int C_, int Q_, bool one, bool bwd) {
  if (!one && bwd) return T_ > 0 ? ode_threads_for(T_, Q_, C_, S) : 512;
  return (S > 5 || (T_ > 0 && T_ <= 128)) ? 768 : 1024;
}

__host__ __device__ constexpr int ode_block_bound(int S, int T_, int C_, int Q_, bool one, bool bwd, int pk) {
  return pk > 1 ? pk * ode_threads_for(T_ ? T_ : 1, Q_ ? Q_ : 1, C_ ? C_ : 1, S) : ode_max_threads(S, T_, C_, Q_, one, bwd);
}

// Kernel algorithm variants (ALG).  The dynamics net never sees the state and its hidden layer is relu(w_t t + u_j(z)): every unit is
// switched on over a prefix or a suffix of the (monotone) stage-time table, so
//   ALG 0 (product): forward = piecewise-linear table of the 2S head pre-activations over <= H+1 time segments (one fma per head and
//          stage time instead of a 2S x H product); backward contraction = chunked prefix / suffix sums of the per-sample gradients
//          evaluated at each unit's switching index (no per-sample x per-unit work at all);
//   ALG 1: forward = direct evaluation (SGPR-operand v_fma, round-1 code), backward as ALG 0;
//   ALG 2: forward direct, backward contraction as two f32 MFMA GEMMs [g | g t]^T x mask (v_mfma_f32_16x16x4_f32): the measured
//          A/B arm SURVEY hard part 8 / DESIGN 5 ask for.
// ALG 1 / 2 are instantiated for the metric shape only (handle flag `ode_alg`, tests + bench A/B), never dispatched otherwise.
//   ALG 3: the solver-free scorer of an externally solved trajectory (dopri5 training) with compile-time shape: BASELINE config[2].
//
// S = 8 carries 60% more live state per thread: its instantiations trade one wave/SIMD for a 168-VGPR budget (T <= 768).
// T_, C_, L_, Q_, M_ (time points, channels, latent dim, decoder heads, solver): 0 / -1 = read from the launch struct; the
// shape-specialised instantiations get compile-time LDS offsets, loop bounds and solver.
// ONE: the grid has one workgroup per trajectory: no persistent loop.  In BOTH forms every gradient element goes to its slab entry as soon as it is final:
// nothing but the scalar loss partial is carried in registers from one trajectory to the next (the round-1 looped form kept
// 2S+2 accumulators per thread live across the loop, spilled, and hipcc placed one spill store ahead of the exec restore of a
// control-flow join -- wrong gradients; tools/check_spills.py now rejects any kernel that uses scratch).
// PK > 1 (shape-specialised loop-free forms only): PK trajectories share one workgroup of PK * NT threads -- each trajectory keeps its own
// NT threads (whole waves), its own LDS region and its own slab row, exactly as if it were a workgroup of its own ("virtual workgroup"
// vblk = blockIdx * PK + trajectory); only the barriers are shared.  The serial wave of trajectory i is the workgroup's wave
// i * NW + i: the four serial waves sit on four different SIMDs.
// ENCF (shape-specialised loop-free forms with a short latent; folded encoder path): the workgroup runs the ENCODER FORWARD of its own
// trajectory in the set-up -- pre = b_eff + W_eff x (each wave 13 rows of W_eff, its lanes the columns: two batches of register-resident
// rows, one wave_sum16), tanh, the two head layers -- while the set-up's LDS-DMA is in flight; loc / scale go straight to the LDS slots
// P0a reads, the saved tanh to LDS (P7's head backward) and to global memory (the head-layer GEMMs).  The enc_fwd2 launch disappears.
// SOFTB (PK > 1 with ENCF): the packed trajectories share ONE pass over W_eff in the set-up (wave w of the workgroup takes a few rows for
// all PK trajectories: 120 KB per workgroup through the CU's L1 instead of 120 KB per trajectory) and then run on barriers of their own
// (soft_barrier) -- not in lockstep.
template <int S, int H, bool BWD, int T_ = 0, int C_ = 0, int L_ = 0, int Q_ = 0, int M_ = -1, bool RA = false, bool ONE = false, int ALG = 0, int PK = 1,
          bool ENCF = false, bool SOFTB = false>
// (waves per SIMD asked of the register allocator: the solver-free scorer (ALG 3) is a throughput kernel that waits 56 % of its wave cycles --
// with the hint it compiles to 79 instead of 101 VGPRs, six waves per SIMD = eight workgroups per CU instead of five: 76 -> 71 us at config[2])
__global__ void __launch_bounds__(ode_block_bound(S, T_, C_, Q_, ONE, BWD, PK)) __attribute__((amdgpu_waves_per_eu(ALG == 3 ? 5 : (PK == 2 ? 4 : 1))))
ode_elbo_kernel(const float* __restrict__ pl_stage_t, const float* __restrict__ pl_pseg, const float* __restrict__ pl_loc,
                const float* __restrict__ pl_scale, const float* __restrict__ pl_eps, const float* __restrict__ pl_u,
                const float* __restrict__ pl_sigtab, const OdeK k) {
  // The seven leading pointers repeat fields of `k` (stage_t, pseg, loc | z_in, scale, eps, u, sigtab | cstd): as plain pointer arguments
  // they are PRELOADED into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count, Makefile), so the first global loads of the
  // set-up go out at once instead of behind an s_load of the kernel-argument segment -- one of the two serialised cold misses every
  // kernel of the step starts with.
  static_assert(PK == 1 || (ONE && T_ > 0), "packed trajectories: shape-specialised loop-free forms only");
  static_assert(!ENCF || (ONE && T_ > 0 && ALG == 0 && (PK == 1 || SOFTB)), "fused encoder forward: shape-specialised loop-free product form only");
  static_assert(!SOFTB || (PK > 1 && ENCF), "per-trajectory barriers: the packed form with the shared encoder pass");
  extern __shared__ __attribute__((aligned(16))) float smem_wg[];
  constexpr int NWT = PK > 1 ? ode_threads_for(T_ ? T_ : 1, Q_ ? Q_ : 1, C_ ? C_ : 1, S) / 64 : 1;   // waves per trajectory (PK > 1)
  const int wave_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int traj = PK > 1 ? wave_wg / NWT : 0;
  const int tid = PK > 1 ? (((wave_wg % NWT) + NWT - (traj % NWT)) % NWT) * 64 + (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int vblk = PK > 1 ? (int)blockIdx.x * PK + traj : (int)blockIdx.x, vgrid = (int)gridDim.x * PK;
  const int T = T_ ? T_ : k.T, C = C_ ? C_ : k.C, L = L_ ? L_ : k.L, Q = Q_ ? Q_ : k.Q;
  const int method = M_ >= 0 ? M_ : k.method;
  const int R = M_ >= 0 ? (M_ == SLODE_EULER ? 1 : (M_ == SLODE_MIDPOINT ? 2 : 3)) : k.R;
  const int n_stage_t = (M_ >= 0 && T_) ? R * (T - 1) + 1 : k.nt;
  const int uses_next = M_ >= 0 ? (M_ == SLODE_RK4 ? 1 : 0) : k.uses_next, gauss = Q_ ? (Q_ == 1 ? 1 : 0) : k.gauss;
  // RA (grad_mode = reference_adjoint): the backward pass also needs a, d at node n+1 for euler / midpoint
  const int need_next = RA ? 1 : uses_next;
  const int NT = T_ ? ode_threads_for(T_, Q_, C_, S) : (int)blockDim.x;
  constexpr bool COLDG = ode_cold_global(L_);
  const LdsMap m = lds_map(T, S, H, C, L, Q, n_stage_t, COLDG ? ode_hot_floats(S, H) : k.npar, NT, k.n_aux_lds, ONE, ALG == 3);
  float* const smem = smem_wg + (PK > 1 ? traj * m.total : 0);
  float* s_ts = smem + m.ts;
  float* s_sig = smem + m.sig;
  float* s_A = smem + m.A;
  float* s_x = smem + m.x;
  float* s_lam = smem + m.lam;
  float* s_st = smem + m.st;
  float* s_tab = s_A;              // P1: piecewise-linear table [H+1][V (2S) | slope (2S)]
  float* s_G = s_A;                // P5 -> P6: per-sample gradient rows [nt][2S] = [dLoss/d(pre_a) (S) | dLoss/d(pre_d) (S)]
  float* s_ct = smem + m.ct;
  float* s_gm = smem + m.gm;
  float* s_hp = smem + m.hp;
  float* s_tau = smem + m.tau;
  int* s_ps = reinterpret_cast<int*>(smem + m.ps);
  int* s_ord = reinterpret_cast<int*>(smem + m.ord);
  float* s_sgs = smem + m.sgs;     // events in table order: sign, unit's w_t, unit's pre-activation at the event time (= s_tau[k+1])
  float* s_ewt = smem + m.ewt;
  float* s_epre = smem + m.epre;
  float* s_epre0 = smem + m.epre0;   // ... and at the first stage time
  float* s_esw = smem + m.esw;       // sign * (column `unit` of [W_g; W_d]), [event][2S]
  int* s_ms = reinterpret_cast<int*>(smem + m.ms);
  int* s_sf = reinterpret_cast<int*>(smem + m.sf);
  float* s_par = smem + m.par;  // small weights, staged once per workgroup (cold phases read LDS, not HBM/L2)
  // offsets of the hot tensors inside s_par (the whole segment is there unless COLDG: then only the two hot ranges, back to back), and
  // where the cold ones are read from
  const int hb1 = COLDG ? 0 : k.o_b1, hw2 = hb1 + H, hb2 = hw2 + S * H;
  const int hbh = COLDG ? ode_hot_r1(S, H) : k.o_bh, hwg = hbh + H, hbg = hwg + S * H, hwd = hbg + S, hbd = hwd + S * H;
  const float* const cpar = COLDG ? pl_pseg : s_par;
  // COLDB: during P0 (latent sample, priors, P0b's two L-long dot products, label heads) the cold parameters of a long-latent fixed-grid
  // kernel sit in the -- then idle -- work block A | x | lam | st, put there by the set-up's LDS-DMA: three ranges back to back,
  // [priors | W_1] (from the start of the segment), W_z (shifted by the hot range it follows), the label heads (shifted past both hot
  // ranges and the decoder heads).  P0c / P1 overwrite the block; P7's second look at W_z / W_1 / the label heads reads global memory.
  constexpr bool COLDB = COLDG && ALG != 3;
  const float* const cp0 = COLDB ? s_A : cpar;
  const int sh_wh = COLDB ? ode_hot_r1(S, H) : 0;
  const int sh_aux = COLDB ? k.o_aux_w1[0] - (k.o_b1 + H * (1 + L)) : 0;
  float* s_uu = smem + m.uu;  // this trajectory's label row u[b, :]
  float* s_auxh = smem + m.auxh;    // label heads: hidden activations [head][32]
  float* s_auxd = smem + m.auxd;    //              softplus' then dLoss/d(hidden pre-activation)
  float* s_auxgo = smem + m.auxgo;  //              dLoss/d(output logits) [head][12] (slot 8: d/d constant_std_*)
  float* s_z = smem + m.z;
  float* s_gzl = smem + m.gzl;
  float* s_gpl = smem + m.gpl;
  float* s_gls = smem + m.gls;
  float* s_u = smem + m.u;
  float* s_wt = smem + m.wt;
  float* s_pre0 = smem + m.pre0;
  float* s_hid0 = smem + m.hid0;
  float* s_x0 = smem + m.x0;
  float* s_go = smem + m.go;
  float* s_gp0 = smem + m.gp0;
  float* s_gu = smem + m.gu;
  float* s_red = smem + m.red;
  int* s_meta = reinterpret_cast<int*>(smem + m.meta);
  float* s_pf = smem + m.pf;      // [3][pad4(L)]: loc | scale | eps of the trajectory about to start (or z_in | - | -)
  float* s_encw = s_st;   // P7: encoder head weights [2][L][Hc] staged over the (then idle) stage buffer
  float* s_ehid = PK > 1 ? smem_wg + PK * m.total + traj * 64 : smem + m.total;   // ENCF: this trajectory's tanh(pre) [64] (the launch adds 256 bytes per trajectory)
  // SOFTB: this trajectory's arrival counter (behind the PK tanh rows) and the wave's own arrival number
  unsigned int* const s_sbar = reinterpret_cast<unsigned int*>(smem_wg + PK * m.total + PK * 64) + traj;
  unsigned int sb_gen = 0;
  auto BAR = [&]() {
    if constexpr (SOFTB) soft_barrier(s_sbar, sb_gen, NWT);
    else __syncthreads();
  };

  const cptr wg = (cptr)k.wg, bg = (cptr)k.bg, wd = (cptr)k.wd, bd = (cptr)k.bd;
  // wave 0 carries every serial stretch of a trajectory (latent sample, switching indices, table, both scans): it issues ahead of the
  // bulk waves of the co-resident workgroups
  const int prio_slot = PK > 1 ? traj : __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 8));   // residency slot on its CU (dispatch order: 256 CUs per pass)
  (void)prio_slot;
  const int tid_outer = tid;
  STAMP(0);
#ifndef SLODE_STAMPS
  if (ENCF) {
    // Set-up of the fused form: issue priority = residency slot, the YOUNGEST workgroup of the CU first.  The set-up is a queue of W_eff
    // loads through one L1 and the CU arbitrates oldest-first, so the last slot's requests used to trail (set-up 5.6 / 10.6 us for a
    // first- / last-slot workgroup) and the kernel ends when that workgroup does.  Measured: 54.4 -> 53.8 us per step; keeping the slot
    // priority through P0 as well (wave 0's serial stretch) instead of the rotation: 55.5.  The rotation resumes at the next boundary.
    if (prio_slot >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio_slot == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio_slot == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
#endif
  // scoring an externally solved trajectory (dopri5 training, generic instantiation): no solve, nothing flows through a solver here
  const bool ext = (ALG == 3) || ((T_ == 0) && k.x_ext != nullptr);   // ALG 3: shape-specialised scorer (every solver phase is dead code)

  // ---- per-workgroup setup (shared by all trajectories this workgroup integrates) ----------------------
  // The stage-time table and the parameter segment go global -> LDS by LDS-DMA (global_load_lds: 64 consecutive floats per wave
  // instruction, no registers, no address arithmetic per element); the few per-thread values (this trajectory's latent inputs, its
  // constant_std column) go through registers meanwhile.
  // (COLDB) the cold parameter ranges -> the work block, by LDS-DMA; issued in the set-up and, in the persistent-loop form, again at the
  // top of every later trajectory (the block is overwritten from P0c on)
  auto stage_cold = [&]() {
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6, lane = tid & 63;
    const int n1 = k.o_b1, n2 = H * (1 + L), n3 = k.n_aux > 0 ? k.npar - k.o_aux_w1[0] : 0;   // (no label heads in this launch: nothing to stage)
    const float* g1 = pl_pseg;
    const float* g2 = pl_pseg + k.o_wh;
    const float* g3 = pl_pseg + k.o_aux_w1[0];
    for (int base = wv * 64; base < n1; base += NW * 64)
      if (base + lane < n1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g1 + base + lane),
                                         (__attribute__((address_space(3))) void*)(s_A + base), 4, 0, 0);
    for (int base = wv * 64; base < n2; base += NW * 64)
      if (base + lane < n2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g2 + base + lane),
                                         (__attribute__((address_space(3))) void*)(s_A + n1 + base), 4, 0, 0);
    for (int base = wv * 64; base < n3; base += NW * 64)
      if (base + lane < n3)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g3 + base + lane),
                                         (__attribute__((address_space(3))) void*)(s_A + n1 + n2 + base), 4, 0, 0);
  };
  float sigr[SLODE_MAX_C] = {1.f, 1.f, 1.f, 1.f};   // ONE: softplus(constant_std[c, t = tid]) stays in registers until P3
  float e_hw[4] = {0.f, 0.f, 0.f, 0.f}, e_hb = 0.f;   // ENCF: this lane's head weights and bias, from the set-up's loads to the head layers
  // packed form with the shared encoder pass: this wave's first batch of W_eff rows is the kernel's FIRST request (the longest wait of the
  // set-up: everything else is issued behind it)
  constexpr int ECT0 = (C_ ? C_ : 1) * (T_ ? T_ : 1), ENU0 = (ECT0 + 127) / 128, EBP0 = 4;
  f32x2 w_first[(ENCF && PK > 1) ? EBP0 : 1][(ENCF && PK > 1) ? ENU0 : 1];
  if (ENCF && PK > 1) {
    constexpr int EWG0 = PK * 4, ERWP0 = (52 + EWG0 - 1) / EWG0, NRB0 = (52 + ERWP0 - 1) / ERWP0;
    const int rblk = wave_wg < NRB0 ? (wave_wg + (int)blockIdx.x) % NRB0 : wave_wg, lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int r = 0; r < EBP0; ++r)
#pragma unroll
      for (int u = 0; u < ENU0; ++u) {
        const int row = min(rblk * ERWP0 + r, k.Hc - 1);
        w_first[r][u] = (r < ERWP0) ? *reinterpret_cast<const f32x2*>(k.enc_weff + (long long)row * ECT0 + min(2 * lane + 128 * u, ECT0 - 2)) : f32x2{0.f, 0.f};
      }
    asm volatile("" ::: "memory");
  }
  {
    const int n_ts = n_stage_t, n_par = k.npar, n_sig = (!ONE && k.with_ll) ? C * T : 0;
    {
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6, lane = tid & 63;
      for (int base = wv * 64; base < n_ts; base += NW * 64)
        if (base + lane < n_ts)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl_stage_t + base + lane),
                                           (__attribute__((address_space(3))) void*)(s_ts + base), 4, 0, 0);
      if (!COLDG) {
        for (int base = wv * 64; base < n_par; base += NW * 64)
          if (base + lane < n_par)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl_pseg + base + lane),
                                             (__attribute__((address_space(3))) void*)(s_par + base), 4, 0, 0);
      } else {   // the two hot ranges
        const float* r1 = pl_pseg + k.o_b1;
        const float* r2 = pl_pseg + k.o_bh;
        for (int base = wv * 64; base < ode_hot_r1(S, H); base += NW * 64)
          if (base + lane < ode_hot_r1(S, H))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r1 + base + lane),
                                             (__attribute__((address_space(3))) void*)(s_par + base), 4, 0, 0);
        for (int base = wv * 64; base < ode_hot_r2(S, H); base += NW * 64)
          if (base + lane < ode_hot_r2(S, H))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r2 + base + lane),
                                             (__attribute__((address_space(3))) void*)(s_par + ode_hot_r1(S, H) + base), 4, 0, 0);
      }
    }
    if (COLDB) stage_cold();
    float v_wt = 0.f;   // COLDG: time column of dynamics_hidden, straight from global with the rest of the set-up loads
    if (COLDG && tid < H) v_wt = pl_pseg[k.o_wh + tid * (1 + L)];
    float v_l0 = 0.f, v_l1 = 1.f, v_l2 = 0.f, v_u = 0.f;
    float v_c[SLODE_MAX_C] = {0.f, 0.f, 0.f, 0.f};
    const int b_first = vblk;
    if (b_first < k.B) {
      const int lc = min(tid, L - 1);
      if (!ENCF) v_l0 = pl_loc[(long long)b_first * L + lc];     // (pure solve: z_in)
      if (pl_scale != nullptr) {
        if (!ENCF) v_l1 = pl_scale[(long long)b_first * L + lc];
        v_l2 = k.rng.on ? slode_rng_normal(k.rng, b_first, lc) : pl_eps[(long long)b_first * L + lc];
      }
      if (pl_u != nullptr) v_u = k.lab.n ? slode_label_at(k.lab, pl_u, k.nu, b_first, min(tid, k.nu - 1)) : pl_u[(long long)b_first * k.nu + min(tid, k.nu - 1)];
    }
    // ENCF: the encoder forward of this trajectory (see the template comment).  Everything it reads from global memory is requested
    // here, beside the set-up loads; its results are finished behind the set-up barrier.
    constexpr int ERW = 13;                       // rows of W_eff per wave (4 waves: Hc <= 52, checked by the launcher)
    constexpr int ECT = (C_ ? C_ : 1) * (T_ ? T_ : 1), ENU = (ECT + 127) / 128;
    float e_acc[16], e_be = 0.f;
    constexpr int EWG = PK * 4, ERWP = (52 + EWG - 1) / EWG;   // packed form: waves of the workgroup, rows of W_eff per wave (all PK trajectories)
    STAMP(23);   // (set-up: LDS-DMA and per-thread loads requested)
    if (ENCF && PK > 1) {
      static_assert(!(ENCF && PK > 1) || (ERWP * PK <= 16 && ode_threads_for(T_ ? T_ : 1, Q_ ? Q_ : 1, C_ ? C_ : 1, S) == 256), "one wave_sum16 per wave");
      static_assert(!(ENCF && PK > 1) || ((ECT & 1) == 0 && 2 * (L_ ? L_ : 1) * 16 <= 256), "even C*T, 16 lanes per head output");
      if (SOFTB && tid == 0) *s_sbar = 0u;
      // (the row block a wave takes rotates with the workgroup index: at any instant the chip's workgroups ask for different lines)
      constexpr int NRB = (52 + ERWP - 1) / ERWP;   // row blocks that hold rows (13 of 4 rows, 8 of 7)
      const int rblk = wave_wg < NRB ? (wave_wg + (int)blockIdx.x) % NRB : wave_wg;
      const int lane = (int)(threadIdx.x & 63), Hc = k.Hc, row0 = rblk * ERWP;
      f32x2 xv[PK][ENU];
#pragma unroll
      for (int t = 0; t < PK; ++t) {
        const float* xrow = k.obs + (long long)min((int)blockIdx.x * PK + t, k.B - 1) * ECT;
#pragma unroll
        for (int u = 0; u < ENU; ++u) xv[t][u] = *reinterpret_cast<const f32x2*>(xrow + min(2 * lane + 128 * u, ECT - 2));
      }
      {   // this trajectory's head weights and bias (as in the unpacked form); b_eff of the row whose sum this lane will finish
        const int o = tid >> 4, l16 = tid & 15, which = o / L, l = o - which * L;
        const float* W = (which ? k.enc_zls_w : k.enc_zloc_w) + l * Hc;
#pragma unroll
        for (int q = 0; q < 4; ++q) e_hw[q] = W[min(l16 + 16 * q, Hc - 1)];
        e_hb = which ? k.enc_zls_b[l] : k.enc_zloc_b[l];
        e_be = k.enc_beff[min(row0 + ((lane >> 2) & 15) / PK, Hc - 1)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) e_acc[r] = 0.f;
      constexpr int EBP = 4;   // rows per batch: 4 x ENU float2 = 40 registers in flight beside the PK x ENU float2 of the observations
#pragma unroll
      for (int r0 = 0; r0 < ERWP; r0 += EBP) {
        f32x2 w[EBP][ENU];
#pragma unroll
        for (int r = 0; r < EBP; ++r)
#pragma unroll
          for (int u = 0; u < ENU; ++u) {
            const int row = min(row0 + r0 + r, Hc - 1);   // (rows past Hc: a valid address, never used)
            if (r0 == 0) w[r][u] = w_first[r][u];   // (requested at the top of the kernel)
            else w[r][u] = (r0 + r < ERWP) ? *reinterpret_cast<const f32x2*>(k.enc_weff + (long long)row * ECT + min(2 * lane + 128 * u, ECT - 2)) : f32x2{0.f, 0.f};
          }
#pragma unroll
        for (int r = 0; r < EBP; ++r)
#pragma unroll
          for (int u = 0; u < ENU; ++u) {
            if (r0 + r < ERWP) {
              const bool in = 2 * lane + 128 * u < ECT;
              const float wx = in ? w[r][u].x : 0.f, wy = in ? w[r][u].y : 0.f;
#pragma unroll
              for (int t = 0; t < PK; ++t) e_acc[(r0 + r) * PK + t] = fmaf(wy, xv[t][u].y, fmaf(wx, xv[t][u].x, e_acc[(r0 + r) * PK + t]));
            }
          }
#pragma unroll
        for (int r = 0; r < EBP; ++r)
          if (r0 + r < ERWP) {
#pragma unroll
            for (int t = 0; t < PK; ++t) asm volatile("" : "+v"(e_acc[(r0 + r) * PK + t]) : : "memory");
          }
      }
    }
    STAMP(24);   // (packed form: this wave's W_eff x observation products done, i.e. its loads have returned)
    if (ENCF && PK == 1) {
      static_assert(!ENCF || ode_threads_for(T_ ? T_ : 1, Q_ ? Q_ : 1, C_ ? C_ : 1, S) == 256, "fused encoder forward: four waves");
      static_assert(!ENCF || ((ECT & 3) == 0 && 2 * (L_ ? L_ : 1) * 16 <= 256), "fused encoder forward: C*T a multiple of 4, 16 lanes per head output");
      // 16 bytes per lane and request (global_load_dwordx4: 1 KiB per wave instruction, the width the memory pipeline is built for; the
      // round-3 form asked for 8 bytes per lane, five requests per row instead of three)
      constexpr int ENU4 = (ECT + 255) / 256;
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, Hc = k.Hc;
      const float* xrow = k.obs + (long long)min(b_first, k.B - 1) * ECT;   // dense row, memory order = the column order of W_eff
      f32x4 xv[ENU4];
#pragma unroll
      for (int u = 0; u < ENU4; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xrow + min(4 * lane + 256 * u, ECT - 4));
      {   // head weights (16 lanes per output: lane l16 takes hidden units l16 + 16 q), b_eff of the row this lane will finish, head bias
        const int o = tid >> 4, l16 = tid & 15, which = o / L, l = o - which * L;
        const float* W = (which ? k.enc_zls_w : k.enc_zloc_w) + l * Hc;
#pragma unroll
        for (int q = 0; q < 4; ++q) e_hw[q] = W[min(l16 + 16 * q, Hc - 1)];
        e_hb = which ? k.enc_zls_b[l] : k.enc_zloc_b[l];
        e_be = k.enc_beff[min(wv * ERW + ((lane >> 2) & 15), Hc - 1)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) e_acc[r] = 0.f;
      const float* wbase = k.enc_weff + (long long)(wv * ERW) * ECT;
      constexpr int EB0 = 5;   // rows per batch: 5 x ENU4 float4 = 60 registers in flight, one batch at a time (the kernel's budget is 128)
#pragma unroll
      for (int r0 = 0; r0 < ERW; r0 += EB0) {
        f32x4 w[EB0][ENU4];
#pragma unroll
        for (int r = 0; r < EB0; ++r)
#pragma unroll
          for (int u = 0; u < ENU4; ++u) {
            const int row = min(wv * ERW + r0 + r, Hc - 1) - wv * ERW;   // (rows past Hc: a valid address, never used)
            w[r][u] = (r0 + r < ERW) ? *reinterpret_cast<const f32x4*>(wbase + (long long)row * ECT + min(4 * lane + 256 * u, ECT - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int r = 0; r < EB0; ++r)
#pragma unroll
          for (int u = 0; u < ENU4; ++u) {
            if (r0 + r < ERW) {
              const bool in = 4 * lane + 256 * u < ECT;
              const f32x4 wv4 = in ? w[r][u] : f32x4{0.f, 0.f, 0.f, 0.f};
              e_acc[r0 + r] = fmaf(wv4.w, xv[u].w, fmaf(wv4.z, xv[u].z, fmaf(wv4.y, xv[u].y, fmaf(wv4.x, xv[u].x, e_acc[r0 + r]))));
            }
          }
        // the next batch's loads must not be issued over this batch's registers: its sums are finished ahead of a compiler barrier
        // for memory operations (a scheduling barrier alone is not enough: the loads are clustered before the scheduler runs)
#pragma unroll
        for (int r = 0; r < EB0; ++r)
          if (r0 + r < ERW) asm volatile("" : "+v"(e_acc[r0 + r]) : : "memory");
      }
    }
    if (ONE && k.with_ll) {   // (with the per-step table: the scale itself instead of its parameter)
      const float* src = pl_sigtab;   // the table, or constant_std itself when there is none
#pragma unroll
      for (int c = 0; c < SLODE_MAX_C; ++c) v_c[c] = src[min(c, C - 1) * T + min(tid, T - 1)];
    }
    if (tid < L) {
      // prior-net lookup table for latent dim l (mechanistic_cvs.py:225-237): resolved once per workgroup, while the loads fly
      //   [0] in a conditional group  [1] loc bias  [2] log-scale bias  [3] loc weight row  [4] log-scale weight row  [5] u_off  [6] u_dim
      const int l = tid;
      int me[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int g = 0; g < k.ng; ++g) {
        const slode_group gr = k.grp[g];
        if (l >= gr.z_off && l < gr.z_off + gr.z_dim) {
          const int ll = l - gr.z_off;
          me[0] = 1; me[1] = k.o_ploc_b[g] + ll; me[2] = k.o_pls_b[g] + ll;
          me[3] = k.o_ploc_w[g] + ll * gr.u_dim; me[4] = k.o_pls_w[g] + ll * gr.u_dim; me[5] = gr.u_off; me[6] = gr.u_dim;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) s_meta[l * 8 + q] = me[q];
    }
    if (!ONE) {   // persistent-loop form: softplus(constant_std) once per workgroup, kept in LDS
      for (int i = tid; i < n_sig; i += NT) s_sig[i] = k.sigtab ? k.sigtab[i] : softplusf(k.cstd[i]);
    }
    if (tid < L) {
      if (!ENCF) { s_pf[tid] = v_l0; s_pf[pad4(L) + tid] = v_l1; }
      s_pf[2 * pad4(L) + tid] = v_l2;
    }
    if (ENCF && PK > 1) {   // sum (row r, trajectory t) = entry r * PK + t of the butterfly; tanh; into trajectory t's LDS row and to memory
      constexpr int NRB = (52 + ERWP - 1) / ERWP;
      const int rblk = wave_wg < NRB ? (wave_wg + (int)blockIdx.x) % NRB : wave_wg;
      const int lane = (int)(threadIdx.x & 63), idx = (lane >> 2) & 15, r = idx / PK, t = idx - r * PK, mrow = rblk * ERWP + r;
      const float v = wave_sum16(e_acc, lane);
      const float hv = tanhf(v + e_be);
      if ((lane & 3) == 0 && r < ERWP && mrow < k.Hc) {
        smem_wg[PK * m.total + t * 64 + mrow] = hv;
        k.enc_hid_out[(long long)((int)blockIdx.x * PK + t) * k.Hc + mrow] = hv;
      }
    }
    if (ENCF && PK == 1) {   // rows of this wave: one halving butterfly leaves the sum of row (lane >> 2) & 15 in every lane; tanh; saved
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, idx = (lane >> 2) & 15, mrow = wv * ERW + idx;
      const float v = wave_sum16(e_acc, lane);
      const float hv = tanhf(v + e_be);
      if ((lane & 3) == 0 && idx < ERW && mrow < k.Hc) {
        s_ehid[mrow] = hv;
        if (b_first < k.B) k.enc_hid_out[(long long)b_first * k.Hc + mrow] = hv;
      }
    }
    STAMP(25);   // (encoder products reduced, tanh stored)
    if (tid < k.nu) s_uu[tid] = v_u;
    if (COLDG && tid < 32) s_wt[tid] = v_wt;
    if (ONE && k.with_ll) {
#pragma unroll
      for (int c = 0; c < SLODE_MAX_C; ++c)
        if (c < C) sigr[c] = k.sigtab ? v_c[c] : softplusf(v_c[c]);
    }
    if (!ext && tid < ode_pad_rows(n_stage_t, NT, S)) s_ts[n_stage_t + tid] = 0.f;   // pad stage times (P6: full chunks)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed (the barrier below covers the other waves')
  }
  STAMP(13);
  if (BWD) {   // elements no phase owns (e.g. label-head parameters the main loss does not score): zero gradient
    float* sl = k.slabs + (long long)vblk * k.slab_stride + 1;
    for (int z = 0; z < k.nz; ++z)
      for (int i = k.zlo[z] + tid; i < k.zhi[z]; i += NT) sl[i] = 0.f;
  }
  __syncthreads();
  STAMP(14);
  if (!COLDG && tid < 32) s_wt[tid] = (tid < H) ? s_par[k.o_wh + tid * (1 + L)] : 0.f;  // time column of dynamics_hidden (col 0)
  if (ENCF) {   // head layers (models/encoder_conv.py:49-51): 16 lanes per output, xor butterfly inside the lane group
    const int o = tid >> 4, l16 = tid & 15, which = o / L, l = o - which * L;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = fmaf((l16 + 16 * q < k.Hc) ? e_hw[q] : 0.f, s_ehid[min(l16 + 16 * q, k.Hc - 1)], acc);
    acc = row16_sum(acc);   // (four DPP adds: the 16 lanes of an output are one DPP row)
    acc += e_hb;
    if (l16 == 0 && o < 2 * L) s_pf[which * pad4(L) + l] = which ? expf(acc) : acc;
    BAR();
  }

  float loss_acc = 0.f;   // the only value a thread carries from one trajectory to the next
  static_assert(H < 32, "the hidden units and the constant-1 bias unit share one 32-lane group");
  constexpr int SP = 2 * S + 1;         // stage-0 exchange rows [a (S) | d (S) | pad]: an odd pitch, lane n <-> row n is bank-conflict free (12: 4-way)
  constexpr int GP = 2 * S;             // pitch of the per-sample gradient rows
  const int hg_base = 64;  // head-grad role lives on waves >= 1 (the launcher guarantees NT >= 64 + roundup64(Q*C*S))
  const int n_headw = Q * C * S;
  int hsplit = (NT - hg_base) / n_headw;  // time range split over hsplit threads per head-weight entry
  hsplit = hsplit > 4 ? 4 : hsplit;
  const int NQ = NT / (2 * S);                          // sample chunks of the backward contraction
  const int CL = (n_stage_t + NQ - 1) / NQ;             // samples per chunk
  int n_it = 1;                                         // bisection steps that cover [0, nt)
  while ((1 << n_it) < n_stage_t) ++n_it;
  // the stage-time table must be monotone (torchdiffeq rejects anything else: "t must be strictly increasing or decreasing"); every
  // unit's relu is then switched over a prefix or a suffix of it.  A table that is not turns the loss into NaN (checked in P1).
  bool ts_bad = false;
  STAMP(1);

  for (int b = vblk; b < k.B; b += vgrid) {
    // Launder the thread id once per trajectory: with it opaque, the compiler cannot hoist the dozens of per-thread
    // address computations of the phases below out of this loop (which only lengthens live ranges and spills).
    int tid = tid_outer;
    if (!ONE) asm volatile("" : "+v"(tid));
    // Every gradient element of the segment has exactly one owning thread per trajectory and goes straight to this workgroup's
    // slab (no LDS copy): written on the workgroup's first trajectory, added to on later ones (same owner, program order).
    float* const sl1 = k.slabs + (long long)vblk * k.slab_stride + 1;
    const bool first_traj = ONE || b == vblk;
    auto accum = [&](int idx, float v) { float* d = sl1 + idx; *d = first_traj ? v : (*d + v); };
    if (COLDB && !ONE && b != vblk) {
      BAR();   // the previous trajectory's last readers of the work block (its encoder-head block) are done
      stage_cold();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!ONE && b != vblk) BAR();   // s_pf / s_uu of this trajectory are in place (first one: the setup barrier)

    // ---- P0a: latent sample, log q, log p (mechanistic_cvs.py:125-135, 225-237) -------------------------
    if (tid < L) {
      const int l = tid;
      if (k.loc != nullptr) {
        const float loc = s_pf[l], sc = s_pf[pad4(L) + l], e = s_pf[2 * pad4(L) + l];
        s_gpl[L + l] = e;   // kept for the latent gradient in P7
        s_gls[L + l] = sc;
        const float z = fmaf(sc, e, loc);
        float pl = 0.f, pls = 0.f;
        {
          const int4 m0 = reinterpret_cast<const int4*>(s_meta)[2 * l], m1 = reinterpret_cast<const int4*>(s_meta)[2 * l + 1];
          if (m0.x) {
            if (COLDG && !COLDB) {   // all of the two weight rows in flight at once (global memory), then the dot products
              float wl[SLODE_MAX_NU], ws[SLODE_MAX_NU];
#pragma unroll
              for (int q = 0; q < SLODE_MAX_NU; ++q) {
                wl[q] = cpar[m0.w + min(q, m1.z - 1)];
                ws[q] = cpar[m1.x + min(q, m1.z - 1)];
              }
              pl = cpar[m0.y];
              pls = cpar[m0.z];
#pragma unroll
              for (int q = 0; q < SLODE_MAX_NU; ++q) {
                const float uv = (q < m1.z) ? s_uu[m1.y + min(q, m1.z - 1)] : 0.f;
                pl = fmaf(wl[q], uv, pl);
                pls = fmaf(ws[q], uv, pls);
              }
            } else {
              pl = cp0[m0.y];
              pls = cp0[m0.z];
              for (int q = 0; q < m1.z; ++q) {
                const float uv = s_uu[m1.y + q];
                pl = fmaf(cp0[m0.w + q], uv, pl);
                pls = fmaf(cp0[m1.x + q], uv, pls);
              }
            }
          }
        }
        const float ips = expf(-pls);  // 1 / prior scale
        const float dz = (z - pl) * ips;
        const float zq = (z - loc) / sc;
        const float HL2PI = 0.91893853320467274178f;
        const float log_q = -logf(sc) - HL2PI - 0.5f * zq * zq;
        const float log_p = -pls - HL2PI - 0.5f * dz * dz;
        loss_acc += log_q - log_p;
        s_z[l] = z;
        s_gzl[l] = dz * ips;        // d(-log p)/dz
        s_gpl[l] = -dz * ips;       // d(-log p)/d prior loc
        s_gls[l] = 1.f - dz * dz;   // d(-log p)/d prior log-scale
        if (k.z_out) k.z_out[(long long)b * L + l] = z;
      } else {
        s_z[l] = s_pf[l];
        s_gzl[l] = 0.f;
      }
    }
    STAMP(15);
    // (P0a, P0b and the table part of P0c run on wave 0 only unless label heads are scored here: a wave's LDS accesses execute in
    //  program order, so the workgroup barriers in between are only needed for the label-head threads of the proc family)
    if (COLDG || k.n_aux > 0) BAR();
    // ---- P0b: u = W_z z + b_h (time-invariant part of the hidden layer), init-net hidden; each unit's switching index ------
    float uj = 0.f;
    if (tid < 64) {
      // the two L-long dot products of hidden unit j run side by side: lane j the dynamics' hidden layer, lane 32 + j the init net's
      const int j = tid & 31;
      const bool init_net = tid >= 32;
      float acc = 0.f;
      if (j < H && !ext) {   // (the scorer of an external solution has no use for either: zeros keep the shared gradient code finite)
        acc = s_par[(init_net ? hb1 : hbh) + j];
        const float* row = cp0 + (init_net ? k.o_w1 + j * L : k.o_wh - sh_wh + j * (1 + L) + 1);
#pragma unroll 4
        for (int l = 0; l < L; ++l) acc = fmaf(row[l], s_z[l], acc);
      }
      if (init_net) {
        if (j < H) { s_pre0[j] = acc; s_hid0[j] = fmaxf(acc, 0.f); }
      } else {
        uj = acc;
      }
    }
    // (the label heads run on the waves BEHIND wave 0 when the workgroup has them to spare -- the proc shapes: 3 waves, 4 heads -- so that
    //  wave 0's serial stretch is the two dot products and the switching indices only; they read z (behind the barrier above) and leave
    //  results that P7 picks up many barriers later)
    const int aux_t0 = (NT >= 64 + k.n_aux * 32) ? 64 : 0;
    if (k.n_aux > 0 && tid >= aux_t0 && tid < aux_t0 + k.n_aux * 32) {
      // q(label | z_g) on the replayed z at aux_mult x (mechanistic_proc.py:145-146,334-353): half-wave = head, lane j = hidden unit.  The sums
      // over hidden units are xor-butterflies inside the half-wave (offsets 16..1); every lane of a head then holds its few logits and
      // works out log p and dLoss/dlogit redundantly -- no serial per-head thread (round 2: one thread per head walked U x u_dim
      // dependent FMAs three times over, 5.8 us of the proc trajectory's 30), no exchange, no barrier.
      const int hd = (tid - aux_t0) >> 5, j = tid & 31;
      const slode_aux ax = k.aux[hd];
      const bool on = j < k.U;
      float hvv = 0.f, dv = 0.f;
      if (on) {
        float pre = cp0[k.o_aux_b1[hd] - sh_aux + j];
        if (COLDG && !COLDB) {
          float rw[16];   // (a head reads at most 16 latent dims: slode_launch_ode dispatches this instantiation only then)
#pragma unroll
          for (int l = 0; l < 16; ++l) rw[l] = cpar[k.o_aux_w1[hd] + j * ax.z_dim + min(l, ax.z_dim - 1)];
#pragma unroll
          for (int l = 0; l < 16; ++l) pre = fmaf((l < ax.z_dim) ? rw[l] : 0.f, s_z[ax.z_off + min(l, ax.z_dim - 1)], pre);
        } else {
          for (int l = 0; l < ax.z_dim; ++l) pre = fmaf(cp0[k.o_aux_w1[hd] - sh_aux + j * ax.z_dim + l], s_z[ax.z_off + l], pre);
        }
        hvv = softplusf(pre);
        dv = 1.f / (1.f + expf(-pre));   // softplus'
      }
      constexpr int QM = 8;              // label columns of one head (check_shape)
      float lg[QM], go[QM], w2c[QM];
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        lg[q] = 0.f; go[q] = 0.f;
        w2c[q] = (on && q < ax.u_dim) ? cp0[k.o_aux_w2[hd] - sh_aux + min(q, ax.u_dim - 1) * k.U + j] : 0.f;
        if (q < ax.u_dim) {   // (uniform per half-wave)
          const float v = half_wave_sum(w2c[q] * hvv);
          lg[q] = v + cp0[k.o_aux_b2[hd] - sh_aux + q];
        }
      }
      float lp = 0.f, gcst = 0.f;
      if (ax.kind == SLODE_AUX_SOFTMAX) {
        float mx = -3.0e38f, ysum = 0.f, se = 0.f;
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) mx = fmaxf(mx, lg[q]);
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) { se += expf(lg[q] - mx); ysum += s_uu[ax.u_off + q]; }
        const float lse = mx + logf(se);
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) {
          const float lq = lg[q] - lse, y = s_uu[ax.u_off + q];
          lp = fmaf(y, lq, lp);
          go[q] = k.aux_mult * (expf(lq) * ysum - y);
        }
      } else if (ax.kind == SLODE_AUX_SIGMOID) {
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) {
          const float o = lg[q], y = s_uu[ax.u_off + q];
          const float sp_pos = (o > 0.f ? o : 0.f) + log1pf(expf(-fabsf(o)));  // softplus(o), stable
          lp += y * (o - sp_pos) + (1.f - y) * (-sp_pos);
          go[q] = k.aux_mult * (1.f / (1.f + expf(-o)) - y);
        }
      } else {  // EXPEXP: Laplace(loc = exp(head 0), b = softplus(constant_std_*)); the second Exp head is unused
        const float c = cp0[k.o_aux_c[hd] - sh_aux];
        const float bsc = softplusf(c), ib = 1.f / bsc;
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) {
          const float loc = expf(lg[q]), y = s_uu[ax.u_off + q];
          const float r = y - loc, ar = fabsf(r);
          lp += -logf(2.f * bsc) - ar * ib;
          const float sg = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
          go[q] = -k.aux_mult * sg * ib * loc;
          gcst += k.aux_mult * (ib - ar * ib * ib);
        }
        gcst = gcst / (1.f + expf(-c));
      }
      float gh = 0.f;   // back through the output layer and the Softplus (P7 reads s_auxd again)
#pragma unroll
      for (int q = 0; q < QM; ++q) gh = fmaf(w2c[q], go[q], gh);
      if (on) { s_auxh[hd * 32 + j] = hvv; s_auxd[hd * 32 + j] = dv * gh; }
      if (j == 0) {
        loss_acc -= k.aux_mult * lp;
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < ax.u_dim) s_auxgo[hd * 12 + q] = go[q];
        if (ax.kind == SLODE_AUX_EXPEXP) s_auxgo[hd * 12 + 8] = gcst;
      }
    }
    if (tid < 32) {
      const int j = tid;
      s_u[j] = uj;
      if (!ext && (ALG == 0 || BWD)) {
        // relu(w_t t + u_j) is on exactly where fma(w_t, t, u_j) > 0 (the predicate the gradient uses too); fma is monotone in t and the
        // table is monotone, so the predicate flips at most once along it: bisection on the predicate itself.
        //   sf = 1: on for m >= ms (ms = 0: always, ms = nt: never);   sf = 0: on for m < ms.
        const float wtj = s_wt[j];
        const int nt = n_stage_t;
        const float t_first = s_ts[0], t_last = s_ts[nt - 1];
        const bool p_first = fmaf(wtj, t_first, uj) > 0.f, p_last = fmaf(wtj, t_last, uj) > 0.f;
        const bool flips = p_first != p_last;
        int hi = nt - 1;
        bool need = flips;
        if (nt >= 16) {
          // a near-uniform table puts the flip next to the interpolated index: eight probes around it, accepted only if they bracket it
          const float gf = (-uj / wtj - t_first) / (t_last - t_first) * (float)(nt - 1);
          const int c0 = min(max((int)gf - 3, 0), nt - 8);
          float tp[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) tp[i] = s_ts[flips ? c0 + i : i];
          int cnt = 0;
          bool pa = p_first, pb = p_last;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool pi = fmaf(wtj, tp[i], uj) > 0.f;
            cnt += (pi == p_first) ? 1 : 0;
            if (i == 0) pa = pi;
            if (i == 7) pb = pi;
          }
          if (flips && pa == p_first && pb == p_last) { hi = c0 + cnt; need = false; }
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {   // irregular tables: plain bisection on the predicate
          int lo = 0;
          hi = need ? nt - 1 : hi;
          for (int it = 0; it < n_it; ++it) {
            const int mid = (lo + hi) >> 1;
            const bool pm = fmaf(wtj, s_ts[mid], uj) > 0.f;
            const bool go = need && hi - lo > 1;
            if (go && pm == p_last) hi = mid;
            if (go && pm != p_last) lo = mid;
          }
        }
        int ms = hi, sf = p_last ? 1 : 0;
        if (!flips) { sf = 1; ms = p_first ? 0 : nt; }
        if (j >= H) { sf = 1; ms = nt; }
        s_ms[j] = ms;
        s_sf[j] = sf;
        if (ALG == 0) {
          // rank of this unit's switching index among the 32 lanes (ties by lane; idle lanes sort last): the events in table order
          const int key = ms * 32 + j;
          int rk = 0;
#pragma unroll
          for (int i = 0; i < 32; ++i) rk += (__builtin_amdgcn_readlane(key, i) < key) ? 1 : 0;
          const float tn = s_ts[min(ms, nt - 1)];
          s_ps[rk] = ms;
          s_ord[rk] = j;
          s_sgs[rk] = sf ? 1.f : -1.f;
          s_ewt[rk] = wtj;
          s_epre[rk] = fmaf(wtj, tn, uj);
          s_epre0[rk] = fmaf(wtj, t_first, uj);
          {
            const float sg = sf ? 1.f : -1.f;
            const int jc = min(j, H - 1);
            float wc[2 * S];
#pragma unroll
            for (int c = 0; c < 2 * S; ++c) wc[c] = s_par[(c < S ? hwg + c * H : hwd + (c - S) * H) + jc];
#pragma unroll
            for (int c = 0; c < 2 * S; ++c) s_esw[rk * 2 * S + c] = (j < H) ? sg * wc[c] : 0.f;
          }
          s_tau[rk + 1 < 32 ? rk + 1 : 0] = (rk + 1 < 32) ? tn : t_first;   // segment k+1 starts at event k's time; s_tau[0] = first stage time
        }
      }
    }
    STAMP(16);
    // ---- P0c: x0 = sigmoid(W2 relu(.) + b2)  (blackbox_ode.py:19-22)  ||  the piecewise-linear table ------------------------
    if (tid < S) {
      float o = s_par[hb2 + tid];
      const float* w2r = s_par + hw2 + tid * H;
#pragma unroll
      for (int j = 0; j < H; ++j) o = fmaf(w2r[j], s_hid0[j], o);
      const float x0 = sigmoidf_fast(o);
      s_x0[tid] = x0;
    }
    if (ALG == 0 && !ext && tid >= 32 && tid < 32 + 2 * S) {
      // Head c's pre-activation o_c(t) = bias_c + sum_j W_cj relu(w_t,j t + u_j) is continuous and piecewise linear in t; segment k
      // starts at the k-th switching event (table order).  Row k = [value at the segment's first stage time tau_k | slope], advanced
      // event by event in a form whose terms stay at the scale of o_c itself: V += slope (tau' - tau) + sign W_ce pre_e(tau').
      // The rows stay in registers until the chain is done: with no store in the loops every LDS read is issued up front.
      const int c = tid - 32;
      // pass 1: the units that are on over a prefix (sign -1) are on at the first stage time
      float al = 0.f, V = s_par[c < S ? hbg + c : hbd + (c - S)];
#pragma unroll
      for (int k0 = 0; k0 < H; k0 += 8) {
        float sw[8], ew[8], p0[8], sg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kk = min(k0 + q, H - 1);
          sw[q] = s_esw[kk * 2 * S + c]; ew[q] = s_ewt[kk]; p0[q] = s_epre0[kk]; sg[q] = s_sgs[kk];
        }
        __builtin_amdgcn_sched_barrier(0);   // the batch's LDS reads above, the arithmetic below
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float w = -sw[q] * ((k0 + q < H && sg[q] < 0.f) ? 1.f : 0.f);   // sign -1: sw = -W
          al = fmaf(w, ew[q], al);
          V = fmaf(w, p0[q], V);
        }
      }
      float Vr[H + 1], Ar[H + 1];
      Vr[0] = V; Ar[0] = al;
      // pass 2: event by event
#pragma unroll
      for (int k0 = 0; k0 < H; k0 += 8) {
        float sw[8], ew[8], pe[8], ta[9];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kk = min(k0 + q, H - 1);
          sw[q] = s_esw[kk * 2 * S + c]; ew[q] = s_ewt[kk]; pe[q] = s_epre[kk]; ta[q] = s_tau[min(k0 + q, H)];
        }
        ta[8] = s_tau[min(k0 + 8, H)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (k0 + q < H) {
            V = fmaf(al, ta[q + 1] - ta[q], V);
            V = fmaf(sw[q], pe[q], V);
            al = fmaf(sw[q], ew[q], al);
            Vr[k0 + q + 1] = V; Ar[k0 + q + 1] = al;
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk <= H; ++kk) {
        s_tab[kk * 4 * S + c] = Vr[kk];
        s_tab[kk * 4 * S + 2 * S + c] = Ar[kk];
      }
    }
    BAR();   // table complete (ALG 0); u, w_t in place for every wave

    STAMP(2);
    // ---- P1: stage evaluations + step coefficients (thread n <-> grid step n) ---------------------------
    float av[3][S], dv[3][S];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int s = 0; s < S; ++s) { av[r][s] = 0.f; dv[r][s] = 0.f; }
    }
    const int n = tid;
    const bool own_step = n < T - 1;
    const bool own_last = (n == T - 1) && (uses_next || (BWD && need_next));
    if (!ext && own_step) {   // this step's stretch of the stage-time table runs in the table's overall direction
      const bool inc = s_ts[n_stage_t - 1] >= s_ts[0];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r < R) {
          const float d = s_ts[R * n + r + 1] - s_ts[R * n + r];
          if (inc ? !(d >= 0.f) : !(d <= 0.f)) ts_bad = true;
        }
      }
    }
    if (!ext) {
      if (ALG == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (r < R) {   // wave-uniform
            const int mm = min(R * n + r, n_stage_t - 1);
            const float t = s_ts[mm];
            int kseg = 0;   // number of switching events at or before sample mm = the table row
#pragma unroll
            for (int stp = 16; stp >= 1; stp >>= 1)
              if (s_ps[kseg + stp - 1] <= mm) kseg += stp;
            const float dtk = t - s_tau[kseg];
            const f32x4* row = reinterpret_cast<const f32x4*>(s_tab + kseg * 4 * S);
            float rv[4 * S];
#pragma unroll
            for (int q4 = 0; q4 < S; ++q4) {
              const f32x4 v4 = row[q4];
              rv[4 * q4] = v4.x; rv[4 * q4 + 1] = v4.y; rv[4 * q4 + 2] = v4.z; rv[4 * q4 + 3] = v4.w;
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
              av[r][s] = sigmoidf_fast(fmaf(rv[2 * S + s], dtk, rv[s]));
              dv[r][s] = sigmoidf_fast(fmaf(rv[3 * S + s], dtk, rv[S + s]));
            }
          }
        }
      } else {
        // every lane evaluates (idle lanes on a clamped time): under a divergent branch the compiler hoisted all R x 2*S*H weight
        // s_loads above the branch and spilled them through VGPR lanes
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (r < R) {   // wave-uniform
            eval_ad<S, H>(s_ts[min(R * n + r, n_stage_t - 1)], s_wt, s_u, wg, bg, wd, bd, av[r], dv[r]);
            __builtin_amdgcn_sched_barrier(0);  // do not interleave the R evaluations (3x the live registers)
          }
        }
      }
      if (own_step || own_last) {
        if (uses_next) {
#pragma unroll
          for (int s = 0; s < S; ++s) {
            s_st[n * SP + s] = av[0][s];
            s_st[n * SP + S + s] = dv[0][s];
          }
        }
      }
      BAR();
      STAMP(3);
      if (own_step) {
        const float h = s_ts[R * (n + 1)] - s_ts[R * n];   // == times[n+1] - times[n]: stage 0 of a step sits on its node
#pragma unroll
        for (int s = 0; s < S; ++s) {
          float a3 = 0.f, d3 = 0.f;
          if (uses_next) {
            a3 = s_st[(n + 1) * SP + s];
            d3 = s_st[(n + 1) * SP + S + s];
          }
          const float a4[4] = {av[0][s], av[1][s], av[2][s], a3};
          const float d4[4] = {dv[0][s], dv[1][s], dv[2][s], d3};
          float A, bb;
          step_fwd(method, h, a4, d4, A, bb);
          s_A[n * S + s] = A;
          s_x[(n + 1) * S + s] = bb;
        }
      }
      if (tid < S) s_x[tid] = s_x0[tid];
      BAR();
    }
    STAMP(4);
    // this trajectory's observation column (thread t <-> time point t): in flight during the scan
    float pf_ob0 = 0.f, pf_ob1 = 0.f, pf_ob2 = 0.f, pf_ob3 = 0.f;
    if (k.with_ll && tid < T) {
      const float* op_ = k.obs + (long long)b * k.sb + (long long)tid * k.st;
      pf_ob0 = op_[0];
      pf_ob1 = op_[(long long)min(1, C - 1) * k.sc];
      pf_ob2 = op_[(long long)min(2, C - 1) * k.sc];
      pf_ob3 = op_[(long long)min(3, C - 1) * k.sc];
    }
    // ---- P2: forward scan x_{n+1} = A_n x_n + b_n (the only serial part of the solve) -------------------
    if (!ext) {
      constexpr int NTc = T_ ? ode_threads_for(T_ ? T_ : 1, Q_ ? Q_ : 1, C_ ? C_ : 1, S) : 64;
      constexpr int CLc = T_ ? ((T_ - 1) + (NTc / 64) * SCAN_NCH - 1) / ((NTc / 64) * SCAN_NCH) : 8;   // steps per lane of the forward scan
      block_affine_scan<S, false, (CLc < 8 ? CLc : 8)>(s_A, s_x, T, tid, NT, s_ct, BAR);   // (the chunk-sum buffer of P6 carries the waves' total maps)
    } else {   // score the adaptive solver's trajectory instead
      const float* xe = k.x_ext + (long long)b * T * S;
      for (int i = tid; i < T * S; i += NT) s_x[i] = xe[i];
    }
    // what P3 needs of the likelihood scales besides the scale itself comes from the per-step table when there is one: issued ahead of
    // the barrier so that the loads fly while the scan finishes
    float t_inv[SLODE_MAX_C] = {0.f, 0.f, 0.f, 0.f}, t_lg[SLODE_MAX_C] = {0.f, 0.f, 0.f, 0.f}, t_ds[SLODE_MAX_C] = {0.f, 0.f, 0.f, 0.f};
    const bool use_tab = k.with_ll && k.sigtab != nullptr;
    if (use_tab) {
      const int tt = min(tid, T - 1), CTn = C * T;
#pragma unroll
      for (int c = 0; c < SLODE_MAX_C; ++c) {
        const int i = min(c, C - 1) * T + tt;
        t_inv[c] = k.sigtab[CTn + i]; t_lg[c] = k.sigtab[2 * CTn + i]; t_ds[c] = k.sigtab[3 * CTn + i];
      }
    }
    BAR();
    STAMP(5);
    if (k.x_out) {
      float* xo = k.x_out + (long long)b * T * S;
      for (int i = tid; i < T * S; i += NT) xo[i] = s_x[i];
    }

    // ---- P3: decoder heads + likelihood + dLoss/dx (decoders.py:45-53; mechanistic_cvs.py:142-211) -------
    if (k.with_ll) {
      if (tid < T) {
        const int t = tid;
        float xs[S], gx[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { xs[s] = s_x[t * S + s]; gx[s] = 0.f; }
        float ll = 0.f;
#pragma unroll
        for (int c = 0; c < SLODE_MAX_C; ++c) {
          if (c >= C) continue;
          const float sig = ONE ? sigr[c] : s_sig[c * T + t];
          const float inv = use_tab ? t_inv[c] : 1.0f / sig;
          const float lg = use_tab ? t_lg[c] : (gauss ? logf(sig) : logf(2.f * sig));
          const float obv = (c == 0) ? pf_ob0 : ((c == 1) ? pf_ob1 : ((c == 2) ? pf_ob2 : pf_ob3));
          float gsig = 0.f;
          for (int q = 0; q < Q; ++q) {
            const cptr W = (cptr)k.head[q] + c * S;
            float mu = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) mu = fmaf(W[s], xs[s], mu);
            const float r = obv - mu;
            float gmu;
            if (gauss) {
              ll += -lg - 0.91893853320467274178f - 0.5f * r * r * inv * inv;
              gmu = -r * inv * inv;
              gsig += inv - r * r * inv * inv * inv;
            } else {
              const float w = (obv >= mu) ? k.tau[q] : 1.f - k.tau[q];
              const float ar = fabsf(r);
              ll += w * (-lg - ar * inv);
              const float sg = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
              gmu = -w * sg * inv;
              gsig += w * (inv - ar * inv * inv);
            }
            if (BWD) {
#pragma unroll
              for (int s = 0; s < S; ++s) gx[s] = fmaf(gmu, W[s], gx[s]);
              s_st[(q * C + c) * T + t] = gmu;
            }
          }
          if (BWD) {  // constant_std gradient: thread t owns slab entry (c, t); softplus'(x) = 1 - exp(-softplus(x))
            float* dst = k.slabs + (long long)vblk * k.slab_stride + 1 + k.o_cstd + c * T + t;
            const float val = gsig * (use_tab ? t_ds[c] : 1.f - expf(-sig));
            *dst = (ONE || b == vblk) ? val : (*dst + val);
          }
        }
        loss_acc -= ll;
        if (BWD) {
          if (ext) {   // dLoss/dx goes to the adaptive solver's backward pass
#pragma unroll
            for (int s = 0; s < S; ++s) {
              if (k.gx_out) k.gx_out[((long long)b * T + t) * S + s] = gx[s];
            }
          } else {
#pragma unroll
            for (int s = 0; s < S; ++s) s_lam[t * S + s] = gx[s];
          }
        }
      }
    } else if (BWD) {
      const float* gi = k.gx_in + (long long)b * T * S;
      for (int i = tid; i < T * S; i += NT) s_lam[i] = gi[i];
    }

    if (BWD) {
      if (RA && !ext) {
        // reference_adjoint: the adjoint recurrence runs on the backward step maps M_n (see radj_M), exchanged / stored through s_A
        // (the forward A is dead; the stage buffer still holds P3's dLoss/dmu for the head-gradient role)
        BAR();
        if (own_step || own_last) {
#pragma unroll
          for (int s = 0; s < S; ++s) s_A[n * S + s] = dv[0][s];
        }
        BAR();
        float Mn[S];
        if (own_step) {
          const float hb = -(s_ts[R * (n + 1)] - s_ts[R * n]);
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float D[4] = {s_A[(n + 1) * S + s], method == SLODE_RK4 ? dv[2][s] : dv[1][s], dv[1][s], dv[0][s]};
            Mn[s] = radj_M(method, hb, D);
          }
        }
        BAR();
        if (own_step) {
#pragma unroll
          for (int s = 0; s < S; ++s) s_A[n * S + s] = Mn[s];
        }
      }
      BAR();
      STAMP(6);
      // ---- P4: adjoint scan (wave 0) || head-weight gradients (waves >= 1) ------------------------------
      if (!ext && tid < 64) wave_affine_scan<S, true, wave_scan_cl(T_)>(s_A, s_lam, T, tid);
      if (k.with_ll) {
        const int e = tid - hg_base;
        if (e >= 0 && e < n_headw * hsplit) {
          const int part = e / n_headw, ew = e - part * n_headw;
          const int qc = ew / S, s = ew - qc * S;
          const int tper = (T + hsplit - 1) / hsplit, t0 = part * tper, t1 = min(T, t0 + tper);
          float a0 = 0.f, a1 = 0.f;
          int t = t0;
          for (; t + 7 < t1; t += 8) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              a0 = fmaf(s_st[qc * T + t + q], s_x[(t + q) * S + s], a0);
              a1 = fmaf(s_st[qc * T + t + q + 1], s_x[(t + q + 1) * S + s], a1);
            }
            if (S > 5) __builtin_amdgcn_sched_barrier(0);   // one batch of 16 LDS reads at a time (the S = 8 kernels have 168 VGPRs)
          }
          for (; t < t1; ++t) a0 = fmaf(s_st[qc * T + t], s_x[t * S + s], a0);
          s_hp[e] = a0 + a1;
        }
      }
      BAR();
      STAMP(7);
      if (!ext) {
      // ---- P5: reverse mode of the step coefficients (thread n <-> step n) ------------------------------
      // (the stage buffer is free again: re-exchange the first-stage values instead of carrying a3/d3 in registers)
      if (need_next && (own_step || own_last)) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          s_st[n * SP + s] = av[0][s];
          s_st[n * SP + S + s] = dv[0][s];
        }
      }
      BAR();
      float g3a[S], g3d[S];   // contribution to the NEXT step's first stage (shared evaluation a(t_{n+1}))
#pragma unroll
      for (int s = 0; s < S; ++s) { g3a[s] = 0.f; g3d[s] = 0.f; }
      if (own_step) {
        const float h = s_ts[R * (n + 1)] - s_ts[R * n];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float gb = s_lam[(n + 1) * S + s];
          const float gA = gb * s_x[n * S + s];
          float a3 = 0.f, d3 = 0.f;
          if (need_next) {
            a3 = s_st[(n + 1) * SP + s];
            d3 = s_st[(n + 1) * SP + S + s];
          }
          const float a4[4] = {av[0][s], av[1][s], av[2][s], a3};
          const float d4[4] = {dv[0][s], dv[1][s], dv[2][s], d3};
          float ga[4], gd[4];
          if (RA) {
            // backward stages: node n+1, then this step's forward stages in reverse; gb = lambda at node n+1 (after its jump)
            const int r2 = method == SLODE_RK4 ? 2 : 1;
            const float Ab[4] = {a3, a4[r2], a4[1], a4[0]}, Db[4] = {d3, d4[r2], d4[1], d4[0]};
            float gaq[4], gdq[4];
            radj_stage_grads(method, -h, Ab, Db, gb, s_x[(n + 1) * S + s], gaq, gdq);
            // back to the forward slots (0..2 = this step's stages, 3 = node n+1)
            ga[3] = gaq[0]; gd[3] = gdq[0];
            ga[0] = ga[1] = ga[2] = 0.f; gd[0] = gd[1] = gd[2] = 0.f;
            if (method == SLODE_RK4) { ga[2] = gaq[1]; gd[2] = gdq[1]; ga[1] = gaq[2]; gd[1] = gdq[2]; ga[0] = gaq[3]; gd[0] = gdq[3]; }
            else if (method == SLODE_MIDPOINT) { ga[1] = gaq[1]; gd[1] = gdq[1]; }
          } else {
            step_bwd(method, h, a4, d4, gA, gb, ga, gd);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // through the sigmoids
            ga[r] *= a4[r] * (1.f - a4[r]);
            gd[r] *= d4[r] * (1.f - d4[r]);
          }
          av[0][s] = ga[0]; av[1][s] = ga[1]; av[2][s] = ga[2];
          dv[0][s] = gd[0]; dv[1][s] = gd[1]; dv[2][s] = gd[2];
          g3a[s] = ga[3]; g3d[s] = gd[3];
          if (S > 5) __builtin_amdgcn_sched_barrier(0);   // one state component at a time: eight interleaved reverse-mode chains cost 12 VGPRs too many
        }
      } else if (own_last) {
#pragma unroll
        for (int s = 0; s < S; ++s) { av[0][s] = 0.f; dv[0][s] = 0.f; }
      }
      if (tid >= 64 && tid < 128) {  // wave 1 (NT >= 128 is guaranteed): dLoss/d(x0 pre-activation), before the adjoint is overwritten,
        const int j = tid - 64;      // then back through the init net's output layer (same wave: in-order LDS)
        if (j < S) {
          const float x0 = s_x0[j];
          s_go[j] = s_lam[j] * x0 * (1.f - x0);
        }
        if (j < 32) {
          float gh0 = 0.f;
          if (j < H) {
#pragma unroll
            for (int s = 0; s < S; ++s) gh0 = fmaf(s_par[hw2 + s * H + j], s_go[s], gh0);
            gh0 = (s_pre0[j] > 0.f) ? gh0 : 0.f;
          }
          s_gp0[j] = gh0;
        }
      }
      BAR();   // every read of A | x | lam | st is done: the block becomes the sample rows G[nt][2S]
      if (tid < ode_pad_rows(n_stage_t, NT, S) * 2 * S) s_G[n_stage_t * GP + tid] = 0.f;   // zero pad rows: every P6 chunk is full
      if (need_next && own_step) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          s_G[R * (n + 1) * GP + s] = g3a[s];
          s_G[R * (n + 1) * GP + S + s] = g3d[s];
        }
      }
      BAR();
      if (own_step || own_last) {
        if (need_next && n >= 1) {
#pragma unroll
          for (int s = 0; s < S; ++s) {
            av[0][s] += s_G[R * n * GP + s];
            dv[0][s] += s_G[R * n * GP + S + s];
          }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          if (r < R && (own_step || r == 0)) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
              s_G[(R * n + r) * GP + s] = av[r][s];
              s_G[(R * n + r) * GP + S + s] = dv[r][s];
            }
          }
        }
      } else if (n == T - 1) {   // no shared node evaluation: the last table entry carries no gradient
#pragma unroll
        for (int c = 0; c < 2 * S; ++c) s_G[(n_stage_t - 1) * GP + c] = 0.f;
      }
      BAR();
      STAMP(8);
      // ---- P6: contraction of the sample rows with the hidden layer -------------------------------------
      //   GM[r][j] = sum_{m: unit j on} g[m][r],  GT[r][j] = sum_{m: unit j on} g[m][r] t_m   (unit H = constant 1: the head biases)
      //   dW[r][j] = w_t,j GT + u_j GM;   dLoss/du_j = sum_r W[r][j] GM[r][j];   dLoss/dw_t,j = sum_r W[r][j] GT[r][j]
      if (ALG != 2) {
        // (A) chunk sums: lane (chunk q, channel r) adds up CL consecutive samples.  Loads go in batches of 8 samples with a
        // scheduling barrier behind them (left alone, the scheduler waits for each LDS read in turn); a 0/1 mask keeps them
        // unconditional (a select on a loaded value is turned into a branch + wait).
        // Shape-specialised instantiations whose chunks are full (pad rows): compile-time chunk length, immediate offsets, no bounds --
        // the same additions in the same order as the general form below.
        constexpr bool STATIC_SHAPE = T_ > 0 && M_ >= 0;
        const bool fullc = STATIC_SHAPE && ode_full_chunks(n_stage_t, NT, S);
        if (fullc && tid < NQ * 2 * S) {
          const int q = tid / (2 * S), r = tid - q * (2 * S);
          const float* gp = s_G + q * CL * GP + r;
          const float* tp = s_ts + q * CL;
          float cg = 0.f, cgt = 0.f;
#pragma unroll
          for (int i0 = 0; i0 < CL; i0 += 8) {
            float gv[8], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              gv[u] = (i0 + u < CL) ? gp[(i0 + u) * GP] : 0.f;
              tv[u] = (i0 + u < CL) ? tp[i0 + u] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (i0 + u < CL) { cg += gv[u]; cgt = fmaf(gv[u], tv[u], cgt); }
            }
          }
          s_ct[q * 4 * S + r] = cg;
          s_ct[q * 4 * S + 2 * S + r] = cgt;
        } else if (tid < NQ * 2 * S) {
          const int q = tid / (2 * S), r = tid - q * (2 * S);
          float cg = 0.f, cgt = 0.f;
          for (int i0 = 0; i0 < CL; i0 += 8) {
            float gv[8], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int mc = min(q * CL + i0 + u, n_stage_t - 1);
              gv[u] = s_G[mc * GP + r]; tv[u] = s_ts[mc];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float g = gv[u] * ((i0 + u < CL && q * CL + i0 + u < n_stage_t) ? 1.f : 0.f);
              cg += g;
              cgt = fmaf(g, tv[u], cgt);
            }
          }
          s_ct[q * 4 * S + r] = cg;
          s_ct[q * 4 * S + 2 * S + r] = cgt;
        }
        BAR();
        // (A') shape-specialised kernels: one lane per column turns the chunk sums into exclusive prefix sums (in place; row NQ = the
        // total) and suffix sums (in the event arrays of P0c / P1, dead by now), each column held in registers meanwhile -- (B) then
        // reads ONE row instead of adding up to NQ.  Still only additions on a unit's "on" side.
        const bool scanq = fullc && NQ <= 32 && (NQ + 1) * 4 * S <= 7 * 32 + 32 * 2 * S;
        float* s_sx = s_tau;   // [NQ + 1][4S] suffix sums over tau | ps | ord | sgs | ewt | epre | epre0 | esw
        if (scanq) {
          if (tid < 4 * S) {
            float cq[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) cq[q] = s_ct[min(q, NQ - 1) * 4 * S + tid];
            __builtin_amdgcn_sched_barrier(0);
            float run = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q)
              if (q < NQ) { s_ct[q * 4 * S + tid] = run; run += cq[q]; }
            s_ct[NQ * 4 * S + tid] = run;
            float rs = 0.f;
            s_sx[NQ * 4 * S + tid] = 0.f;
#pragma unroll
            for (int q = 31; q >= 0; --q)
              if (q < NQ) { rs += cq[q]; s_sx[q * 4 * S + tid] = rs; }
          }
          BAR();
        }
        // (B) lane (unit j, channel r): whole chunks on the unit's "on" side + the samples of the chunk its switching index cuts.
        //     Only additions on the "on" side: no total-minus-prefix cancellation.  Fixed trip counts, masked adds.
        for (int e = tid; e < H * 2 * S; e += NT) {
          const int j = e / (2 * S), r = e - j * (2 * S);
          const int ms = s_ms[j], sf = s_sf[j];
          const int qs = ms / CL;          // chunk that holds sample ms (>= NQ: none)
          const int qa = sf ? qs + 1 : 0, qn = max(sf ? NQ - qa : qs, 0);    // whole chunks [qa, qa + qn)  (a never-on unit: ms = nt, none)
          const int ma = sf ? ms : qs * CL, mn = max(sf ? (qs + 1) * CL - ms : ms - qs * CL, 0);   // cut chunk: samples [ma, ma + mn)
          float gmv = 0.f, gtv = 0.f;
          if (scanq) {   // whole chunks [qs + 1, NQ) (on from ms upwards) or [0, qs) (on below ms): one row of the suffix / prefix sums
            const float* src = (sf ? s_sx : s_ct) + min(sf ? qs + 1 : qs, NQ) * 4 * S;
            gmv = src[r]; gtv = src[2 * S + r];
          } else
          for (int q0 = 0; q0 < NQ; q0 += 8) {
            float v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int qc = min(q0 + u, NQ - 1);
              v[u] = s_ct[qc * 4 * S + r]; w[u] = s_ct[qc * 4 * S + 2 * S + r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float on = (q0 + u < NQ && (unsigned)(q0 + u - qa) < (unsigned)qn) ? 1.f : 0.f;
              gmv = fmaf(on, v[u], gmv);
              gtv = fmaf(on, w[u], gtv);
            }
          }
          if (fullc) {
            const int qc = min(qs, NQ - 1), rel = ma - qc * CL;   // first selected sample, relative to the chunk
            const float* gp = s_G + qc * CL * GP + r;
            const float* tp = s_ts + qc * CL;
#pragma unroll
            for (int i0 = 0; i0 < CL; i0 += 8) {
              float gv[8], tv[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                gv[u] = (i0 + u < CL) ? gp[(i0 + u) * GP] : 0.f;
                tv[u] = (i0 + u < CL) ? tp[i0 + u] : 0.f;
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                if (i0 + u < CL) {
                  const float g = gv[u] * (((unsigned)(i0 + u - rel) < (unsigned)mn) ? 1.f : 0.f);
                  gmv += g;
                  gtv = fmaf(g, tv[u], gtv);
                }
              }
            }
          } else
          for (int i0 = 0; i0 < CL; i0 += 8) {
            float gv[8], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int mc = min(qs * CL + i0 + u, n_stage_t - 1);
              gv[u] = s_G[mc * GP + r]; tv[u] = s_ts[mc];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int mm = qs * CL + i0 + u;
              const float g = gv[u] * ((i0 + u < CL && mm < n_stage_t && (unsigned)(mm - ma) < (unsigned)mn) ? 1.f : 0.f);
              gmv += g;
              gtv = fmaf(g, tv[u], gtv);
            }
          }
          s_gm[r * 32 + j] = gmv;
          s_gm[(2 * S + r) * 32 + j] = gtv;
        }
        if (tid >= NT - 2 * S) {   // the constant-1 unit: every sample
          const int r = tid - (NT - 2 * S);
          float gmv = 0.f;
          if (scanq) gmv = s_ct[NQ * 4 * S + r];
          else {
#pragma unroll 8
            for (int q = 0; q < NQ; ++q) gmv += s_ct[q * 4 * S + r];
          }
          s_gm[r * 32 + H] = gmv;
        }
      } else {
        // MFMA arm: D[16 x 16] += A[16 x 4] B[4 x 16] with A = [g | 0]^T or [g t | 0]^T (rows = channel) and B = the units' on/off
        // mask of 4 consecutive samples (cols = unit; col H = 1): four tiles per k-step; every wave takes every NW-th k-step.
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
        const int xx = lane & 15, kq = lane >> 4;
        const float wt0 = s_wt[xx], u0 = s_u[xx];
        const int j1 = 16 + xx;
        const float wt1 = (j1 < H) ? s_wt[j1] : 0.f, u1 = (j1 < H) ? s_u[j1] : ((j1 == H) ? 1.f : -1.f);
        f32x4 d00 = {0.f, 0.f, 0.f, 0.f}, d01 = d00, d10 = d00, d11 = d00;
        const int nks = (n_stage_t + 3) >> 2;
        for (int ks = wv; ks < nks; ks += NW) {
          const int mm = 4 * ks + kq, mc = min(mm, n_stage_t - 1);
          float g = s_G[mc * GP + min(xx, 2 * S - 1)];
          g = (xx < 2 * S && mm < n_stage_t) ? g : 0.f;
          const float t = s_ts[mc];
          const float b0 = (fmaf(wt0, t, u0) > 0.f) ? 1.f : 0.f, b1 = (fmaf(wt1, t, u1) > 0.f) ? 1.f : 0.f;
          const float gt = g * t;
          d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(g, b0, d00, 0, 0, 0);
          d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(g, b1, d01, 0, 0, 0);
          d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(gt, b0, d10, 0, 0, 0);
          d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(gt, b1, d11, 0, 0, 0);
        }
        BAR();   // all waves are past their last read of G: the tiles go over it, [wave][tile][reg][lane]
        float* tl = s_G + wv * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tl[(0 * 4 + i) * 64 + lane] = d00[i];
          tl[(1 * 4 + i) * 64 + lane] = d01[i];
          tl[(2 * 4 + i) * 64 + lane] = d10[i];
          tl[(3 * 4 + i) * 64 + lane] = d11[i];
        }
        BAR();
        // D[row = 4 (lane >> 4) + i][col = lane & 15]: element (which, r, j) sits in tile 2 which + (j >> 4), reg r & 3, lane 16 (r >> 2) + (j & 15)
        for (int e = tid; e < 2 * 2 * S * 32; e += NT) {
          const int which = e / (2 * S * 32), rem = e - which * (2 * S * 32), r = rem >> 5, j = rem & 31;
          const int tile = 2 * which + (j >> 4), src = (tile * 4 + (r & 3)) * 64 + 16 * (r >> 2) + (j & 15);
          float v = 0.f;
          for (int w = 0; w < NW; ++w) v += s_G[w * 1024 + src];
          s_gm[(which * 2 * S + r) * 32 + j] = v;
        }
      }
      BAR();
      STAMP(9);
      // (C) into the gradient segment: head weights / biases, dLoss/du (P7 reads it), time column of the hidden layer
      {
        for (int e = tid; e < H * 2 * S; e += NT) {
          const int j = e / (2 * S), r = e - j * (2 * S);
          const float v = fmaf(s_wt[j], s_gm[(2 * S + r) * 32 + j], s_u[j] * s_gm[r * 32 + j]);
          accum((r < S ? k.o_wg + r * H : k.o_wd + (r - S) * H) + j, v);
        }
        if (tid >= NT - 32) {
          const int j = tid - (NT - 32);
          float gu = 0.f, gwt = 0.f;
          if (j < H) {
#pragma unroll
            for (int r = 0; r < 2 * S; ++r) {
              const float w = s_par[(r < S ? hwg + r * H : hwd + (r - S) * H) + j];
              gu = fmaf(w, s_gm[r * 32 + j], gu);
              gwt = fmaf(w, s_gm[(2 * S + r) * 32 + j], gwt);
            }
            accum(k.o_wh + j * (1 + L), gwt);
          }
          s_gu[j] = gu;
        }
        if (tid >= NT - 64 && tid < NT - 64 + 2 * S) {
          const int r = tid - (NT - 64);
          accum(r < S ? k.o_bg + r : k.o_bd + (r - S), s_gm[r * 32 + H]);
        }
      }
      } else {   // ext: no gradient reaches the dynamics or x0 through this kernel (the elements step (C) owns are zeros)
        if (tid < 32) { s_gu[tid] = 0.f; s_gp0[tid] = 0.f; }
        if (tid >= 64 && tid < 64 + S) s_go[tid - 64] = 0.f;
        if (!k.ext_skip) {   // (ext_skip: the reduction behind this launch does not read the solver-side range of these rows)
          for (int e = tid; e < H * 2 * S; e += NT) {
            const int j = e / (2 * S), r = e - j * (2 * S);
            accum((r < S ? k.o_wg + r * H : k.o_wd + (r - S) * H) + j, 0.f);
          }
          if (tid < H) accum(k.o_wh + tid * (1 + L), 0.f);
          if (tid >= 64 && tid < 64 + 2 * S) accum((tid - 64) < S ? k.o_bg + (tid - 64) : k.o_bd + (tid - 64 - S), 0.f);
        }
      }
      BAR();
      STAMP(17);
      // ---- P7: small nets (init net, z-part of the hidden layer, priors) and the latent gradient ----------
      if (tid < 64) {
        // latent gradient on wave 0: lane = (part, l); the sum over hidden units is split over `parts` lane groups
        const int Lp = (L > 32) ? 64 : ((L > 16) ? 32 : ((L > 8) ? 16 : 8)), parts = 64 / Lp;
        const int l = tid & (Lp - 1), part = tid / Lp, lc = min(l, L - 1);
        float gz = 0.f;
        if (COLDG) {   // columns lc of W_z and W_1 from global memory: up to 2 x 13 loads in flight
          for (int j0 = part; j0 < H; j0 += 13 * parts) {
            float a[13], c[13];
#pragma unroll
            for (int q = 0; q < 13; ++q) {
              const int j = min(j0 + q * parts, H - 1);
              a[q] = RA ? 0.f : cpar[k.o_wh + j * (1 + L) + 1 + lc];
              c[q] = cpar[k.o_w1 + j * L + lc];
            }
#pragma unroll
            for (int q = 0; q < 13; ++q) {
              const int j = j0 + q * parts;
              if (j < H) {
                if (!RA) gz = fmaf(a[q], s_gu[j], gz);   // reference_adjoint: z is not an adjoint parameter
                gz = fmaf(c[q], s_gp0[j], gz);
              }
            }
          }
        } else
        for (int j = part; j < H; j += parts) {
          if (!RA) gz = fmaf(s_par[k.o_wh + j * (1 + L) + 1 + lc], s_gu[j], gz);   // reference_adjoint: z is not an adjoint parameter
          gz = fmaf(s_par[k.o_w1 + j * L + lc], s_gp0[j], gz);
        }
        // the `parts` lane groups' partial sums (lane bits >= log2(Lp)): DPP inside a row, v_permlane swaps across rows and halves
        if (Lp <= 8) gz += dpp_f<0x128>(gz);   // row_ror:8
        if (Lp <= 16) {
          const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(gz), __float_as_uint(gz), false, false);
          gz = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        if (Lp <= 32) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(gz), __float_as_uint(gz), false, false);
          gz = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        if (tid < L) {
          gz += s_gzl[l];
          for (int hd = 0; hd < k.n_aux; ++hd) {
            const slode_aux ax = k.aux[hd];
            if (l >= ax.z_off && l < ax.z_off + ax.z_dim)
              for (int j0 = 0; j0 < k.U; j0 += 16) {   // (batched: the weights may sit in global memory)
                float wv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) wv[q] = cpar[k.o_aux_w1[hd] + min(j0 + q, k.U - 1) * ax.z_dim + (l - ax.z_off)];
#pragma unroll
                for (int q = 0; q < 16; ++q) gz = fmaf((j0 + q < k.U) ? wv[q] : 0.f, s_auxd[hd * 32 + min(j0 + q, k.U - 1)], gz);
              }
          }
          if (k.loc != nullptr) {
            const float sc = s_gls[L + l];
            const float gsc = fmaf(gz, s_gpl[L + l], -1.0f / sc);
            if (k.g_loc) {
              k.g_loc[(long long)b * L + l] = gz;
              k.g_scale[(long long)b * L + l] = gsc;
            }
            if (k.g_pre) {  // hand (g_loc, g_scale * scale) to the head-backward block after the trajectory's last barrier
              s_gzl[l] = gz;
              s_gpl[L + l] = gsc * sc;
              k.glat[(long long)b * 128 + l] = gz;
              k.glat[(long long)b * 128 + 64 + l] = gsc * sc;
            }
          } else {
            k.g_loc[(long long)b * L + l] = gz;
          }
        }
        STAMP(19);
      } else {
        // waves >= 1, beside the latent gradient: encoder head weights -> LDS, then the owner-thread accumulation into the LDS
        // gradient segment (unique owner per element => no atomics)
        const int t1 = tid - 64, n1 = NT - 64;
        if (k.stage_encw) {
          // encoder head layers [z_loc.weight | z_loc.bias | z_scale.0.weight] (one contiguous block of the flat vector) -> LDS by
          // LDS-DMA loads (no registers; 64 consecutive floats per wave-instruction), in flight during the accumulation below; every
          // issuing wave drains its own loads (s_waitcnt vmcnt(0)) ahead of the barrier that ends the trajectory
          const int n_encw = 2 * L * k.Hc + L;
          const int w1 = __builtin_amdgcn_readfirstlane((tid >> 6) - 1), nw1 = (NT >> 6) - 1, lane = tid & 63;
          for (int base = w1 * 64; base < n_encw; base += nw1 * 64) {
            if (base + lane < n_encw)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k.enc_zloc_w + base + lane),
                                               (__attribute__((address_space(3))) void*)(s_encw + base), 4, 0, 0);
          }
        }
        if (!(ext && k.ext_skip)) {   // (the scorer of an external solution would write zeros here: init net and dynamics are the reverse sweep's)
          for (int e = t1; e < H * L; e += n1) {
            const int j = e / L, l = e - j * L;
            accum(k.o_wh + j * (1 + L) + 1 + l, s_gu[j] * s_z[l]);
            accum(k.o_w1 + e, s_gp0[j] * s_z[l]);
          }
          for (int e = t1; e < S * H; e += n1) {
            const int s = e / H, j = e - s * H;
            accum(k.o_w2 + e, s_go[s] * s_hid0[j]);
          }
          if (t1 < H) { accum(k.o_bh + t1, s_gu[t1]); accum(k.o_b1 + t1, s_gp0[t1]); }
          if (t1 < S) accum(k.o_b2 + t1, s_go[t1]);
        }
        if (k.with_ll && t1 < n_headw) {   // decoder head weights: the hsplit partials of P4, fixed order
          const int qc = t1 / S, s = t1 - qc * S, q = qc / C, c = qc - q * C;
          float v = 0.f;
          for (int part = 0; part < hsplit; ++part) v += s_hp[part * n_headw + t1];
          accum(k.o_head[q] + c * S + s, v);
        }
        for (int hd = 0; hd < k.n_aux; ++hd) {
          const slode_aux ax = k.aux[hd];
          for (int e = t1; e < k.U * ax.z_dim; e += n1) {
            const int j = e / ax.z_dim, l = e - j * ax.z_dim;
            accum(k.o_aux_w1[hd] + e, s_auxd[hd * 32 + j] * s_z[ax.z_off + l]);
          }
          for (int e = t1; e < ax.u_dim * k.U; e += n1) {
            const int q = e / k.U, j = e - q * k.U;
            accum(k.o_aux_w2[hd] + e, s_auxgo[hd * 12 + q] * s_auxh[hd * 32 + j]);
          }
          if (t1 < k.U) accum(k.o_aux_b1[hd] + t1, s_auxd[hd * 32 + t1]);
          if (t1 < ax.u_dim) accum(k.o_aux_b2[hd] + t1, s_auxgo[hd * 12 + t1]);
          if (t1 == 0 && ax.kind == SLODE_AUX_EXPEXP) accum(k.o_aux_c[hd], s_auxgo[hd * 12 + 8]);
        }
        if (k.loc != nullptr && t1 < L) {  // prior nets: latent dim t1 owns its bias entries and weight rows (setup's lookup table)
          const int4 m0 = reinterpret_cast<const int4*>(s_meta)[2 * t1], m1 = reinterpret_cast<const int4*>(s_meta)[2 * t1 + 1];
          if (m0.x) {
            const float gpl = s_gpl[t1], gls = s_gls[t1];
            accum(m0.y, gpl);
            accum(m0.z, gls);
            for (int q = 0; q < m1.z; ++q) {
              const float uv = s_uu[m1.y + q];
              accum(m0.w + q, gpl * uv);
              accum(m1.x + q, gls * uv);
            }
          }
        }
        if (k.stage_encw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed
      }
      STAMP(20);
    }
    BAR();
    STAMP(21);
    if (BWD && k.g_pre != nullptr && tid >= 64 && tid < 64 + k.Hc) {
      // encoder heads + tanh, backward (models/encoder_conv.py:48-51): thread <-> hidden unit.  s_gzl / s_gpl[L..] are next
      // written in P0a and s_encw (= the stage buffer) in P1, both behind the barrier at the top of the loop
      const int mm = tid - 64, Hc = k.Hc;
      const float hv = ENCF ? s_ehid[mm] : k.enc_hid[(long long)b * Hc + mm];
      float g0 = 0.f, g1 = 0.f;
      if (k.stage_encw) {
#pragma unroll 5
        for (int l = 0; l < L; ++l) {
          g0 = fmaf(s_encw[l * Hc + mm], s_gzl[l], g0);
          g1 = fmaf(s_encw[L * Hc + L + l * Hc + mm], s_gpl[L + l], g1);
        }
      } else {
#pragma unroll 5
        for (int l = 0; l < L; ++l) {
          g0 = fmaf(k.enc_zloc_w[l * Hc + mm], s_gzl[l], g0);
          g1 = fmaf(k.enc_zls_w[l * Hc + mm], s_gpl[L + l], g1);
        }
      }
      k.g_pre[(long long)b * 64 + mm] = (g0 + g1) * (1.f - hv * hv);
    }
    if (!ONE && b + vgrid < k.B && tid >= 64 && tid < 128) {   // next trajectory's latent inputs (P0a is long past)
      const int bn = b + vgrid, t1 = tid - 64;
      if (t1 < L) {
        if (k.loc != nullptr) {
          const float a0 = k.loc[(long long)bn * L + t1], a1 = k.scale[(long long)bn * L + t1], a2 = slode_eps_at(k.rng, k.eps, bn, L, t1);
          s_pf[t1] = a0; s_pf[pad4(L) + t1] = a1; s_pf[2 * pad4(L) + t1] = a2;
        } else {
          s_pf[t1] = k.z_in[(long long)bn * L + t1];
        }
      }
      if (k.u != nullptr && t1 < k.nu) s_uu[t1] = slode_label_at(k.lab, k.u, k.nu, bn, t1);
    }
    STAMP(10);
    if (ONE) break;
  }  // trajectories

  // ---- workgroup epilogue: loss partial, then the LDS gradient segment leaves as one slab ------------------------------
  float* slab = k.slabs + (long long)vblk * k.slab_stride;
  if (ts_bad) loss_acc = __builtin_nanf("");
  const float lw = wave_sum(loss_acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = lw;
  BAR();
  STAMP(22);
  if (tid == 0) {
    float loss = 0.f;
    for (int w = 0; w < (NT >> 6); ++w) loss += s_red[w];   // fixed order
    if (k.slabs) slab[0] = loss;
  }
  if (BWD) {
    if (!k.with_ll) {  // pure solve backward: the likelihood-only entries of the segment carry no gradient
      for (int i = tid; i < C * T; i += NT) slab[1 + k.o_cstd + i] = 0.f;
    }
  }
  STAMP(11);
}

template <int S, int H, bool BWD, int T_, int C_, int L_, int Q_, int M_, bool RA, bool ONE, int ALG, int PK = 1, bool ENCF = false, bool SOFTB = false>
hipError_t launch_one(const OdeK& k, int grid, int nthreads, size_t lds, hipStream_t stream) {
  auto fn = ode_elbo_kernel<S, H, BWD, T_, C_, L_, Q_, M_, RA, ONE, ALG, PK, ENCF, SOFTB>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  SLODE_LAUNCH("ode_elbo", fn, dim3(grid), dim3(nthreads), lds, stream, k.stage_t, k.pseg, k.loc ? k.loc : k.z_in, k.loc ? k.scale : nullptr,
               k.eps, k.u, k.sigtab ? k.sigtab : k.cstd, k);
  return hipGetLastError();
}

// one = loop-free form (every trajectory has its own workgroup); ra = reference_adjoint backward
template <int S, int H, int T_ = 0, int C_ = 0, int L_ = 0, int Q_ = 0, int M_ = -1>
hipError_t launch_sh(const OdeK& k, int grid, int nthreads, size_t lds, bool bwd, bool one, bool ra, hipStream_t stream) {
  if (!bwd) return launch_one<S, H, false, T_, C_, L_, Q_, M_, false, false, 0>(k, grid, nthreads, lds, stream);
  if (one) {
    if (ra) return launch_one<S, H, true, T_, C_, L_, Q_, M_, true, true, 0>(k, grid, nthreads, lds, stream);
    return launch_one<S, H, true, T_, C_, L_, Q_, M_, false, true, 0>(k, grid, nthreads, lds, stream);
  }
  if (ra) return launch_one<S, H, true, T_, C_, L_, Q_, M_, true, false, 0>(k, grid, nthreads, lds, stream);
  return launch_one<S, H, true, T_, C_, L_, Q_, M_, false, false, 0>(k, grid, nthreads, lds, stream);
}

}  // namespace

// shapes whose loop-free backward kernel has an instantiation that runs the encoder forward itself (ENCF): the metric shape
bool slode_ode_can_fuse_encoder(const slode_shape& s, bool bwd, int grid) {
  const int Q = s.likelihood == SLODE_GAUSS ? 1 : 3;
  return bwd && grid == s.B && s.H == 25 && s.S == 5 && s.T == 200 && s.C == 3 && s.L == 8 && Q == 3 && s.method == SLODE_RK4 && s.Hc <= 52;
}

int slode_ode_threads(const slode_shape& s) {
  return ode_threads_for(s.T, s.likelihood == SLODE_GAUSS ? 1 : 3, s.C, s.S);
}

static int n_stage(const slode_shape& s) {
  const int R = s.method == SLODE_EULER ? 1 : (s.method == SLODE_MIDPOINT ? 2 : 3);
  return R * (s.T - 1) + 1;
}

// the shape-specialised instantiations with a long latent keep only the hot parameter ranges in LDS (ode_cold_global): BASELINE config[2]
static bool static_cold(const slode_shape& s, bool force_generic) {
  const int Q = s.likelihood == SLODE_GAUSS ? 1 : 3;
  return !force_generic && s.H == 25 && s.S == 8 && s.T == 100 && s.C == 4 && s.L == 50 && Q == 3 &&
         (s.method == SLODE_RK4 || s.method == SLODE_EULER) && ode_cold_global(s.L);
}
static int lds_npar(const slode_shape& s, const slode_layout& lay, bool force_generic, bool ext) {   // (occupancy estimates only)
  const bool st = static_cold(s, force_generic) && (ext ? s.method == SLODE_EULER : s.method == SLODE_RK4);
  return st ? ode_hot_floats(s.S, s.H) : lay.cstd - lay.ode_begin;
}

size_t slode_ode_lds_bytes(const slode_shape& s, int nthreads, bool one) {
  slode_layout lay;
  slode_layout_init(&s, &lay);
  const int Q = s.likelihood == SLODE_GAUSS ? 1 : 3;
  const LdsMap m = lds_map(s.T, s.S, s.H, s.C, s.L, Q, n_stage(s), lds_npar(s, lay, false, false), nthreads, s.aux_in_main ? s.n_aux : 0, one);
  return (size_t)m.total * sizeof(float);
}

hipError_t slode_launch_ode(const OdeLaunch& a, hipStream_t stream, char* err, size_t errlen) {
  const slode_shape& s = a.s;
  const slode_layout& lay = a.lay;
  OdeK k;
  k.B = s.B; k.T = s.T; k.C = s.C; k.L = s.L; k.nu = s.n_u; k.ng = s.n_groups; k.method = s.method;
  k.R = s.method == SLODE_EULER ? 1 : (s.method == SLODE_MIDPOINT ? 2 : 3);
  k.nt = n_stage(s);
  k.gauss = s.likelihood == SLODE_GAUSS;
  k.Q = k.gauss ? 1 : 3;
  k.uses_next = s.method == SLODE_RK4;
  k.tau[0] = 0.5f; k.tau[1] = 0.5f + s.quantile_diff; k.tau[2] = 0.5f - s.quantile_diff;
  const float* p = a.params;
  const int ob = lay.ode_begin;
  for (int g = 0; g < SLODE_MAX_GROUPS; ++g) {
    k.grp[g] = s.groups[g];
    const bool on = g < s.n_groups;
    k.ploc_w[g] = on ? p + lay.ploc_w[g] : nullptr; k.ploc_b[g] = on ? p + lay.ploc_b[g] : nullptr;
    k.pls_w[g] = on ? p + lay.pls_w[g] : nullptr;   k.pls_b[g] = on ? p + lay.pls_b[g] : nullptr;
    k.o_ploc_w[g] = lay.ploc_w[g] - ob; k.o_ploc_b[g] = lay.ploc_b[g] - ob;
    k.o_pls_w[g] = lay.pls_w[g] - ob;   k.o_pls_b[g] = lay.pls_b[g] - ob;
  }
  k.w1 = p + lay.init_w1; k.b1 = p + lay.init_b1; k.w2 = p + lay.init_w2; k.b2 = p + lay.init_b2;
  k.wh = p + lay.dyn_wh; k.bh = p + lay.dyn_bh; k.wg = p + lay.dyn_wg; k.bg = p + lay.dyn_bg;
  k.wd = p + lay.dyn_wd; k.bd = p + lay.dyn_bd;
  for (int q = 0; q < SLODE_MAX_HEADS; ++q) { k.head[q] = p + lay.head_w[q]; k.o_head[q] = lay.head_w[q] - ob; }
  k.cstd = p + lay.cstd;
  k.sigtab = a.sigtab;
  k.o_w1 = lay.init_w1 - ob; k.o_b1 = lay.init_b1 - ob; k.o_w2 = lay.init_w2 - ob; k.o_b2 = lay.init_b2 - ob;
  k.o_wh = lay.dyn_wh - ob; k.o_bh = lay.dyn_bh - ob; k.o_wg = lay.dyn_wg - ob; k.o_bg = lay.dyn_bg - ob;
  k.o_wd = lay.dyn_wd - ob; k.o_bd = lay.dyn_bd - ob; k.o_cstd = lay.cstd - ob;
  k.nseg = lay.ode_end - lay.ode_begin;
  k.npar = lay.cstd - lay.ode_begin;
  k.n_aux = (a.with_ll && s.aux_in_main) ? s.n_aux : 0; k.U = s.U; k.aux_mult = s.aux_mult;
  k.n_aux_lds = s.aux_in_main ? s.n_aux : 0;
  for (int q = 0; q < SLODE_MAX_AUX; ++q) {
    k.aux[q] = s.aux[q];
    k.o_aux_w1[q] = lay.aux_w1[q] - ob; k.o_aux_b1[q] = lay.aux_b1[q] - ob; k.o_aux_w2[q] = lay.aux_w2[q] - ob;
    k.o_aux_b2[q] = lay.aux_b2[q] - ob; k.o_aux_c[q] = lay.aux_c[q] - ob;
  }
  k.pseg = p + lay.ode_begin;
  k.times = a.times; k.stage_t = a.stage_t; k.obs = a.obs; k.u = a.u; k.eps = a.eps; k.loc = a.loc; k.scale = a.scale;
  k.z_in = a.z_in; k.gx_in = a.gx_in; k.sb = a.sb; k.sc = a.sc; k.st = a.st;
  k.x_out = a.x_out; k.z_out = a.z_out; k.g_loc = a.g_loc; k.g_scale = a.g_scale; k.slabs = a.slabs;
  k.slab_stride = a.slab_stride; k.backward = a.backward; k.with_ll = a.with_ll;
  k.enc_hid = a.enc_hid; k.g_pre = a.g_pre; k.glat = a.glat; k.Hc = s.Hc;
  k.x_ext = a.x_ext; k.gx_out = a.gx_out; k.ext_skip = (a.x_ext && a.ext_skip) ? 1 : 0;
  k.enc_zloc_w = p + lay.zloc_w; k.enc_zls_w = p + lay.zls_w; k.enc_zloc_b = p + lay.zloc_b; k.enc_zls_b = p + lay.zls_b;
  k.enc_weff = a.enc_weff; k.enc_beff = a.enc_beff; k.enc_hid_out = a.enc_hid_out;
  k.rng = a.rng; k.lab = a.lab;
  if (k.lab.n > 0 && !k.u) k.u = k.lab.p[0];   // (non-null = "this launch has labels"; the reads go through the accessor)
  if (k.rng.on && !k.eps) k.eps = k.loc;       // (never dereferenced: a valid address for the preloaded argument)

  const int nthreads = slode_ode_threads(s);
  const bool bwd = a.backward != 0;
  {
    // which elements of the segment [ode_begin, cstd) a backward launch writes (owner threads in P6 (C) / P7); the rest carry no
    // gradient in this launch and are written as zeros: e.g. the label heads when the main loss does not score them, the unused
    // second Exp head of a Laplace label head, the decoder heads and prior nets in a pure solve backward
    std::vector<char> cov((size_t)k.npar, 0);
    auto mark = [&](int off, int n) { for (int i = 0; i < n; ++i) cov[(size_t)(off - ob) + i] = 1; };
    if (a.loc != nullptr)
      for (int g = 0; g < s.n_groups; ++g) {
        const slode_group& gr = s.groups[g];
        mark(lay.ploc_w[g], gr.z_dim * gr.u_dim); mark(lay.ploc_b[g], gr.z_dim);
        mark(lay.pls_w[g], gr.z_dim * gr.u_dim);  mark(lay.pls_b[g], gr.z_dim);
      }
    mark(lay.init_w1, s.H * s.L); mark(lay.init_b1, s.H); mark(lay.init_w2, s.S * s.H); mark(lay.init_b2, s.S);
    mark(lay.dyn_wh, s.H * (1 + s.L)); mark(lay.dyn_bh, s.H);
    mark(lay.dyn_wg, s.S * s.H); mark(lay.dyn_bg, s.S); mark(lay.dyn_wd, s.S * s.H); mark(lay.dyn_bd, s.S);
    if (a.with_ll)
      for (int q = 0; q < k.Q; ++q) mark(lay.head_w[q], s.C * s.S);
    for (int q = 0; q < k.n_aux; ++q) {
      const slode_aux& x = s.aux[q];
      mark(lay.aux_w1[q], s.U * x.z_dim); mark(lay.aux_b1[q], s.U); mark(lay.aux_w2[q], x.u_dim * s.U); mark(lay.aux_b2[q], x.u_dim);
      if (x.kind == SLODE_AUX_EXPEXP) mark(lay.aux_c[q], 1);
    }
    k.nz = 0;
    for (int i = 0; i < k.npar;) {
      if (cov[(size_t)i]) { ++i; continue; }
      int j = i;
      while (j < k.npar && !cov[(size_t)j]) ++j;
      if (k.nz == 8) { snprintf(err, errlen, "ode kernel: more than 8 unowned gradient ranges (layout not supported)"); return hipErrorInvalidValue; }
      k.zlo[k.nz] = i; k.zhi[k.nz] = j; ++k.nz;
      i = j;
    }
  }
  // loop-free form when every trajectory has its own workgroup (the grid policy is the caller's: ode_grid_for)
  const bool one = bwd && a.grid == s.B && !a.force_loop;
  const bool ra = bwd && s.grad_mode == SLODE_GRAD_REFERENCE_ADJOINT;
  // which launches take a long-latent shape-specialised instantiation (cold parameters in global memory, ode_cold_global) -- decided
  // here, ONCE, for the LDS size and for the dispatch below
  const bool shape_c2 = s.H == 25 && s.S == 8 && s.T == 100 && s.C == 4 && s.L == 50 && k.Q == 3;
  bool aux16 = true;   // the scorer's cold-parameter path keeps a label head's weight row in 16 registers
  for (int q = 0; q < s.n_aux; ++q) aux16 = aux16 && s.aux[q].z_dim <= 16;
  const bool static_scorer = a.x_ext && !a.force_generic && a.alg == 0 && !ra && shape_c2 && aux16 && s.method == SLODE_EULER && (!bwd || one);
  bool static_c2 = !a.x_ext && !a.force_generic && a.alg == 0 && shape_c2 && s.method == SLODE_RK4;
  if (static_c2) {
    // its cold parameter ranges [priors | W_1], W_z and the label heads are staged in the idle work block A | x | lam | st during P0
    // (COLDB): they must fit (they do for the reference's proc config: 4713 of 4768 floats); another split of the same dims may not
    const LdsMap mc = lds_map(s.T, s.S, s.H, s.C, s.L, k.Q, k.nt, ode_hot_floats(s.S, s.H), nthreads, k.n_aux_lds, one);
    const int cold = k.o_b1 + s.H * (1 + s.L) + (k.n_aux > 0 ? k.npar - k.o_aux_w1[0] : 0);
    if (cold > 3 * mc.ax + mc.stn || k.o_b1 < 0) static_c2 = false;
  }
  const int npar_lds = (static_scorer || static_c2) ? ode_hot_floats(s.S, s.H) : k.npar;
  const size_t lds = sizeof(float) * (size_t)lds_map(s.T, s.S, s.H, s.C, s.L, k.Q, k.nt, npar_lds, nthreads, k.n_aux_lds, one).total;
  {
    const LdsMap m = lds_map(s.T, s.S, s.H, s.C, s.L, k.Q, k.nt, npar_lds, nthreads, k.n_aux_lds, one);
    // z_loc.weight | z_loc.bias | z_scale.0.weight must be one block of the flat vector and fit the stage buffer
    k.stage_encw = (k.g_pre != nullptr && 2 * s.L * s.Hc + s.L <= m.stn && lay.zloc_b == lay.zloc_w + s.L * s.Hc &&
                    lay.zls_w == lay.zloc_b + s.L) ? 1 : 0;
  }
  if (lds > 160 * 1024) {
    snprintf(err, errlen, "ode kernel needs %zu B of LDS (> 160 KiB): T=%d S=%d too large", lds, s.T, s.S);
    return hipErrorInvalidValue;
  }
  {
    const int cap = ode_max_threads(s.S, 0, 0, 0, one, bwd);   // the generic instantiation's declared block size
    if (nthreads > cap) {
      snprintf(err, errlen, "ode kernel: ode_state_dim %d with %s supports at most %d time points (got T=%d)", s.S,
               (one || !bwd) ? "one workgroup per trajectory" : "the persistent-loop grid (B > 65536)", cap, s.T);
      return hipErrorInvalidValue;
    }
  }
  // measured A/B arms: direct evaluation of the dynamics heads, MFMA contraction -- metric shape, exact gradients, loop-free
  if (a.alg != 0 && bwd) {
    if (!(s.H == 25 && s.S == 5 && s.T == 200 && s.C == 3 && s.L == 8 && k.Q == 3 && s.method == SLODE_RK4 && one && !ra && !a.x_ext)) {
      snprintf(err, errlen, "ode kernel variant %d is instantiated for the metric shape only (cvs T=200 L=8 rk4, exact gradients, B <= 65536)", a.alg);
      return hipErrorInvalidValue;
    }
    if (a.alg == 1) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 1>(k, a.grid, nthreads, lds, stream);
    if (a.alg == 2) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 2>(k, a.grid, nthreads, lds, stream);
    snprintf(err, errlen, "unknown ode kernel variant %d", a.alg);
    return hipErrorInvalidValue;
  }
  // fused encoder forward (metric shape, loop-free, folded encoder path): the caller skipped the enc_fwd2 launch
  if (a.enc_fuse) {
    if (!slode_ode_can_fuse_encoder(s, bwd, a.grid) || a.alg != 0 || a.x_ext || a.force_generic || a.force_loop || (a.pack && a.pack < 10)) {
      snprintf(err, errlen, "ode kernel: the fused encoder forward has no instantiation for this launch");
      return hipErrorInvalidValue;
    }
    // packed forms with the shared encoder pass and per-trajectory barriers (SOFTB): pack = 12 / 14 -> 2 / 4 trajectories per workgroup
    const int pk = a.pack == 14 ? 4 : (a.pack == 12 ? 2 : 1);
    const size_t lds_pk = (size_t)pk * lds + (size_t)pk * 256 + 64;
    if (pk > 1 && s.B % pk == 0 && lds_pk <= 160 * 1024) {
      if (pk == 4) {
        if (ra) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, true, true, 0, 4, true, true>(k, a.grid / 4, nthreads * 4, lds_pk, stream);
        return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 0, 4, true, true>(k, a.grid / 4, nthreads * 4, lds_pk, stream);
      }
      if (ra) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, true, true, 0, 2, true, true>(k, a.grid / 2, nthreads * 2, lds_pk, stream);
      return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 0, 2, true, true>(k, a.grid / 2, nthreads * 2, lds_pk, stream);
    }
    if (ra) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, true, true, 0, 1, true>(k, a.grid, nthreads, lds + 256, stream);
    return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 0, 1, true>(k, a.grid, nthreads, lds + 256, stream);
  }
  // packed form: four trajectories per workgroup (metric shape, loop-free, batch a multiple of four)
  if (a.pack == 4 && one && a.alg == 0 && !a.x_ext && !a.force_generic && s.B % 4 == 0 && 4 * lds <= 160 * 1024 &&
      s.H == 25 && s.S == 5 && s.T == 200 && s.C == 3 && s.L == 8 && k.Q == 3 && s.method == SLODE_RK4) {
    if (ra) return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, true, true, 0, 4>(k, a.grid / 4, nthreads * 4, 4 * lds, stream);
    return launch_one<5, 25, true, 200, 3, 8, 3, SLODE_RK4, false, true, 0, 4>(k, a.grid / 4, nthreads * 4, 4 * lds, stream);
  }
  // the scorer of BASELINE config[2] as written (proc, dopri5): shape-specialised, solver phases compiled out
  if (static_scorer) {
    const size_t lds_sc = sizeof(float) * (size_t)lds_map(s.T, s.S, s.H, s.C, s.L, k.Q, k.nt, npar_lds, nthreads, k.n_aux_lds, one, true).total;
    if (!bwd) return launch_one<8, 25, false, 100, 4, 50, 3, SLODE_EULER, false, false, 3>(k, a.grid, nthreads, lds_sc, stream);
    if (one) return launch_one<8, 25, true, 100, 4, 50, 3, SLODE_EULER, false, true, 3>(k, a.grid, nthreads, lds_sc, stream);
  }
  // shape-specialised instantiations (compile-time LDS offsets, loop bounds, solver): the BASELINE.json shapes and the reference default
  if (s.H == 25 && !a.x_ext && !a.force_generic) {
#define SLODE_STATIC(SS, TT, CC, LL, QQ, MM)                                                              \
    if (s.S == SS && s.T == TT && s.C == CC && s.L == LL && k.Q == QQ && s.method == MM)                  \
      return launch_sh<SS, 25, TT, CC, LL, QQ, MM>(k, a.grid, nthreads, lds, bwd, one, ra, stream)
    SLODE_STATIC(5, 200, 3, 8, 3, SLODE_RK4);        // configs [1] / [3]: cvs, latent 3+3+2, ALD
    SLODE_STATIC(5, 100, 3, 4, 3, SLODE_RK4);        // config [0]: cvs, latent 1+1+2
    if (static_c2) return launch_sh<8, 25, 100, 4, 50, 3, SLODE_RK4>(k, a.grid, nthreads, lds, bwd, one, ra, stream);   // config [2] shapes: proc, fixed grid
    SLODE_STATIC(5, 300, 4, 15, 1, SLODE_RK4);       // config [4]: challenge, Gauss
    SLODE_STATIC(5, 86, 3, 15, 3, SLODE_MIDPOINT);   // reference default: training_cvs.py, config_cvs.py
#undef SLODE_STATIC
  }
  if (s.H == 25 && s.S == 5) return launch_sh<5, 25>(k, a.grid, nthreads, lds, bwd, one, ra, stream);
  if (s.H == 25 && s.S == 8) return launch_sh<8, 25>(k, a.grid, nthreads, lds, bwd, one, ra, stream);
  snprintf(err, errlen, "ode kernel is instantiated for (ode_state_dim, ode_hidden_dim) in {(5,25),(8,25)}; got (%d,%d)", s.S, s.H);
  return hipErrorInvalidValue;
}
