// Adaptive Dormand-Prince 5(4) latent-ODE solve and its reverse sweep for gfx950: the `solver="dopri5"` string the reference can pass
// through to torchdiffeq.odeint (models/blackbox_ode.py:41-45; BASELINE config[2]).
//
// torchdiffeq's controller uses ONE step size for the whole [B,S] tensor (its error norm is an RMS over the batch), which
// makes results depend on batch composition and cannot shard.  Contract here (SURVEY hard part 3): one controller PER
// TRAJECTORY, running torchdiffeq's algorithm on its own state (FSAL, Hairer initial step, ratio = rms(err / (atol +
// rtol*max(|y0|,|y1|))), factor clamp [0.2, 10] with safety 0.9, quartic dense output through (y0, y_mid, y1, f0, f1)) --
// validated at solution level against the oracle's per-trajectory restatement and scipy RK45.
//
// Mapping: EIGHT LANES PER TRAJECTORY (8 trajectories per wave), lane g = state component g.  An adaptive solve is one long
// dependent chain per trajectory, so its latency -- not the FLOPs -- sets the kernel time.  The chain is kept short by the structure
// of the dynamics net: its hidden layer is relu(w_t t + u_j), so the head pre-activations are piecewise linear in t and a lane
// reads value and slope of its two heads from a per-trajectory table of the H + 1 linear segments (load_units / eval_ad): one
// evaluation of the coefficients is 4 compares per lane, an OR butterfly over the group on the DPP crossbar, a popcount, one LDS row,
// 2 fmas and one sigmoid pair -- branch-free, the evaluations of a step independent of each other; all the Runge-Kutta arithmetic is
// scalar per lane.
//
// Training with dopri5 (BASELINE config[2]): the forward kernel also records every accepted step (t, dt, y) and
// `dopri5_bwd_kernel` walks a trajectory's record backwards -- the exact reverse mode of the accepted Dormand-Prince steps and of
// the dense-output polynomial, step sizes held fixed (the controller is not differentiated) -- in the same 8-lane mapping.  The
// right-hand side is linear in the state with coefficients that depend on time only, so a step's seven stages are re-evaluated from
// its recorded (t, dt, y) instead of being stored, and the reverse mode is component-wise: one scalar of everything per lane.  The
// coefficients come from the same segment table and the weight gradients from running sums parked at each hidden unit's switching
// time (grp::sweep_sample).  A workgroup's 16 trajectories are summed in a fixed order into one slab row in the
// layout of the fixed-grid kernel's slabs (reduced by the same deterministic tail).
#include "slode_common.h"
#include <cstddef>

namespace {

constexpr int G = 8;            // lanes per trajectory
constexpr int DNT = 128;        // threads per workgroup (B = 4096: 256 workgroups, one per CU)
constexpr int TPB = DNT / G;    // trajectories per workgroup
constexpr int JL = 4;           // hidden units per lane (H <= 32)

struct DpK {
  int B, T, L;
  const float *times, *z, *w1, *b1, *w2, *b2, *wh, *bh, *wg, *bg, *wd, *bd;
  const float *loc, *scale, *eps;   // z == nullptr: z = loc + scale * eps (the guide's sample), written to z_out
  float* x;
  float* z_out;
  float* rec;    // [kmax][B][S + 2] accepted steps (t, dt, y) or nullptr
  int* nrec;     // [B] accepted steps per trajectory (> kmax: record overflow; -1: max_steps exhausted)
  int kmax;
  float rtol, atol;
  int max_steps;
  float* tabs;       // [workgroup][TABF] or nullptr: the set-up's tables (GroupLds block + init-net pre-activations) for the reverse sweep
  RngK rng;          // on: eps is drawn here (Philox, slode_common.h) and written to eps_out for the scorer and the reverse sweep
  float* eps_out;
};

// reduce-scatter over the 8 lanes of a trajectory: lane g returns the group's sum of v[g] (halving butterfly, 4 + 2 + 1 shuffles)
__device__ __forceinline__ float group_scatter8(float (&v)[8], int g) {
  {
    const bool hi = (g & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
      v[i] = keep + __shfl_xor(send, 4, 64);
    }
  }
  {
    const bool hi = (g & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float keep = hi ? v[i + 2] : v[i], send = hi ? v[i] : v[i + 2];
      v[i] = keep + __shfl_xor(send, 2, 64);
    }
  }
  const bool hi = (g & 1) != 0;
  const float keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
  return keep + __shfl_xor(send, 1, 64);
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
// butterflies over the 8 lanes of a trajectory on the DPP crossbar (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror): every
// lane of the group ends with the same bits (each step adds the same two operands on both sides)
__device__ __forceinline__ unsigned group_or(unsigned v) {
  v |= dpp_u<0xB1>(v);
  v |= dpp_u<0x4E>(v);
  v |= dpp_u<0x141>(v);
  return v;
}
__device__ __forceinline__ unsigned group_add_u(unsigned v) {
  v += dpp_u<0xB1>(v);
  v += dpp_u<0x4E>(v);
  v += dpp_u<0x141>(v);
  return v;
}
__device__ __forceinline__ float group_add(float v) {
  v += __uint_as_float(dpp_u<0xB1>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u<0x4E>(__float_as_uint(v)));
  v += __uint_as_float(dpp_u<0x141>(__float_as_uint(v)));
  return v;
}

// global -> LDS copies in batches of NB loads per thread, then NB stores (written as `dst[i] = src[i]` in a loop, every LDS store waits
// for its own load: one memory round trip per element and thread)
template <typename T, int NB = 8>
__device__ __forceinline__ void stage_to_lds(T* __restrict__ dst, const T* __restrict__ src, int n, int tid, int nthreads) {
  for (int i0 = tid; i0 < n; i0 += NB * nthreads) {
    T v[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) v[q] = src[min(i0 + q * nthreads, n - 1)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (i0 + q * nthreads < n) dst[i0 + q * nthreads] = v[q];
  }
}

// The hidden layer is relu(w_t t + u_j): unit j switches at th_j = -u_j / w_t,j, so between two consecutive switching times the head
// pre-activations o_c(t) = bias_c + sum_{j on} W_cj (w_t,j t + u_j) are LINEAR in t.  Per trajectory the H + 1 segments are tabulated
// once (load_units): row r = (value of the growth / degradation head of component g at the segment's centre, their slopes), the
// segments ordered by switching time.  The segment of a time t is the NUMBER of switching times <= t -- no search: each lane counts
// its 4 units, an add butterfly over the group on the DPP crossbar sums the counts -- so one evaluation of the coefficients is 4
// compares per lane, 3 DPP adds, one 16-byte LDS read, 2 fmas and a sigmoid pair: branch-free, and the evaluations of one step are
// independent.
// (The switch is placed at the rounded th_j instead of at the exact sign change of fma(w_t, t, u_j): the heads are continuous there,
// the difference is a few ulp of o_c.)
constexpr float BIGT = 3.0e38f;
struct Units {
  float wt[JL], u[JL];   // this lane's hidden units g, g+8, g+16, g+24
  float th[JL];          // their switching times (+BIGT beyond H: never passed)
};
// number of switching times <= t: 4 compares per lane, summed over the group (the same count in the 8 lanes of a trajectory)
__device__ __forceinline__ int count_passed(float t, const Units& w) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < JL; ++i) c += (t >= w.th[i]) ? 1 : 0;
  return (int)group_add_u(static_cast<unsigned>(c));
}
// growth / degradation coefficient of this lane's state component at time t (blackbox_ode.py:97-109); returns the segment index
template <int H>
__device__ __forceinline__ int eval_ad(float t, const Units& w, int g, bool own, const float4* __restrict__ tab,
                                       const float* __restrict__ ctr, float& a, float& d) {
  const int r = count_passed(t, w);
  const float4 row = tab[r * G + g];
  const float dtau = t - ctr[r];
  a = own ? sigmoidf_fast(fmaf(row.z, dtau, row.x)) : 0.f;
  d = own ? sigmoidf_fast(fmaf(row.w, dtau, row.y)) : 0.f;
  return r;
}

// N evaluations at once: all the counts, then all the LDS reads in one batch (left alone, the scheduler waits for each row before it
// issues the next evaluation's reads), then the sigmoids
template <int H, int N>
__device__ __forceinline__ void eval_ad_batch(const float (&t)[N], const Units& w, int g, bool own, const float4* __restrict__ tab,
                                              const float* __restrict__ ctr, float (&a)[N], float (&d)[N], int (&r)[N]) {
#pragma unroll
  for (int n = 0; n < N; ++n) r[n] = count_passed(t[n], w);
  float4 row[N];
  float cc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { row[n] = tab[r[n] * G + g]; cc[n] = ctr[r[n]]; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const float dtau = t[n] - cc[n];
    a[n] = own ? sigmoidf_fast(fmaf(row[n].z, dtau, row[n].x)) : 0.f;
    d[n] = own ? sigmoidf_fast(fmaf(row[n].w, dtau, row[n].y)) : 0.f;
  }
}

// x0 = sigmoid(W2 relu(W1 z + b1) + b2), this lane's component (blackbox_ode.py:19-22)
template <int S, int H>
__device__ __forceinline__ float init_state(const float* w2, const float* b2, const float (&pre0)[JL], int g, bool own) {
  float o[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) o[s] = 0.f;
#pragma unroll
  for (int i = 0; i < JL; ++i) {
    const int j = g + G * i, jj = j < H ? j : 0;
    const float hp = j < H ? fmaxf(pre0[i], 0.f) : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) o[s] = fmaf(w2[s * H + jj], hp, o[s]);
  }
  const float v = group_scatter8(o, g);
  return own ? sigmoidf_fast(v + b2[g]) : 0.f;
}

// the same with the output layer's weights requested at the top of the kernel (w2r[i][s] = w2[s][unit g + 8 i], b2r = b2[g])
template <int S, int H>
struct InitRegs { float w2[JL][8], b2; };
template <int S, int H>
__device__ __forceinline__ void init_request(const float* w2, const float* b2, int g, InitRegs<S, H>& r) {
#pragma unroll
  for (int i = 0; i < JL; ++i) {
    const int jj = (g + G * i) < H ? g + G * i : 0;
#pragma unroll
    for (int q = 0; q < S; ++q) r.w2[i][q] = w2[q * H + jj];
  }
  r.b2 = b2[g < S ? g : 0];
}
template <int S, int H>
__device__ __forceinline__ float init_state(const InitRegs<S, H>& r, const float (&pre0)[JL], int g, bool own) {
  float o[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = 0.f;
#pragma unroll
  for (int i = 0; i < JL; ++i) {
    const float hp = (g + G * i) < H ? fmaxf(pre0[i], 0.f) : 0.f;
#pragma unroll
    for (int q = 0; q < S; ++q) o[q] = fmaf(r.w2[i][q], hp, o[q]);
  }
  const float v = group_scatter8(o, g);
  return own ? sigmoidf_fast(v + r.b2) : 0.f;
}

// LDS of one trajectory group: the shared tables (time weights, head weights by unit) and per trajectory the hidden offsets, the
// switching times, their order, the segment centres and the segment table
template <int H>
struct GroupLds {
  float *wt, *wgd, *u, *th, *ctr;
  int *ord, *rnk;   // unit of a rank, rank of a unit
  float4* tab;
  static constexpr int floats(int ntraj) { return 32 + H * 16 + ntraj * 32 * 5 + ntraj * (H + 1) * G * 4; }
  __device__ __forceinline__ float* carve(float* base, int ntraj) {   // base 16-byte aligned; returns the first float behind
    tab = reinterpret_cast<float4*>(base);
    wt = base + ntraj * (H + 1) * G * 4;
    wgd = wt + 32;
    u = wgd + H * 16;
    th = u + ntraj * 32;
    ctr = th + ntraj * 32;
    ord = reinterpret_cast<int*>(ctr + ntraj * 32);
    rnk = ord + ntraj * 32;
    return ctr + ntraj * 96;
  }
};

// LDS-DMA of `nrows` rows of a row-major matrix (row pitch `pitch` floats, `L` columns starting at src) into dst[nrows][LP] -- the pad
// columns repeat the row's last element and are never used.  One 4-byte element per lane: the LDS destination of a wave instruction is
// contiguous, the global source of each lane is free.  wave / nwaves: wave-uniform.
__device__ __forceinline__ void dma_rows(float* dst, const float* src, int nrows, int L, int pitch, int LP, int wave, int nwaves, int lane) {
  const int n = nrows * LP, step = nwaves * 64;
  // (row, column) of this lane's element, advanced by `step` elements per instruction without dividing again (an integer division per
  // DMA instruction was 7 k cycles of the set-up)
  const int sq = step / LP, sr = step - sq * LP;   // (uniform)
  int j = (wave * 64 + lane) / LP, l = (wave * 64 + lane) - j * LP;
  for (int b0 = wave * 64; b0 < n; b0 += step) {
    if (b0 + lane < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)j * pitch + min(l, L - 1)),
                                       (__attribute__((address_space(3))) void*)(dst + b0), 4, 0, 0);
    j += sq; l += sr;
    if (l >= LP) { l -= LP; ++j; }
  }
}
__device__ __forceinline__ void dma_flat(float* dst, const float* src, int n, int wave, int nwaves, int lane) {
  for (int b0 = wave * 64; b0 < n; b0 += nwaves * 64)
    if (b0 + lane < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b0 + lane),
                                       (__attribute__((address_space(3))) void*)(dst + b0), 4, 0, 0);
}

// Set-up of the adaptive kernels, part 1 (the kernel's FIRST instructions): every global operand of the set-up is requested at once -- the
// z-columns of the dynamics net's hidden layer and the init net's first layer straight into LDS (LDS-DMA, rows padded to LP = 4 ceil(L / 4)
// floats so that the unit sums read them 16 bytes at a time; they borrow the table region, which is written only after their last read),
// the output times likewise, the small per-unit operands into registers.  One memory round trip instead of eight (cycle stamps in DESIGN
// 3.3: 17 k of the set-up's 43 k cycles were spent waiting for one staged copy after the other).
template <int H>
struct UnitRegs {
  float wt, wgd[4], bh[JL], b1[JL];   // w_t of unit tid (tid < 32); this thread's elements of the by-unit head table; biases of this lane's units
};
template <int S, int H>
__device__ __forceinline__ void units_request(const float* wh, const float* bh, const float* wg, const float* wd, const float* w1, const float* b1,
                                              const float* times, int T, float* s_times, int L, int tid, int nthreads, const GroupLds<H>& m,
                                              UnitRegs<H>& ur) {
  const int LP = (L + 3) & ~3, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6, lane = tid & 63, g = tid & (G - 1);
  float* s_whz = reinterpret_cast<float*>(m.tab);   // [H][LP]  z-columns of the hidden layer
  float* s_w1m = s_whz + H * LP;                    // [H][LP]
  dma_rows(s_whz, wh + 1, H, L, 1 + L, LP, wave, nwaves, lane);
  dma_rows(s_w1m, w1, H, L, L, LP, wave, nwaves, lane);
  dma_flat(s_times, times, T, wave, nwaves, lane);
  ur.wt = wh[(tid < H ? tid : 0) * (1 + L)];
#pragma unroll
  for (int q = 0; q < 4; ++q) {   // (H * 16 <= 4 x nthreads: 400 elements, 128 threads or more)
    const int i = min(tid + q * nthreads, H * 16 - 1), j = i >> 4, c = i & 15, gg = min(c & 7, S - 1);
    ur.wgd[q] = c < 8 ? wg[gg * H + j] : wd[gg * H + j];
  }
#pragma unroll
  for (int i = 0; i < JL; ++i) {
    const int jj = (g + G * i) < H ? g + G * i : 0;
    ur.bh[i] = bh[jj]; ur.b1[i] = b1[jj];
  }
}

// Part 2: fills the shared tables, this lane's hidden units (offsets u = W_z z + b_h of the dynamics net -> registers + LDS,
// pre-activations of the init net -> pre0) and the trajectory's segment table.  The caller has written the latent rows s_z[slot][LP]
// (pad = 0) and waits for nothing: the DMA of part 1 is drained here.  The integration range [tlo, thi] (segment centres are clamped to it) is read from the staged output times.  Returns the dir bits.  Contains barriers: every thread of the workgroup calls it.
// LPT: lanes per trajectory -- G (eight trajectories per wave) or 16 / 32 / 64 (the lane groups of a trajectory run the set-up side by
// side and write the same values)
template <int S, int H, int LPT = G>
__device__ __forceinline__ unsigned load_units(const UnitRegs<H>& ur, const float* bg, const float* bd, const float* zrow, int L, int tid,
                                               int nthreads, const float* s_times, int T, const GroupLds<H>& m, Units& w, float (&pre0)[JL]) {
  static_assert(H * 16 <= 4 * 128, "by-unit head table: four elements per thread");
  const int g = tid & (G - 1), slot = tid / LPT, LP = (L + 3) & ~3;
  const bool own = g < S;
  if (tid < 32) m.wt[tid] = tid < H ? ur.wt : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid + q * nthreads;
    if (i < H * 16) m.wgd[i] = ((i & 7) < S) ? ur.wgd[q] : 0.f;
  }
  const float* s_whz = reinterpret_cast<const float*>(m.tab);   // [H][LP]
  const float* s_w1m = s_whz + H * LP;                          // [H][LP]
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's LDS-DMA has landed; the barrier covers the others'
  __syncthreads();
  const float tlo = fminf(s_times[0], s_times[T - 1]), thi = fmaxf(s_times[0], s_times[T - 1]);
  float* s_us = m.u + slot * 32;
  float* s_th = m.th + slot * 32;
  unsigned dm = 0u;
  {
    float uj[JL], pj[JL];
    const float4 *whr[JL], *w1r[JL];
#pragma unroll
    for (int i = 0; i < JL; ++i) {
      const int jj = (g + G * i) < H ? g + G * i : 0;
      uj[i] = ur.bh[i]; pj[i] = ur.b1[i];
      whr[i] = reinterpret_cast<const float4*>(s_whz + jj * LP);
      w1r[i] = reinterpret_cast<const float4*>(s_w1m + jj * LP);
    }
    const float4* z4 = reinterpret_cast<const float4*>(zrow);
    // sixteen bytes per read, the next quad's nine reads issued before this quad's 32 multiply-adds (two register sets in rotation); the
    // sums run over l in order and the pad columns are not added at all
    float4 zv[2], a[2][JL], b[2][JL];
    const int nq = LP >> 2;
    auto fetch = [&](int l4, int set) __attribute__((always_inline)) {
      const int q = min(l4, nq - 1);
      zv[set] = z4[q];
#pragma unroll
      for (int i = 0; i < JL; ++i) { a[set][i] = whr[i][q]; b[set][i] = w1r[i][q]; }
    };
    auto sums = [&](int l4, int set) __attribute__((always_inline)) {
      const int nv = min(4, L - 4 * l4);   // (uniform) elements of this quad inside the row
#pragma unroll
      for (int i = 0; i < JL; ++i) {
        uj[i] = fmaf(a[set][i].x, zv[set].x, uj[i]); pj[i] = fmaf(b[set][i].x, zv[set].x, pj[i]);
        if (nv > 1) { uj[i] = fmaf(a[set][i].y, zv[set].y, uj[i]); pj[i] = fmaf(b[set][i].y, zv[set].y, pj[i]); }
        if (nv > 2) { uj[i] = fmaf(a[set][i].z, zv[set].z, uj[i]); pj[i] = fmaf(b[set][i].z, zv[set].z, pj[i]); }
        if (nv > 3) { uj[i] = fmaf(a[set][i].w, zv[set].w, uj[i]); pj[i] = fmaf(b[set][i].w, zv[set].w, pj[i]); }
      }
    };
    fetch(0, 0);
    for (int l4 = 0; l4 < nq; l4 += 2) {
      fetch(l4 + 1, 1);
      sums(l4, 0);
      if (l4 + 1 < nq) {   // (uniform)
        fetch(l4 + 2, 0);
        sums(l4 + 1, 1);
      }
    }
#pragma unroll
    for (int i = 0; i < JL; ++i) {
      const int j = g + G * i;
      const bool valid = j < H;
      const float wt = valid ? m.wt[j] : 0.f;
      w.wt[i] = wt;
      w.u[i] = valid ? uj[i] : 0.f;
      pre0[i] = valid ? pj[i] : 0.f;
      // switching time; w_t == 0: the predicate is the sign of u (always / never on)
      float th = wt != 0.f ? -uj[i] / wt : (uj[i] > 0.f ? -BIGT : BIGT);
      th = fminf(fmaxf(th, -BIGT), BIGT);
      if (!valid) th = BIGT;
      w.th[i] = th;
      dm |= (wt >= 0.f || !valid) ? (1u << j) : 0u;
      s_us[j] = w.u[i];
      s_th[j] = th;
    }
  }
  const unsigned dirmask = group_or(dm);
  __syncthreads();
  // order of the switching times: rank by counting (ties by unit index); the 32 switching times come in as eight 16-byte reads
  {
    float thk[32];
    const float4* t4 = reinterpret_cast<const float4*>(s_th);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float4 v = t4[q]; thk[4 * q] = v.x; thk[4 * q + 1] = v.y; thk[4 * q + 2] = v.z; thk[4 * q + 3] = v.w; }
    int rank[JL];
#pragma unroll
    for (int i = 0; i < JL; ++i) rank[i] = 0;
#pragma unroll
    for (int kk = 0; kk < H; ++kk) {
#pragma unroll
      for (int i = 0; i < JL; ++i) rank[i] += (int)((thk[kk] < w.th[i]) | ((thk[kk] == w.th[i]) & (kk < g + G * i)));   // (bitwise: no branches)
    }
#pragma unroll
    for (int i = 0; i < JL; ++i)
      if (g + G * i < H) { m.ord[slot * 32 + rank[i]] = g + G * i; m.rnk[slot * 32 + g + G * i] = rank[i]; }
  }
  __syncthreads();
  // segment table, event by event in the centred form: V(c') = V(c) + slope (c' - c) +- W pre(c'), slope +- W w_t
  {
    float4* tab = m.tab + slot * (H + 1) * G;
    float* ctr = m.ctr + slot * 32;
    float c = tlo;
    float va = own ? bg[g] : 0.f, vd = own ? bd[g] : 0.f, ala = 0.f, ald = 0.f;
    constexpr int EB = 5;
    for (int j0 = 0; j0 < H; j0 += EB) {   // below every switching time the units with w_t < 0 are on; operands five units at a time
      float wtv[EB], usv[EB], w1v[EB], w2v[EB];
#pragma unroll
      for (int q = 0; q < EB; ++q) {
        const int j = min(j0 + q, H - 1);
        wtv[q] = m.wt[j]; usv[q] = s_us[j]; w1v[q] = m.wgd[j * 16 + g]; w2v[q] = m.wgd[j * 16 + 8 + g];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < EB; ++q) {
        const int j = j0 + q;
        if (j < H) {   // (uniform)
          const bool on = !((dirmask >> j) & 1u);
          const float pre = fmaf(wtv[q], c, usv[q]);
          const float h = on ? pre : 0.f, hw = on ? wtv[q] : 0.f;
          va = fmaf(w1v[q], h, va); vd = fmaf(w2v[q], h, vd);
          ala = fmaf(w1v[q], hw, ala); ald = fmaf(w2v[q], hw, ald);
        }
      }
    }
    tab[g] = make_float4(va, vd, ala, ald);
    if (g == 0) ctr[0] = c;
    // five events at a time: their units, then the units' five operands, are read in two batches ahead of the chain (one event at a
    // time costs two dependent LDS round trips per event on the trajectory's critical path)
    for (int r0 = 0; r0 < H; r0 += EB) {
      int jv[EB];
      float thv[EB], wtv[EB], usv[EB], w1v[EB], w2v[EB];
#pragma unroll
      for (int q = 0; q < EB; ++q) jv[q] = m.ord[slot * 32 + min(r0 + q, H - 1)];
#pragma unroll
      for (int q = 0; q < EB; ++q) {
        thv[q] = s_th[jv[q]]; wtv[q] = m.wt[jv[q]]; usv[q] = s_us[jv[q]];
        w1v[q] = m.wgd[jv[q] * 16 + g]; w2v[q] = m.wgd[jv[q] * 16 + 8 + g];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < EB; ++q) {
        const int r = r0 + q;
        if (r < H) {   // (uniform)
          const float c1 = fminf(fmaxf(thv[q], tlo), thi);
          const float sg = ((dirmask >> jv[q]) & 1u) ? 1.f : -1.f;
          const float pre = sg * fmaf(wtv[q], c1, usv[q]), swt = sg * wtv[q];
          const float dc = c1 - c;
          va = fmaf(w1v[q], pre, fmaf(ala, dc, va)); vd = fmaf(w2v[q], pre, fmaf(ald, dc, vd));
          ala = fmaf(w1v[q], swt, ala); ald = fmaf(w2v[q], swt, ald);
          c = c1;
          tab[(r + 1) * G + g] = make_float4(va, vd, ala, ald);
          if (g == 0) ctr[r + 1] = c;
        }
      }
    }
  }
  __syncthreads();
  return dirmask;
}

template <int S>
__device__ __forceinline__ float group_rms(float v, bool own) { return sqrtf(group_add(own ? v * v : 0.f) * (1.0f / S)); }
// the controller's copy: v_sqrt_f32 (1 ulp) -- the ratio only steers the step size
#ifdef SLODE_DP5_PRECISE   // diagnostic build (make dp5precise; tools/dp5_accuracy.py): IEEE division / sqrt / pow in the controller, fp64 dense output
#define DP5_RCP(x) (1.0f / (x))
template <int S>
__device__ __forceinline__ float group_rms_fast(float v, bool own) { return sqrtf(group_add(own ? v * v : 0.f) * (1.0f / S)); }
#else
#define DP5_RCP(x) __builtin_amdgcn_rcpf(x)
template <int S>
__device__ __forceinline__ float group_rms_fast(float v, bool own) { return __builtin_amdgcn_sqrtf(group_add(own ? v * v : 0.f) * (1.0f / S)); }
#endif

// The set-up's result -- the workgroup's whole GroupLds block (segment tables, switching times, their order, head tables) and the init
// net's pre-activations -- goes to global memory for the reverse sweep, which rebuilds exactly these tables for exactly these sixteen
// trajectories otherwise (18 k cycles of its set-up): sixteen bytes per lane and store, fire and forget.
template <int H>
constexpr int tab_floats() { return GroupLds<H>::floats(16) + 16 * 32; }
template <int H>
__device__ __forceinline__ void dump_tables(float* tabs, const float* smem, const float (&pre0)[JL], int slot, int g, bool writer, int tid, int nthreads) {
  float* dst = tabs + (size_t)blockIdx.x * tab_floats<H>();
  constexpr int n4 = GroupLds<H>::floats(16) / 4;
  const float4* src = reinterpret_cast<const float4*>(smem);
  for (int i = tid; i < n4; i += nthreads) reinterpret_cast<float4*>(dst)[i] = src[i];
  if (writer) {
#pragma unroll
    for (int i = 0; i < JL; ++i) dst[4 * n4 + slot * 32 + g + G * i] = pre0[i];
  }
}

// z = loc + scale * eps (or the given z) of one trajectory -> LDS row [LP] (and z_out / eps_out), LPT lanes per trajectory.  Two parts: the
// loads are the kernel's FIRST requests (their round trip -- the first touch of three arrays, ~8 k cycles in the stamps -- passes while the
// set-up's other requests are issued), the stores follow those.
template <int LPT>
struct LatentRegs { float lo[(SLODE_MAX_L + LPT - 1) / LPT], sc[(SLODE_MAX_L + LPT - 1) / LPT], ep[(SLODE_MAX_L + LPT - 1) / LPT]; };
template <int LPT>
__device__ __forceinline__ void latent_request(const DpK& k, long long bb, int lg, int L, LatentRegs<LPT>& r) {
  constexpr int ZQ = (SLODE_MAX_L + LPT - 1) / LPT;
#pragma unroll
  for (int q = 0; q < ZQ; ++q) {
    const int l = min(lg + LPT * q, L - 1);
    const long long i = bb * L + l;
    if (k.z) { r.lo[q] = k.z[i]; r.sc[q] = 0.f; r.ep[q] = 0.f; }
    else { r.lo[q] = k.loc[i]; r.sc[q] = k.scale[i]; r.ep[q] = slode_eps_at(k.rng, k.eps, bb, L, l); }
  }
}
template <int LPT>
__device__ __forceinline__ void latent_store(const DpK& k, long long bb, bool live, int lg, int L, int LP, float* s_zrow, const LatentRegs<LPT>& r) {
  constexpr int ZQ = (SLODE_MAX_L + LPT - 1) / LPT;
#pragma unroll
  for (int q = 0; q < ZQ; ++q) {
    const int l = lg + LPT * q;
    if (l < LP) {
      float zl = 0.f;
      if (l < L && live) {
        const long long i = bb * L + l;
        if (k.z) zl = r.lo[q];
        else {
          if (k.rng.on && k.eps_out) k.eps_out[i] = r.ep[q];
          zl = fmaf(r.sc[q], r.ep[q], r.lo[q]);
        }
        if (k.z_out) k.z_out[i] = zl;
      }
      s_zrow[l] = zl;
    }
  }
}

template <int S, int H>
__global__ void __launch_bounds__(DNT) dopri5_kernel(const DpK k) {
  static_assert(S <= G && H <= G * JL && H <= 32, "one state component and JL hidden units per lane; unit bits in one word");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, g = tid & (G - 1), slot = tid >> 3, b = blockIdx.x * TPB + slot, L = k.L, T = k.T;
  GroupLds<H> m;
  const int LP = (L + 3) & ~3;
  float* s_z = m.carve(smem, TPB);      // [TPB][LP]
  float* s_times = s_z + TPB * LP;      // [T]
  const bool live = b < k.B, own = g < S;
  const long long bb = live ? b : 0;
  LatentRegs<G> zr;
  latent_request<G>(k, bb, g, L, zr);
  UnitRegs<H> ur;
  units_request<S, H>(k.wh, k.bh, k.wg, k.wd, k.w1, k.b1, k.times, T, s_times, L, tid, DNT, m, ur);   // every set-up operand: one round trip
  InitRegs<S, H> ir;
  init_request<S, H>(k.w2, k.b2, g, ir);
  latent_store<G>(k, bb, live, g, L, LP, s_z + slot * LP, zr);
  Units w;
  float pre0[JL];
  load_units<S, H>(ur, k.bg, k.bd, s_z + slot * LP, L, tid, DNT, s_times, T, m, w, pre0);
  if (k.tabs) dump_tables<H>(k.tabs, smem, pre0, slot, g, true, tid, DNT);
  // The controller only integrates forward in time: a grid that is not strictly increasing (torchdiffeq accepts a decreasing one by
  // integrating in -t; this engine rejects it -- Engine.set_times raises, and a caller that comes through the bare C ABI gets NaN
  // trajectories, hence a NaN loss, instead of a quietly extrapolated dense output) fails the solve for the whole workgroup.
  int bad_grid = 0;
  for (int i = tid; i + 1 < T; i += DNT) bad_grid |= !(s_times[i + 1] > s_times[i]);
  bad_grid = __syncthreads_or(bad_grid);
  const float t_first = s_times[0];
  const float4* tab = m.tab + slot * (H + 1) * G;
  const float* ctr = m.ctr + slot * 32;
  float y = init_state<S, H>(ir, pre0, g, own);
  const int gs = own ? g : 0;
  float* xo = k.x + bb * T * S;
  if (live && own) xo[gs] = y;
  const float rtol = k.rtol, atol = k.atol;
  float t = t_first;
  float a, d;
  eval_ad<H>(t, w, g, own, tab, ctr, a, d);
  float fcur = a - d * y;
  // Hairer's initial step (torchdiffeq _select_initial_step, order 4)
  float dt;
  {
    const float sc = atol + fabsf(y) * rtol;
    const float d0 = group_rms<S>(y / sc, own), d1 = group_rms<S>(fcur / sc, own);
    const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    const float y1 = fmaf(h0, fcur, y);
    eval_ad<H>(t + h0, w, g, own, tab, ctr, a, d);
    const float f1 = a - d * y1;
    const float d2 = group_rms<S>((f1 - fcur) / sc, own) / h0;
    const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
    dt = fminf(100.f * h0, h1);
  }
  float* rrec = k.rec + bb * (S + 2);                      // this trajectory's record of the next accepted step
  const long long rstride = (long long)k.B * (S + 2);
  int j = 1;
  int steps = bad_grid ? k.max_steps : 0, nacc = 0;
  float tj = s_times[j < T ? j : T - 1];   // the next output time, read ahead of its use
  // every lane leaves the loop: either all outputs written or max_steps reached (the missing outputs are then NaN).  The butterflies
  // inside eval_ad / group_rms sit at the top level of the loop body: all 64 lanes execute them.
  while (__any(live && j < T && steps < k.max_steps)) {
    const bool act = live && j < T && steps < k.max_steps;
    ++steps;
    // the five evaluation times of the step are known up front: five independent table look-ups
    const float te[5] = {t + dt * (1.f / 5), t + dt * (3.f / 10), t + dt * (4.f / 5), t + dt * (8.f / 9), t + dt};   // stages 6 and 7 share t + dt
    float av[5], dv[5];
    int rv[5];
    eval_ad_batch<H, 5>(te, w, g, own, tab, ctr, av, dv, rv);
    const float a2 = av[0], d2 = dv[0], a3 = av[1], d3 = dv[1], a4 = av[2], d4 = dv[2], a5 = av[3], d5 = dv[3], a6 = av[4], d6 = dv[4];
    const float k2 = a2 - d2 * fmaf(dt, (1.f / 5) * fcur, y);
    const float k3 = a3 - d3 * fmaf(dt, (3.f / 40) * fcur + (9.f / 40) * k2, y);
    const float k4 = a4 - d4 * fmaf(dt, (44.f / 45) * fcur + (-56.f / 15) * k2 + (32.f / 9) * k3, y);
    const float k5 = a5 - d5 * fmaf(dt, (19372.f / 6561) * fcur + (-25360.f / 2187) * k2 + (64448.f / 6561) * k3 + (-212.f / 729) * k4, y);
    const float k6 = a6 - d6 * fmaf(dt, (9017.f / 3168) * fcur + (-355.f / 33) * k2 + (46732.f / 5247) * k3 + (49.f / 176) * k4 + (-5103.f / 18656) * k5, y);
    const float y1 = fmaf(dt, (35.f / 384) * fcur + (500.f / 1113) * k3 + (125.f / 192) * k4 + (-2187.f / 6784) * k5 + (11.f / 84) * k6, y);
    const float k7 = a6 - d6 * y1;
    const float e = dt * ((35.f / 384 - 1951.f / 21600) * fcur + (500.f / 1113 - 22642.f / 50085) * k3 + (125.f / 192 - 451.f / 720) * k4 +
                          (-2187.f / 6784 + 12231.f / 42400) * k5 + (11.f / 84 - 649.f / 6300) * k6 + (-1.f / 60) * k7);
    const float ratio = group_rms_fast<S>(e * DP5_RCP(atol + rtol * fmaxf(fabsf(y), fabsf(y1))), own);
    // a step at the resolution floor of fp32 time is accepted regardless (torchdiffeq would raise 'underflow in dt')
    const bool accept = act && (ratio <= 1.f || dt <= 16.f * 1.1920929e-7f * fmaxf(fabsf(t), 1.f));
    if (accept) {
      const float t1 = t + dt;
      if (k.rec && nacc < k.kmax) {
        if (g == 0) { rrec[0] = t; rrec[1] = dt; }
        if (own) rrec[2 + gs] = y;
      }
      rrec += rstride;   // (the record of step nacc sits at rec + (nacc B + b)(S + 2): advanced, not multiplied out, per accepted step)
      ++nacc;
      if (j < T && tj <= t1) {
        const float ymid = fmaf(dt, (6025192743.f / 30085553152.f / 2) * fcur + (51252292925.f / 65400821598.f / 2) * k3 +
                                        (-2691868925.f / 45128329728.f / 2) * k4 + (187940372067.f / 1594534317056.f / 2) * k5 +
                                        (-1776094331.f / 19743644256.f / 2) * k6 + (11237099.f / 235043384.f / 2) * k7, y);
        const float ca = 2.f * dt * (k7 - fcur) - 8.f * (y1 + y) + 16.f * ymid;
        const float cb = dt * (5.f * fcur - 3.f * k7) + 18.f * y + 14.f * y1 - 32.f * ymid;
        const float cc = dt * (k7 - 4.f * fcur) - 11.f * y - 5.f * y1 + 16.f * ymid;
        const float cd = dt * fcur;
        const float rdt = DP5_RCP(dt);
        while (j < T && tj <= t1) {
#ifdef SLODE_DP5_PRECISE
          const double xq = ((double)tj - (double)t) / (double)dt;
          if (own) xo[j * S + gs] = (float)((double)y + xq * ((double)cd + xq * ((double)cc + xq * ((double)cb + xq * (double)ca))));
#else
          const float xq = (tj - t) * rdt;
          if (own) xo[j * S + gs] = y + xq * (cd + xq * (cc + xq * (cb + xq * ca)));
#endif
          ++j;
          tj = s_times[j < T ? j : T - 1];
        }
      }
      t = t1;
      y = y1;
      fcur = k7;
    }
    if (act) {
      float factor;
      if (ratio == 0.f) factor = 10.f;
      else {
#ifdef SLODE_DP5_PRECISE
        const float safe = 0.9f * powf(ratio, -0.2f);
#else
        const float safe = 0.9f * __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(ratio));   // 0.9 ratio^(-1/5) (v_log_f32 is log2)
#endif
        factor = fminf(10.f, fmaxf(safe, ratio < 1.f ? 1.f : 0.2f));
      }
      dt *= factor;
    }
  }
  if (live) {
    if (k.nrec && g == 0) k.nrec[b] = (j < T) ? -1 : nacc;
    // max_steps exhausted (or a grid the controller cannot walk): the outputs that were never reached are NaN, so every consumer --
    // the scorer's loss with or without gradients (SVI.evaluate_loss has no backward kernel to look at nrec), a bare solve -- fails
    // loudly instead of scoring a trajectory that was not integrated to the end
    for (; j < T; ++j)
      if (own) xo[j * S + gs] = __builtin_nanf("");
  }
}

// ---- LPT = 16 / 32 / 64 lanes per trajectory (round 4) ---------------------------------------------------------------------------------
// The eight-lanes-per-trajectory kernel above puts B = 4096 trajectories on 512 waves: half the SIMDs idle, every wave a lone dependent
// chain of ~350 instructions per attempted step, its trip count the MAXIMUM over its eight trajectories.  Here a trajectory has NG = LPT / 8
// lane groups, lane = 8 e + g: the step's five stage times are evaluated SIDE BY SIDE by the groups (ceil(5 / NG) table look-ups per lane
// instead of five), gathered with ten ds_bpermute, and every group then runs the Runge-Kutta combination, the error norm and the controller
// on the same values -- the same operations in the same order as above, so the accepted steps, the records and the outputs are bit for bit
// those of the eight-lane kernel.  More waves (1024 / 2048 / 4096 at B = 4096) with fewer instructions each; what the measurement says
// about the three widths is in DESIGN 3.3 (LPT = 64 turns the latency problem into an equally large throughput problem: every lane group
// repeats the combination, and the scalar bookkeeping of 16 waves per CU saturates the issue slots).
constexpr int WNTH = 256;         // threads per workgroup: WNTH / LPT trajectories
template <int S, int H, int LPT>
__global__ void __launch_bounds__(WNTH) dopri5_lpt_kernel(const DpK k) {
  static_assert(S <= G && H <= G * JL && H <= 32, "one state component and JL hidden units per lane; unit bits in one word");
  static_assert(LPT == 16 || LPT == 32 || LPT == 64, "lane groups of eight inside a wave");
  constexpr int NG = LPT / G, NR = (5 + NG - 1) / NG, WTP = WNTH / LPT;   // lane groups, evaluation rounds, trajectories per workgroup
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane & (G - 1), e = (lane & (LPT - 1)) >> 3, slot = tid / LPT, b = blockIdx.x * WTP + slot;
  const int L = k.L, T = k.T;
  GroupLds<H> m;
  const int LP = (L + 3) & ~3;
  float* s_z = m.carve(smem, WTP);      // [WTP][LP]
  float* s_times = s_z + WTP * LP;      // [T]
  const bool live = b < k.B, own = g < S;
  const long long bb = live ? b : 0;
  LatentRegs<LPT> zr;
  latent_request<LPT>(k, bb, lane & (LPT - 1), L, zr);
  UnitRegs<H> ur;
  units_request<S, H>(k.wh, k.bh, k.wg, k.wd, k.w1, k.b1, k.times, T, s_times, L, tid, WNTH, m, ur);
  InitRegs<S, H> ir;
  init_request<S, H>(k.w2, k.b2, g, ir);
  latent_store<LPT>(k, bb, live, lane & (LPT - 1), L, LP, s_z + slot * LP, zr);
  Units w;
  float pre0[JL];
  load_units<S, H, LPT>(ur, k.bg, k.bd, s_z + slot * LP, L, tid, WNTH, s_times, T, m, w, pre0);
  if (LPT == 16 && k.tabs) dump_tables<H>(k.tabs, smem, pre0, slot, g, e == 0, tid, WNTH);   // (sixteen trajectories per workgroup, as the reverse sweep)
  int bad_grid = 0;
  for (int i = tid; i + 1 < T; i += WNTH) bad_grid |= !(s_times[i + 1] > s_times[i]);
  bad_grid = __syncthreads_or(bad_grid);
  const float t_first = s_times[0];
  const float4* tab = m.tab + slot * (H + 1) * G;
  const float* ctr = m.ctr + slot * 32;
  float y = init_state<S, H>(ir, pre0, g, own);
  const int gs = own ? g : 0;
  const bool wr = own && e == 0;          // the lane group that writes the trajectory's outputs and records
  float* xo = k.x + bb * T * S;
  if (live && wr) xo[gs] = y;
  const float rtol = k.rtol, atol = k.atol;
  float t = t_first;
  float a, d;
  eval_ad<H>(t, w, g, own, tab, ctr, a, d);
  float fcur = a - d * y;
  float dt;
  {   // Hairer's initial step (torchdiffeq _select_initial_step, order 4)
    const float sc = atol + fabsf(y) * rtol;
    const float d0 = group_rms<S>(y / sc, own), d1 = group_rms<S>(fcur / sc, own);
    const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    const float y1 = fmaf(h0, fcur, y);
    eval_ad<H>(t + h0, w, g, own, tab, ctr, a, d);
    const float f1 = a - d * y1;
    const float d2 = group_rms<S>((f1 - fcur) / sc, own) / h0;
    const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
    dt = fminf(100.f * h0, h1);
  }
  float* rrec = k.rec + bb * (S + 2);                      // this trajectory's record of the next accepted step
  const long long rstride = (long long)k.B * (S + 2);
  int j = 1;
  int steps = bad_grid ? k.max_steps : 0, nacc = 0;
  float tj = s_times[j < T ? j : T - 1];   // the next output time, read ahead of its use
  float tj1 = s_times[min(j + 1, T - 1)];  // (sixteen lanes per trajectory: and the one after)
  // stage i (t + dt {1/5, 3/10, 4/5, 8/9, 1}; stages 6 and 7 share the last) is evaluated by lane group i % NG in round i / NG
  const int base = ((lane & ~(LPT - 1)) + g) << 2;   // ds_bpermute byte index of lane (this trajectory, group 0, component g)
  // every lane leaves the loop: either all outputs written or max_steps reached (the missing outputs are then NaN).  The butterflies inside
  // eval_ad / group_rms and the gathers sit at the top level of the loop body: all 64 lanes execute them.
  while (__any(live && j < T && steps < k.max_steps)) {
    const bool act = live && j < T && steps < k.max_steps;
    ++steps;
    const float te5[5] = {t + dt * (1.f / 5), t + dt * (3.f / 10), t + dt * (4.f / 5), t + dt * (8.f / 9), t + dt};
    float ar[NR], dr[NR], ter[NR];
    int rr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float te = te5[4];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        if (i / NG == r) te = (e == i % NG) ? te5[i] : te;   // (groups without a stage in this round repeat the last one)
      ter[r] = te;
    }
    eval_ad_batch<H, NR>(ter, w, g, own, tab, ctr, ar, dr, rr);   // the rounds' table reads in one batch (one LDS round trip per attempt)
    float av[5], dv[5];
    if (LPT == 16) {
      // two lane groups = the two halves of a DPP row: the other group's value is one row rotation by eight lanes away (no LDS round trip)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float oa = __uint_as_float(dpp_u<0x128>(__float_as_uint(ar[r]))), od = __uint_as_float(dpp_u<0x128>(__float_as_uint(dr[r])));   // row_ror:8
        av[2 * r] = e == 0 ? ar[r] : oa; dv[2 * r] = e == 0 ? dr[r] : od;
        av[2 * r + 1] = e == 0 ? oa : ar[r]; dv[2 * r + 1] = e == 0 ? od : dr[r];
      }
      av[4] = ar[2]; dv[4] = dr[2];   // (both groups evaluated t + dt in the last round)
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int srcl = base + ((i % NG) << 5);
        av[i] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(srcl, (int)__float_as_uint(ar[i / NG])));
        dv[i] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(srcl, (int)__float_as_uint(dr[i / NG])));
      }
    }
    const float a2 = av[0], d2 = dv[0], a3 = av[1], d3 = dv[1], a4 = av[2], d4 = dv[2], a5 = av[3], d5 = dv[3], a6 = av[4], d6 = dv[4];
    const float k2 = a2 - d2 * fmaf(dt, (1.f / 5) * fcur, y);
    const float k3 = a3 - d3 * fmaf(dt, (3.f / 40) * fcur + (9.f / 40) * k2, y);
    const float k4 = a4 - d4 * fmaf(dt, (44.f / 45) * fcur + (-56.f / 15) * k2 + (32.f / 9) * k3, y);
    const float k5 = a5 - d5 * fmaf(dt, (19372.f / 6561) * fcur + (-25360.f / 2187) * k2 + (64448.f / 6561) * k3 + (-212.f / 729) * k4, y);
    const float k6 = a6 - d6 * fmaf(dt, (9017.f / 3168) * fcur + (-355.f / 33) * k2 + (46732.f / 5247) * k3 + (49.f / 176) * k4 + (-5103.f / 18656) * k5, y);
    const float y1 = fmaf(dt, (35.f / 384) * fcur + (500.f / 1113) * k3 + (125.f / 192) * k4 + (-2187.f / 6784) * k5 + (11.f / 84) * k6, y);
    const float k7 = a6 - d6 * y1;
    const float er = dt * ((35.f / 384 - 1951.f / 21600) * fcur + (500.f / 1113 - 22642.f / 50085) * k3 + (125.f / 192 - 451.f / 720) * k4 +
                           (-2187.f / 6784 + 12231.f / 42400) * k5 + (11.f / 84 - 649.f / 6300) * k6 + (-1.f / 60) * k7);
    const float ratio = group_rms_fast<S>(er * DP5_RCP(atol + rtol * fmaxf(fabsf(y), fabsf(y1))), own);
    // a step at the resolution floor of fp32 time is accepted regardless (torchdiffeq would raise 'underflow in dt')
    const bool accept = act && (ratio <= 1.f || dt <= 16.f * 1.1920929e-7f * fmaxf(fabsf(t), 1.f));
    if (accept) {
      const float t1 = t + dt;
      if (k.rec && nacc < k.kmax && e == 0) {
        if (g == 0) { rrec[0] = t; rrec[1] = dt; }
        if (own) rrec[2 + gs] = y;
      }
      rrec += rstride;
      ++nacc;
      if (j < T && tj <= t1) {
        const float ymid = fmaf(dt, (6025192743.f / 30085553152.f / 2) * fcur + (51252292925.f / 65400821598.f / 2) * k3 +
                                        (-2691868925.f / 45128329728.f / 2) * k4 + (187940372067.f / 1594534317056.f / 2) * k5 +
                                        (-1776094331.f / 19743644256.f / 2) * k6 + (11237099.f / 235043384.f / 2) * k7, y);
        const float ca = 2.f * dt * (k7 - fcur) - 8.f * (y1 + y) + 16.f * ymid;
        const float cb = dt * (5.f * fcur - 3.f * k7) + 18.f * y + 14.f * y1 - 32.f * ymid;
        const float cc = dt * (k7 - 4.f * fcur) - 11.f * y - 5.f * y1 + 16.f * ymid;
        const float cd = dt * fcur;
        const float rdt = DP5_RCP(dt);
        if (LPT == 16) {
          // the step's outputs two at a time: lane group e evaluates and writes output j + e (half the trips of the one-at-a-time loop)
          while (j < T && tj <= t1) {
            const float tq = e ? tj1 : tj;
            const bool in = j + e < T && tq <= t1;
#ifdef SLODE_DP5_PRECISE
            const double xq = ((double)tq - (double)t) / (double)dt;
            if (own && in) xo[(j + e) * S + gs] = (float)((double)y + xq * ((double)cd + xq * ((double)cc + xq * ((double)cb + xq * (double)ca))));
#else
            const float xq = (tq - t) * rdt;
            if (own && in) xo[(j + e) * S + gs] = y + xq * (cd + xq * (cc + xq * (cb + xq * ca)));
#endif
            j += (j + 1 < T && tj1 <= t1) ? 2 : 1;
            tj = s_times[min(j, T - 1)];
            tj1 = s_times[min(j + 1, T - 1)];
          }
        } else {
        while (j < T && tj <= t1) {
#ifdef SLODE_DP5_PRECISE
          const double xq = ((double)tj - (double)t) / (double)dt;
          if (wr) xo[j * S + gs] = (float)((double)y + xq * ((double)cd + xq * ((double)cc + xq * ((double)cb + xq * (double)ca))));
#else
          const float xq = (tj - t) * rdt;
          if (wr) xo[j * S + gs] = y + xq * (cd + xq * (cc + xq * (cb + xq * ca)));
#endif
          ++j;
          tj = s_times[j < T ? j : T - 1];
        }
        }
      }
      t = t1;
      y = y1;
      fcur = k7;
    }
    if (act) {
      float factor;
      if (ratio == 0.f) factor = 10.f;
      else {
#ifdef SLODE_DP5_PRECISE
        const float safe = 0.9f * powf(ratio, -0.2f);
#else
        const float safe = 0.9f * __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(ratio));   // 0.9 ratio^(-1/5) (v_log_f32 is log2)
#endif
        factor = fminf(10.f, fmaxf(safe, ratio < 1.f ? 1.f : 0.2f));
      }
      dt *= factor;
    }
  }
  if (live) {
    if (k.nrec && (lane & (LPT - 1)) == 0) k.nrec[b] = (j < T) ? -1 : nacc;
    for (; j < T; ++j)   // max_steps exhausted (or a grid the controller cannot walk): the outputs that were never reached are NaN
      if (wr) xo[j * S + gs] = __builtin_nanf("");
  }
}

// ---- reverse mode over the recorded steps --------------------------------------------------------------------------------
struct DpBK {
  int B, T, L, kmax, drop_z;   // drop_z: reference_adjoint semantics -- z is not an adjoint parameter of the dynamics
  const float *times, *z, *gx, *rec;
  const int* nrec;
  const float *w1, *b1, *w2, *b2, *wh, *bh, *wg, *bg, *wd, *bd;
  float *g_loc, *g_scale, *slabs;   // the latent gradient through the solver is ADDED to the scorer's dLoss/dloc, dLoss/dscale [B][L]
  const float* eps;                 // (z = loc + scale eps);  slabs: one row per workgroup, slot 0 = loss (0 / NaN on a failed solve), then the ode segment
  const float* tabs;           // the forward kernel's tables of this workgroup's trajectories (dump_tables) or nullptr: rebuilt here
  float* snap;                 // [B][2][H][4S] running sums parked when the sweep passes a switching time, by lane group and RANK (see grp::sweep_sample)
  int slab_stride, nseg, stage_gx;
  int o_w1, o_b1, o_w2, o_b2, o_wh, o_bh, o_wg, o_bg, o_wd, o_bd;
  // folded encoder path: once the latent gradient is complete the kernel also runs the encoder heads + tanh backward of its trajectories
  // (g_pre [B][64], glat [B][128] = [g_loc | pad | g_scale * scale | pad]; see ode_elbo_kernel) -- g_pre == nullptr: not wanted
  const float *scale, *enc_hid, *enc_zloc_w, *enc_zls_w;
  float *g_pre, *glat;
  int Hc, zw_off;   // zw_off: floats from s_big to the dedicated [2][L][Hc] copy of the encoder head weights
};

namespace grp {
// Reverse sweep in the forward kernel's mapping: lane g = state component g.  The reverse mode of a Dormand-Prince step is component-wise
// (the right-hand side a_g(t) - d_g(t) y_g couples components only through the time-dependent coefficients), so every quantity of the
// step -- stage states, stage adjoints, dense-output sums, the running sums of the weight gradients -- is ONE scalar per lane: nothing
// spilled (tools/check_spills.py), no per-stage staging through LDS.
// SIXTEEN LANES PER TRAJECTORY (round 4; eight before): a lone wave issues one vector instruction per ~4.2 cycles whatever its
// dependences (tools/ubench/lone_wave_issue.hip), so the sweep's time is its instruction count, and with eight lanes half the SIMDs
// had no wave.  The two lane groups e = 0 / 1 of a trajectory (the halves of a DPP row) run the recomputation and the adjoint chain side
// by side on the same values, and SHARE what is per stage: group e evaluates the coefficients of stages e, e + 2, e + 4 (three table
// look-ups instead of six; the other three arrive by a row rotation) and owns those stages' samples of the weight-gradient sweep --
// its own running sums, segment indices and parked snapshots, added to the other group's after the sweep.  Four waves per workgroup, one
// per SIMD; the epilogues (column sums, outer products, encoder heads) have twice the threads.
constexpr int BNT = 256;        // threads per workgroup
constexpr int LB = 16;          // lanes per trajectory
constexpr int BTP = BNT / LB;   // trajectories per workgroup (= per slab row)
constexpr int TS = BTP + 1;     // padded trajectory stride of the column-sum tile

// Weight gradients.  Along the time-ordered sequence of evaluation times (all stages of all accepted steps) unit j is switched on over
// a prefix or a suffix, so its share of every head-weight gradient is a partial sum of the per-sample head gradients g (and of g t) up
// to the sample where the sweep passes its switching time.  The sweep walks the samples backwards in time keeping the running sums
// RS = sum g, RT = sum g t (this lane's growth and degradation channel) and the segment index of the last sample; when the index
// falls, the running sums are parked in `snap` for every switching time passed (global, [trajectory][rank][4S], written once per
// rank at most).  After the sweep
//   GM_j = snapshot (unit on at late times) | total - snapshot (on at early times) | total (always on) | 0 (never on),   GT_j likewise,
//   dW[r][j] = w_t,j GT_j[r] + u_j GM_j[r],  dLoss/du_j = sum_r W[r][j] GM_j[r],  dLoss/dw_t,j = sum_r W[r][j] GT_j[r].
// one sample of the sweep (decreasing time): its segment index `now` can only fall; every switching time passed since the previous
// sample -- ranks [now, prev) -- parks the running sums (snap is indexed by RANK), then the sample is added
template <int S>
__device__ __forceinline__ void sweep_sample(float t, int now, float ga, float gd, float& RSa, float& RSd, float& RTa, float& RTd,
                                             int& prev, int& cnt_first, bool act, bool own, int gs, float* __restrict__ snap) {
  // (prev starts at 0: the first sample parks nothing; cnt_first = the largest index seen = the first sample's)
  while (act && prev > now) {   // at most H parkings per trajectory
    --prev;
    if (own) {
      float* d = snap + prev * 4 * S + gs;
      d[0] = RSa; d[S] = RSd; d[2 * S] = RTa; d[3 * S] = RTd;
    }
  }
  prev = act ? now : prev;
  cnt_first = act ? max(cnt_first, now) : cnt_first;
  const float a = act ? ga : 0.f, d = act ? gd : 0.f;
  RSa += a; RSd += d;
  RTa = fmaf(a, t, RTa); RTd = fmaf(d, t, RTd);
}

template <int S, int H>
__global__ void __launch_bounds__(BNT) __attribute__((amdgpu_waves_per_eu(2))) dopri5_bwd_kernel(const DpBK k) {   // (register budget of two waves per SIMD: left alone, the scheduler parks values in AGPRs)
  static_assert(S <= G && H <= G * JL && H <= 32, "one state component and JL hidden units per lane; unit bits in one word");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTMP = 2 * S + 2;
  const int tid = threadIdx.x, g = tid & (G - 1), e = (tid >> 3) & 1, slot = tid / LB, b = blockIdx.x * BTP + slot, L = k.L, T = k.T;
  GroupLds<H> m;
  float* s_gu = m.carve(smem, BTP);     // [BTP][32] dLoss/du_j
  float* s_gp = s_gu + BTP * 32;        // [BTP][32] dLoss/d(init-net pre-activation j)
  float* s_h0 = s_gp + BTP * 32;        // [BTP][32] init-net hidden values
  float* s_go = s_h0 + BTP * 32;        // [BTP][8]  dLoss/d(init-net output pre-activation)
  const int LP = (L + 3) & ~3;
  float* s_z = s_go + BTP * 8;          // [BTP][LP]
  float* s_times = s_z + BTP * LP;      // [T]
  float* s_big = s_times + ((T + 3) & ~3);   // dL/dx rows of the workgroup's trajectories | column-sum tile | W1, W_z for the latent gradient
  const float *s_wt = m.wt, *s_wgd = m.wgd;
  const bool live = b < k.B, own = g < S;
  const long long bb = live ? b : 0;
  const int gs = own ? g : 0;
  float* row = k.slabs + (long long)blockIdx.x * k.slab_stride;
  float* prm = row + 1;
  UnitRegs<H> ur;   // the set-up's operands are requested first (its wait also covers the two larger DMA transfers below)
  float pre0[JL];
  if (k.tabs) {   // the forward kernel's tables of the same sixteen trajectories: one 16-byte LDS-DMA pass instead of the set-up
    const float* src = k.tabs + (size_t)blockIdx.x * tab_floats<H>();
    constexpr int n = GroupLds<H>::floats(BTP);
    static_assert(BTP == 16 && n % 4 == 0, "the forward kernels dump the tables of sixteen trajectories");
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, NW = BNT >> 6;
    for (int b0 = wv * 256; b0 < n; b0 += NW * 256)
      if (b0 + lane * 4 < n)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b0 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + b0), 16, 0, 0);
    dma_flat(s_times, k.times, T, wv, NW, lane);
#pragma unroll
    for (int i = 0; i < JL; ++i) pre0[i] = src[n + slot * 32 + g + G * i];
  } else
    units_request<S, H>(k.wh, k.bh, k.wg, k.wd, k.w1, k.b1, k.times, T, s_times, L, tid, BNT, m, ur);
  {   // encoder head weights for the epilogue: LDS-DMA now, awaited with the dL/dx rows (the sweep hides both).  (These kernel arguments
      // are read through an opaque copy of the kernel-argument pointer, here and in the epilogue: see there.)
    typedef const __attribute__((address_space(4))) char* kaptr;
    typedef float* fptr_t;
    typedef const float* cfptr_t;
    kaptr ka = (kaptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
#define DP5_KFIELD(T, name) (*(const __attribute__((address_space(4))) T*)(ka + offsetof(DpBK, name)))
    if (DP5_KFIELD(fptr_t, g_pre) != nullptr) {
      const float* const zl = DP5_KFIELD(cfptr_t, enc_zloc_w);
      const float* const zs = DP5_KFIELD(cfptr_t, enc_zls_w);
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, NW = BNT >> 6, n = L * DP5_KFIELD(int, Hc);
      float* s_zw = s_big + DP5_KFIELD(int, zw_off);
      for (int b0 = wv * 64; b0 < n; b0 += NW * 64)
        if (b0 + lane < n) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zl + b0 + lane),
                                           (__attribute__((address_space(3))) void*)(s_zw + b0), 4, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zs + b0 + lane),
                                           (__attribute__((address_space(3))) void*)(s_zw + n + b0), 4, 0, 0);
        }
    }
#undef DP5_KFIELD
  }
  if (k.stage_gx) {
    // The workgroup's BTP rows of dL/dx are one contiguous block: global -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane, 1 KB per
    // wave instruction, no registers), issued before anything else and awaited only right before the sweep -- the whole set-up hides it.
    const long long base = (long long)blockIdx.x * BTP * T * S;
    const int ntr = min(BTP, k.B - blockIdx.x * BTP), n = ntr * T * S;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, NW = BNT >> 6;
    if ((n & 3) == 0 && (base & 3) == 0) {
      for (int b0 = wv * 256; b0 < n; b0 += NW * 256)
        if (b0 + lane * 4 < n)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k.gx + base + b0 + lane * 4),
                                           (__attribute__((address_space(3))) void*)(s_big + b0), 16, 0, 0);
    } else {
      for (int b0 = wv * 64; b0 < n; b0 += NW * 64)
        if (b0 + lane < n)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(k.gx + base + b0 + lane),
                                           (__attribute__((address_space(3))) void*)(s_big + b0), 4, 0, 0);
    }
  }
  for (int i = 1 + tid; i <= k.nseg; i += BNT) row[i] = 0.f;   // (every element is written again below; completes long before)
  {
    constexpr int ZQ = SLODE_MAX_L / G;
    float zv[ZQ];
#pragma unroll
    for (int q = 0; q < ZQ; ++q) zv[q] = k.z[bb * L + min(g + G * q, L - 1)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < ZQ; ++q)
      if (g + G * q < LP) s_z[slot * LP + g + G * q] = (live && g + G * q < L) ? zv[q] : 0.f;
  }
  Units w;
  unsigned dirmask;
  if (k.tabs) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed; the barrier covers the others'
    __syncthreads();
    unsigned dm = 0u;
#pragma unroll
    for (int i = 0; i < JL; ++i) {   // this lane's units, as load_units leaves them
      const int j = g + G * i;
      const bool valid = j < H;
      w.wt[i] = valid ? m.wt[j] : 0.f;
      w.u[i] = m.u[slot * 32 + j];
      w.th[i] = m.th[slot * 32 + j];
      dm |= (w.wt[i] >= 0.f || !valid) ? (1u << j) : 0u;
    }
    dirmask = group_or(dm);
  } else
    dirmask = load_units<S, H, LB>(ur, k.bg, k.bd, s_z + slot * LP, L, tid, BNT, s_times, T, m, w, pre0);
  const float* s_us = m.u + slot * 32;
  const int* s_rnk = m.rnk + slot * 32;
  const float4* tab = m.tab + slot * (H + 1) * G;
  const float* ctr = m.ctr + slot * 32;
  const int nr = live ? k.nrec[bb] : 0;
  const bool bad = nr < 0 || nr > k.kmax;
  const int K = bad ? 0 : nr;
  const float* gxb = k.gx + bb * T * S + gs;
  const float* gxs = s_big + slot * T * S + gs;
  float* snap = k.snap + (bb * 2 + e) * H * 4 * S;   // this lane group's snapshots
  float lam = 0.f, RSa = 0.f, RSd = 0.f, RTa = 0.f, RTd = 0.f;
  int cnt_prev = 0, cnt_first = 0;   // segment index of the previous sample of the sweep (falls along it) and of its first sample
  if (k.stage_gx) {   // the LDS-DMA of dL/dx issued at the top has landed (every wave drains its own, the barrier covers the others')
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  int j = T - 1;
  float tj = s_times[j], gj = k.stage_gx ? gxs[j * S] : gxb[j * S];   // the next output sample of the sweep
  gj = own ? gj : 0.f;
  // The record of step K-1-it is loaded two steps ahead of its use into one of three register sets used in rotation (the loop is
  // unrolled by three): a set is never copied, so nothing waits for a load before the step that consumes it.
  float ta, dta, ya, tb, dtb, yb, tc, dtc, yc;
  const long long rstride = (long long)k.B * (S + 2);   // records of consecutive steps are B (S + 2) floats apart
  const float* rp = k.rec + (long long)(K > 0 ? K - 1 : 0) * rstride + bb * (S + 2);   // step K-1 first
#define SLODE_LDREC(IT, T_, DT_, Y_)                                                            \
  {                                                                                             \
    const bool a_ = (IT) < K;                                                                   \
    const float r0_ = rp[0], r1_ = rp[1], r2_ = rp[2 + gs];                                     \
    T_ = a_ ? r0_ : 0.f; DT_ = a_ ? r1_ : 0.f; Y_ = (a_ && own) ? r2_ : 0.f;                    \
    rp -= ((IT) + 1 < K) ? rstride : 0;                                                         \
  }
  SLODE_LDREC(0, ta, dta, ya)
  SLODE_LDREC(1, tb, dtb, yb)
  // every lane leaves the loop after max(K) <= kmax iterations; the butterflies sit at the top level of the body (all lanes run them)
  // `request`: the read-ahead of the step record after next.  It goes out BEHIND the loop over the step's output samples: that loop may read
  // dL/dx from global memory (rows that do not fit in LDS), so it waits with s_waitcnt vmcnt(0) -- which, counting in order, also waits for
  // whatever was requested before it; requested at the top of the step the records were a memory round trip per step on the critical path
  auto step = [&](const float t, const float dt, const float y, const bool act, auto&& request) __attribute__((always_inline)) {
    const float te0 = t, te1 = t + dt * (1.f / 5), te2 = t + dt * (3.f / 10), te3 = t + dt * (4.f / 5), te4 = t + dt * (8.f / 9), te5 = t + dt;
    // ---- forward recomputation of the stages from the recorded (t, dt, y) ------------------------------------------------------
    // this lane group's three stage times: three independent table look-ups, their LDS reads in one batch; the other group's
    // coefficients by a rotation of the DPP row (sixteen lanes = one trajectory) by eight lanes
    const float mt[3] = {e ? te1 : te0, e ? te3 : te2, e ? te5 : te4};
    float ma[3], md[3];
    int mr[3];
    eval_ad_batch<H, 3>(mt, w, g, own, tab, ctr, ma, md, mr);
    float oa[3], od[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      oa[r] = __uint_as_float(dpp_u<0x128>(__float_as_uint(ma[r])));   // row_ror:8
      od[r] = __uint_as_float(dpp_u<0x128>(__float_as_uint(md[r])));
    }
    const float a0 = e ? oa[0] : ma[0], a1 = e ? ma[0] : oa[0], a2 = e ? oa[1] : ma[1], a3 = e ? ma[1] : oa[1], a4 = e ? oa[2] : ma[2], a5 = e ? ma[2] : oa[2];
    const float d0 = e ? od[0] : md[0], d1 = e ? md[0] : od[0], d2 = e ? od[1] : md[1], d3 = e ? md[1] : od[1], d4 = e ? od[2] : md[2], d5 = e ? md[2] : od[2];
    // sigmoid' of the stages whose samples this group owns
    const float map0 = ma[0] * (1.f - ma[0]), mdp0 = md[0] * (1.f - md[0]), map1 = ma[1] * (1.f - ma[1]), mdp1 = md[1] * (1.f - md[1]);
    const float map2 = ma[2] * (1.f - ma[2]), mdp2 = md[2] * (1.f - md[2]);
    const float k1 = a0 - d0 * y;
    const float ys2 = fmaf(dt, (1.f / 5) * k1, y);
    const float k2 = a1 - d1 * ys2;
    const float ys3 = fmaf(dt, (3.f / 40) * k1 + (9.f / 40) * k2, y);
    const float k3 = a2 - d2 * ys3;
    const float ys4 = fmaf(dt, (44.f / 45) * k1 + (-56.f / 15) * k2 + (32.f / 9) * k3, y);
    const float k4 = a3 - d3 * ys4;
    const float ys5 = fmaf(dt, (19372.f / 6561) * k1 + (-25360.f / 2187) * k2 + (64448.f / 6561) * k3 + (-212.f / 729) * k4, y);
    const float k5 = a4 - d4 * ys5;
    const float ys6 = fmaf(dt, (9017.f / 3168) * k1 + (-355.f / 33) * k2 + (46732.f / 5247) * k3 + (49.f / 176) * k4 + (-5103.f / 18656) * k5, y);
    const float k6 = a5 - d5 * ys6;
    const float y1 = fmaf(dt, (35.f / 384) * k1 + (500.f / 1113) * k3 + (125.f / 192) * k4 + (-2187.f / 6784) * k5 + (11.f / 84) * k6, y);
    // ---- dense outputs inside (t, t + dt]: x_j = y + q cd + q^2 cc + q^3 cb + q^4 ca with q = (times[j] - t) / dt ----------------
    float Ga = 0.f, Gb = 0.f, Gc = 0.f, Gd = 0.f, gy = 0.f;
    const float rdt = __builtin_amdgcn_rcpf(dt);   // (1 ulp: q only places the sample on the step's polynomial)
    while (act && j >= 1 && tj > t) {
      const float q = (tj - t) * rdt, q2 = q * q, q3 = q2 * q, q4 = q2 * q2;
      const float gq = gj;
      --j;
      tj = s_times[j];                                  // read ahead of their use (j >= 0)
      gj = k.stage_gx ? gxs[j * S] : gxb[j * S];
      gj = own ? gj : 0.f;
      gy += gq;
      Gd = fmaf(q, gq, Gd); Gc = fmaf(q2, gq, Gc); Gb = fmaf(q3, gq, Gb); Ga = fmaf(q4, gq, Ga);
    }
    request();
    // ---- reverse mode of the step ------------------------------------------------------------------------------------------------
    const float gf0 = dt * (-2.f * Ga + 5.f * Gb - 4.f * Gc + Gd);
    const float gf1 = dt * (2.f * Ga - 3.f * Gb + Gc);
    const float gm = 16.f * Ga - 32.f * Gb + 16.f * Gc;          // dL/dy_mid
    gy += -8.f * Ga + 18.f * Gb - 11.f * Gc + gm;
    float gy1 = lam - 8.f * Ga + 14.f * Gb - 5.f * Gc;
    const float dgm = dt * gm;
    float g1 = fmaf(dgm, 6025192743.f / 30085553152.f / 2, gf0);
    float g2 = 0.f;
    float g3 = dgm * (51252292925.f / 65400821598.f / 2);
    float g4 = dgm * (-2691868925.f / 45128329728.f / 2);
    float g5 = dgm * (187940372067.f / 1594534317056.f / 2);
    float g6 = dgm * (-1776094331.f / 19743644256.f / 2);
    const float g7 = fmaf(dgm, 11237099.f / 235043384.f / 2, gf1);
    // stage 7: k7 = a5 - d5 * y1   (the samples -- head gradients at a stage time -- are formed below, by the group that owns the stage)
    gy1 = fmaf(-d5, g7, gy1);
    // y1 = y + dt * sum b_i k_i
    gy += gy1;
    const float dg = dt * gy1;
    g1 = fmaf(dg, 35.f / 384, g1); g3 = fmaf(dg, 500.f / 1113, g3); g4 = fmaf(dg, 125.f / 192, g4);
    g5 = fmaf(dg, -2187.f / 6784, g5); g6 = fmaf(dg, 11.f / 84, g6);
    {   // stage 6 (same time as stage 7: one sample for both)
      const float ee = -d5 * g6;
      gy += ee;
      const float de = dt * ee;
      g1 = fmaf(de, 9017.f / 3168, g1); g2 = fmaf(de, -355.f / 33, g2); g3 = fmaf(de, 46732.f / 5247, g3);
      g4 = fmaf(de, 49.f / 176, g4); g5 = fmaf(de, -5103.f / 18656, g5);
    }
    float xa, xd;
    {   // samples at te5 (group 1: stages 7 + 6) | te4 (group 0: stage 5)
      const float gs_ = e ? g7 : g5, ys_ = e ? y1 : ys5;
      xa = gs_ * map2; xd = -gs_ * ys_ * mdp2;
      xa = e ? fmaf(g6, map2, xa) : xa;
      xd = e ? fmaf(-g6 * ys6, mdp2, xd) : xd;
    }
    sweep_sample<S>(mt[2], mr[2], xa, xd, RSa, RSd, RTa, RTd, cnt_prev, cnt_first, act, own, gs, snap);
    {   // stage 5
      const float ee = -d4 * g5;
      gy += ee;
      const float de = dt * ee;
      g1 = fmaf(de, 19372.f / 6561, g1); g2 = fmaf(de, -25360.f / 2187, g2); g3 = fmaf(de, 64448.f / 6561, g3);
      g4 = fmaf(de, -212.f / 729, g4);
    }
    {   // stage 4
      const float ee = -d3 * g4;
      gy += ee;
      const float de = dt * ee;
      g1 = fmaf(de, 44.f / 45, g1); g2 = fmaf(de, -56.f / 15, g2); g3 = fmaf(de, 32.f / 9, g3);
    }
    {   // samples at te3 (group 1: stage 4) | te2 (group 0: stage 3)
      const float gs_ = e ? g4 : g3, ys_ = e ? ys4 : ys3;
      xa = gs_ * map1; xd = -gs_ * ys_ * mdp1;
    }
    sweep_sample<S>(mt[1], mr[1], xa, xd, RSa, RSd, RTa, RTd, cnt_prev, cnt_first, act, own, gs, snap);
    {   // stage 3
      const float ee = -d2 * g3;
      gy += ee;
      const float de = dt * ee;
      g1 = fmaf(de, 3.f / 40, g1); g2 = fmaf(de, 9.f / 40, g2);
    }
    {   // stage 2
      const float ee = -d1 * g2;
      gy += ee;
      g1 = fmaf(dt * ee, 1.f / 5, g1);
    }
    {   // samples at te1 (group 1: stage 2) | te0 (group 0: stage 1)
      const float gs_ = e ? g2 : g1, ys_ = e ? ys2 : y;
      xa = gs_ * map0; xd = -gs_ * ys_ * mdp0;
    }
    sweep_sample<S>(mt[0], mr[0], xa, xd, RSa, RSd, RTa, RTd, cnt_prev, cnt_first, act, own, gs, snap);
    // stage 1
    gy = fmaf(-d0, g1, gy);
    lam = act ? gy : lam;
  };
  for (int it = 0; __any(it < K); it += 3) {
    step(ta, dta, ya, it < K, [&]() __attribute__((always_inline)) { SLODE_LDREC(it + 2, tc, dtc, yc) });
    if (!__any(it + 1 < K)) break;
    step(tb, dtb, yb, it + 1 < K, [&]() __attribute__((always_inline)) { SLODE_LDREC(it + 3, ta, dta, ya) });
    if (!__any(it + 2 < K)) break;
    step(tc, dtc, yc, it + 2 < K, [&]() __attribute__((always_inline)) { SLODE_LDREC(it + 4, tb, dtb, yb) });
  }
#undef SLODE_LDREC
  const bool any_bad = __syncthreads_or(live && bad) != 0;   // (also: every wave is done with the staged dL/dx rows)
  if (tid == 0) row[0] = any_bad ? __builtin_nanf("") : 0.f;
  // ---- the units' partial sums: every trajectory's terms go through one LDS tile [unit][term][trajectory] and are summed over the
  //      workgroup's trajectories in a fixed order --------------------------------------------------------------------------------
  {
    float* tile = s_big;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's own snapshot stores are back-readable
    // unit jj has rank rk among the switching times: passed during the sweep iff cnt_prev <= rk < cnt_first; at the earliest sample
    // `t >= th` holds iff rk < cnt_prev, and the unit is on there iff that agrees with its direction.  Snapshots are read thirteen units at
    // a time (two memory round trips for the 25 units), unconditionally (a select on a loaded value otherwise serialises one round trip
    // per unit).
    constexpr int UB = 13;
    for (int j0 = 0; j0 < H; j0 += UB) {
      int rk[UB];
      float sv[UB][4];
#pragma unroll
      for (int q = 0; q < UB; ++q) rk[q] = s_rnk[min(j0 + q, H - 1)];
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const float* sn = snap + rk[q] * 4 * S + gs;
        sv[q][0] = sn[0]; sv[q][1] = sn[S]; sv[q][2] = sn[2 * S]; sv[q][3] = sn[3 * S];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < UB; ++q) {
        const int jj = j0 + q;
        if (jj < H) {   // (uniform)
          const bool flipped = rk[q] >= cnt_prev && rk[q] < cnt_first, on_early = (rk[q] < cnt_prev) == (((dirmask >> jj) & 1u) != 0u);
          const float sma = flipped ? sv[q][0] : 0.f, smd = flipped ? sv[q][1] : 0.f, sta = flipped ? sv[q][2] : 0.f, std_ = flipped ? sv[q][3] : 0.f;
          // on at early times: flipped ? total - snapshot : total;   off at early times: flipped ? snapshot : 0
          // (this lane group's samples; the other group's share is one row rotation away -- both groups end with the same sums)
          const float gma_e = on_early ? RSa - sma : sma, gmd_e = on_early ? RSd - smd : smd;
          const float gta_e = on_early ? RTa - sta : sta, gtd_e = on_early ? RTd - std_ : std_;
          const float gma = gma_e + __uint_as_float(dpp_u<0x128>(__float_as_uint(gma_e))), gmd = gmd_e + __uint_as_float(dpp_u<0x128>(__float_as_uint(gmd_e)));
          const float gta = gta_e + __uint_as_float(dpp_u<0x128>(__float_as_uint(gta_e))), gtd = gtd_e + __uint_as_float(dpp_u<0x128>(__float_as_uint(gtd_e)));
          const float wt = s_wt[jj], uj = s_us[jj];
          const float w1 = s_wgd[jj * 16 + g], w2 = s_wgd[jj * 16 + 8 + g];   // 0 for lanes without a component
          const float gu = group_add(fmaf(w1, gma, w2 * gmd)), gwt = group_add(fmaf(w1, gta, w2 * gtd));
          if (own && e == 0) {
            tile[(jj * NTMP + g) * TS + slot] = fmaf(wt, gta, uj * gma);
            tile[(jj * NTMP + S + g) * TS + slot] = fmaf(wt, gtd, uj * gmd);
          }
          if (g == 0 && e == 0) {
            tile[(jj * NTMP + 2 * S) * TS + slot] = gu;
            tile[(jj * NTMP + 2 * S + 1) * TS + slot] = gwt;
            s_gu[slot * 32 + jj] = gu;
          }
        }
      }
    }
    {   // the constant-1 unit: head biases (both groups' samples)
      const float ra = RSa + __uint_as_float(dpp_u<0x128>(__float_as_uint(RSa))), rd = RSd + __uint_as_float(dpp_u<0x128>(__float_as_uint(RSd)));
      if (own && e == 0) {
        tile[(H * NTMP + g) * TS + slot] = ra;
        tile[(H * NTMP + S + g) * TS + slot] = rd;
      }
    }
    __syncthreads();
    for (int col = tid; col < (H + 1) * NTMP; col += BNT) {
      const int jj = col / NTMP, c = col - jj * NTMP;
      int o;
      if (jj < H) o = c < S ? k.o_wg + c * H + jj : (c < 2 * S ? k.o_wd + (c - S) * H + jj : (c == 2 * S ? k.o_bh + jj : k.o_wh + jj * (1 + L)));
      else o = c < S ? k.o_bg + c : (c < 2 * S ? k.o_bd + (c - S) : -1);
      float acc = 0.f;
      for (int r = 0; r < BTP; ++r) acc += tile[col * TS + r];
      if (o >= 0) prm[o] = acc;
    }
    __syncthreads();
  }
  // ---- init net: x0 = sigmoid(W2 relu(W1 z + b1) + b2); the j = 0 output is x0 itself -------------------------------------------
  {
    float* s_w1 = s_big;            // [H][L]
    float* s_wz = s_big + H * L;    // [H][L]  z-columns of the dynamics net's hidden layer
    InitRegs<S, H> ir;   // the init net's output layer: requested with the batch below (one round trip for everything this section reads)
    init_request<S, H>(k.w2, k.b2, g, ir);
    for (int i0 = tid; i0 < H * L; i0 += 8 * BNT) {   // W1 and the z-columns of the hidden layer (row pitch 1 + L): ONE batch of loads, then the stores
      float v[8], u1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = min(i0 + q * BNT, H * L - 1), jj = i / L;
        u1[q] = k.w1[i];
        v[q] = k.wh[i + jj + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q * BNT < H * L) { s_w1[i0 + q * BNT] = u1[q]; s_wz[i0 + q * BNT] = v[q]; }
    }
    const float x0 = init_state<S, H>(ir, pre0, g, own);
    const float g0 = (live && !bad && own) ? lam + gxb[0] : 0.f;
    s_go[slot * 8 + g] = g0 * x0 * (1.f - x0);
#pragma unroll
    for (int i = 0; i < JL; ++i) s_h0[slot * 32 + g + G * i] = fmaxf(pre0[i], 0.f);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < JL; ++i) {
      const int jj = g + G * i;
      float gh = 0.f;
      if (jj < H) {
#pragma unroll
        for (int s = 0; s < S; ++s) gh = fmaf(ir.w2[i][s], s_go[slot * 8 + s], gh);
      }
      s_gp[slot * 32 + jj] = pre0[i] > 0.f ? gh : 0.f;
      if (jj >= H) s_gu[slot * 32 + jj] = 0.f;
    }
    __syncthreads();
    // latent gradient of this trajectory: through the init net and (exact mode) through u = W_z z + b_h; the scorer's share it is
    // added to (and eps) is fetched up front, eight latent dims at a time
    // (the folded encoder path: the finished latent gradient also goes to glat and, as [g_loc | g_scale * scale], to the LDS rows the
    // encoder-head epilogue reads -- behind W1 | W_z, which this loop still uses; its kernel arguments come through the kernel-argument
    // pointer, see the epilogue)
    typedef const __attribute__((address_space(4))) char* kaptr0;
    kaptr0 ka0 = (kaptr0)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka0));
    float* const l_glat = *(float* const __attribute__((address_space(4)))*)(ka0 + offsetof(DpBK, glat));
    const float* const l_scale = *(const float* const __attribute__((address_space(4)))*)(ka0 + offsetof(DpBK, scale));
    const bool l_enc = *(float* const __attribute__((address_space(4)))*)(ka0 + offsetof(DpBK, g_pre)) != nullptr;
    float* s_g = s_big + 2 * H * L;   // [BTP][2][L]
    for (int l0 = tid & (LB - 1); l0 < L; l0 += LB * 4) {   // (sixteen lanes per trajectory: four latent dims per lane and pass)
      float gl_[4], gs_[4], ep_[4], sc_[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long i = bb * L + min(l0 + q * LB, L - 1);
        gl_[q] = k.g_loc[i]; gs_[q] = k.g_scale[i]; ep_[q] = k.eps[i];
        sc_[q] = l_enc ? l_scale[i] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
      float a1[4], a2[4];   // through the init net / through the dynamics' hidden layer: 8 independent chains, unit-major
#pragma unroll
      for (int q = 0; q < 4; ++q) { a1[q] = 0.f; a2[q] = 0.f; }
      for (int jj = 0; jj < H; ++jj) {
        const float gp = s_gp[slot * 32 + jj], gu = s_gu[slot * 32 + jj];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int l = min(l0 + q * LB, L - 1);
          a1[q] = fmaf(s_w1[jj * L + l], gp, a1[q]);
          a2[q] = fmaf(s_wz[jj * L + l], gu, a2[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int l = l0 + q * LB;
        if (l < L) {
          float o_loc = 0.f, o_ls = 0.f;
          if (live) {
            const float gl = a1[q] + (k.drop_z ? 0.f : a2[q]);
            const long long i = bb * L + l;
            o_loc = gl_[q] + gl;
            const float o_sc = fmaf(gl, ep_[q], gs_[q]);
            o_ls = o_sc * sc_[q];
            k.g_loc[i] = o_loc;
            k.g_scale[i] = o_sc;
            if (l_enc) { l_glat[bb * 128 + l] = o_loc; l_glat[bb * 128 + 64 + l] = o_ls; }
          }
          if (l_enc) { s_g[(2 * slot) * L + l] = o_loc; s_g[(2 * slot + 1) * L + l] = o_ls; }
        }
      }
    }
    // sums over the workgroup's trajectories (fixed order): outer products with z, with the init net's hidden values, bias columns
    for (int o = tid; o < H * L; o += BNT) {
      const int jj = o / L, l = o - jj * L;
      float a1 = 0.f, a2 = 0.f;
      for (int r = 0; r < BTP; ++r) {
        const float zl = s_z[r * LP + l];
        a1 = fmaf(s_gp[r * 32 + jj], zl, a1);
        a2 = fmaf(s_gu[r * 32 + jj], zl, a2);
      }
      prm[k.o_w1 + o] = a1;
      prm[k.o_wh + jj * (1 + L) + 1 + l] = a2;
    }
    for (int o = tid; o < S * H + H + S; o += BNT) {
      float acc = 0.f;
      if (o < S * H) {
        const int s = o / H, jj = o - s * H;
        for (int r = 0; r < BTP; ++r) acc = fmaf(s_go[r * 8 + s], s_h0[r * 32 + jj], acc);
        prm[k.o_w2 + o] = acc;
      } else if (o < S * H + H) {
        const int jj = o - S * H;
        for (int r = 0; r < BTP; ++r) acc += s_gp[r * 32 + jj];
        prm[k.o_b1 + jj] = acc;
      } else {
        const int s = o - S * H - H;
        for (int r = 0; r < BTP; ++r) acc += s_go[r * 8 + s];
        prm[k.o_b2 + s] = acc;
      }
    }
  }
  {
    // ---- encoder heads + tanh, backward (models/encoder_conv.py:48-51) for the workgroup's trajectories: their latent gradient is
    // final now (the stores above are this workgroup's own; the barrier orders them).  W1 | W_z in s_big are dead.
    // The epilogue's own kernel arguments are read HERE, through the kernel-argument pointer, behind an opaque copy of it, so that they
    // are not live across the sweep.  (Round 3, config[2]: with this epilogue the kernel takes 138 us instead of 125 and spills 61
    // SGPRs instead of 33; reading the fields late did not change either figure -- the step as a whole is still 7 us faster than with
    // the separate encoder-head launch, slab_stage1 and reduce launches this replaces.)
    typedef const __attribute__((address_space(4))) char* kaptr;
    kaptr ka = (kaptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    typedef float* fptr_t;
    typedef const float* cfptr_t;
#define DP5_KFIELD(T, name) (*(const __attribute__((address_space(4))) T*)(ka + offsetof(DpBK, name)))
    float* const e_g_pre = DP5_KFIELD(fptr_t, g_pre);
    if (e_g_pre != nullptr) {
      const float* const e_hid = DP5_KFIELD(cfptr_t, enc_hid);
      const int Hc = DP5_KFIELD(int, Hc), zw_off = DP5_KFIELD(int, zw_off), eB = DP5_KFIELD(int, B);
#undef DP5_KFIELD
      const float* s_zw = s_big + zw_off;     // [2][L][Hc]  z_loc.weight | z_scale.0.weight (in place since the prologue)
      const float* s_g = s_big + 2 * H * L;   // [BTP][2][L] dLoss/dloc | dLoss/dscale * scale: written by the latent-gradient loop above
      __syncthreads();
      // item (hidden unit mm, part rq of the trajectories): the unit's two weights per latent dim are read once for RQ trajectories
      constexpr int NQ = BNT / 64, RQ = BTP / NQ;
      for (int item = tid; item < NQ * Hc; item += BNT) {
        const int rq = item / Hc, mm = item - rq * Hc;
        float g0[RQ], hv[RQ];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
          g0[q] = 0.f;
          hv[q] = e_hid[min((long long)blockIdx.x * BTP + rq * RQ + q, (long long)eB - 1) * Hc + mm];   // (requested ahead of the sums)
        }
        for (int l = 0; l < L; ++l) {
          const float w0 = s_zw[l * Hc + mm], w1 = s_zw[(L + l) * Hc + mm];
#pragma unroll
          for (int q = 0; q < RQ; ++q) {
            const int r = rq * RQ + q;
            g0[q] = fmaf(w0, s_g[(2 * r) * L + l], fmaf(w1, s_g[(2 * r + 1) * L + l], g0[q]));
          }
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
          const long long b2 = (long long)blockIdx.x * BTP + rq * RQ + q;
          if (b2 < eB) e_g_pre[b2 * 64 + mm] = g0[q] * (1.f - hv[q] * hv[q]);
        }
      }
    }
  }
}

}  // namespace grp

}  // namespace

size_t slode_dopri5_tab_floats(const slode_shape& s) { (void)s; return (size_t)tab_floats<25>(); }   // per workgroup of sixteen trajectories (H = 25)
int slode_dopri5_rows(const slode_shape& s) { return (s.B + grp::BTP - 1) / grp::BTP; }   // slab rows of the reverse sweep: one per workgroup

int slode_dopri5_kmax(const slode_shape& s) {
  // record capacity per trajectory: 256 MB of (t, dt, y) records, within [64, 2048] steps
  long long kmax = (1ll << 26) / ((long long)s.B * (s.S + 2));
  return (int)(kmax < 64 ? 64 : (kmax > 2048 ? 2048 : kmax));
}

hipError_t slode_launch_dopri5(const slode_shape& s, const slode_layout& lay, const float* p, const float* times, const float* z,
                               float* x, hipStream_t stream, const DopriRec* rec) {
  DpK k;
  k.B = s.B; k.T = s.T; k.L = s.L; k.times = times; k.z = z; k.x = x;
  k.loc = k.scale = k.eps = nullptr; k.z_out = nullptr; k.rec = nullptr; k.nrec = nullptr; k.kmax = 0;
  k.rng = RngK{}; k.eps_out = nullptr; k.tabs = nullptr;
  if (rec) { k.loc = rec->loc; k.scale = rec->scale; k.eps = rec->eps; k.z_out = rec->z_out; k.rec = rec->rec; k.nrec = rec->nrec; k.kmax = rec->kmax;
             k.rng = rec->rng; k.eps_out = rec->eps_out; k.tabs = rec->tabs; }
  k.w1 = p + lay.init_w1; k.b1 = p + lay.init_b1; k.w2 = p + lay.init_w2; k.b2 = p + lay.init_b2;
  k.wh = p + lay.dyn_wh; k.bh = p + lay.dyn_bh; k.wg = p + lay.dyn_wg; k.bg = p + lay.dyn_bg; k.wd = p + lay.dyn_wd; k.bd = p + lay.dyn_bd;
  k.rtol = s.rtol > 0.f ? s.rtol : 1e-7f;
  k.atol = s.atol > 0.f ? s.atol : 1e-9f;
  k.max_steps = 20000;
  const int lpt = rec ? rec->w64 : 8;    // lanes per trajectory of the forward solve: 8 (the round-2 kernel), 16, 32 or 64 -- all the same bits
  if (lpt == 16 || lpt == 32 || lpt == 64) {
    const int wtp = WNTH / lpt, grid = (s.B + wtp - 1) / wtp;
    const size_t lds = sizeof(float) * ((size_t)GroupLds<25>::floats(wtp) + (size_t)wtp * ((s.L + 3) & ~3) + (size_t)s.T);
#define SLODE_DP5_LPT(SS, LL) SLODE_LAUNCH("dopri5_fwd", (dopri5_lpt_kernel<SS, 25, LL>), dim3(grid), dim3(WNTH), lds, stream, k)
    if (s.H != 25 || (s.S != 5 && s.S != 8)) return hipErrorInvalidValue;
    if (s.S == 5) { if (lpt == 16) SLODE_DP5_LPT(5, 16); else if (lpt == 32) SLODE_DP5_LPT(5, 32); else SLODE_DP5_LPT(5, 64); }
    else { if (lpt == 16) SLODE_DP5_LPT(8, 16); else if (lpt == 32) SLODE_DP5_LPT(8, 32); else SLODE_DP5_LPT(8, 64); }
#undef SLODE_DP5_LPT
    return hipGetLastError();
  }
  const int grid = (s.B + TPB - 1) / TPB;
  const size_t lds = sizeof(float) * ((size_t)GroupLds<25>::floats(TPB) + (size_t)TPB * ((s.L + 3) & ~3) + (size_t)s.T);
  if (s.H == 25 && s.S == 5) SLODE_LAUNCH("dopri5_fwd", (dopri5_kernel<5, 25>), dim3(grid), dim3(DNT), lds, stream, k);
  else if (s.H == 25 && s.S == 8) SLODE_LAUNCH("dopri5_fwd", (dopri5_kernel<8, 25>), dim3(grid), dim3(DNT), lds, stream, k);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t slode_launch_dopri5_bwd(const slode_shape& s, const slode_layout& lay, const float* p, const float* times, const DopriRec& rec,
                                   const float* gx, float* g_loc, float* g_scale, float* slabs, int slab_stride, int drop_z, float* snap,
                                   hipStream_t stream, const float* enc_hid, float* g_pre, float* glat) {
  DpBK k;
  k.scale = rec.scale; k.enc_hid = enc_hid; k.g_pre = g_pre; k.glat = glat; k.Hc = s.Hc;
  k.enc_zloc_w = p + lay.zloc_w; k.enc_zls_w = p + lay.zls_w;
  k.tabs = rec.tabs;
  k.snap = snap; k.g_loc = g_loc; k.g_scale = g_scale; k.eps = rec.eps;
  k.B = s.B; k.T = s.T; k.L = s.L; k.kmax = rec.kmax; k.drop_z = drop_z;
  k.times = times; k.z = rec.z_out; k.gx = gx; k.rec = rec.rec; k.nrec = rec.nrec;
  k.w1 = p + lay.init_w1; k.b1 = p + lay.init_b1; k.w2 = p + lay.init_w2; k.b2 = p + lay.init_b2;
  k.wh = p + lay.dyn_wh; k.bh = p + lay.dyn_bh; k.wg = p + lay.dyn_wg; k.bg = p + lay.dyn_bg; k.wd = p + lay.dyn_wd; k.bd = p + lay.dyn_bd;
  k.slabs = slabs; k.slab_stride = slab_stride; k.nseg = lay.ode_end - lay.ode_begin;
  const int ob = lay.ode_begin;
  k.o_w1 = lay.init_w1 - ob; k.o_b1 = lay.init_b1 - ob; k.o_w2 = lay.init_w2 - ob; k.o_b2 = lay.init_b2 - ob;
  k.o_wh = lay.dyn_wh - ob; k.o_bh = lay.dyn_bh - ob; k.o_wg = lay.dyn_wg - ob; k.o_bg = lay.dyn_bg - ob;
  k.o_wd = lay.dyn_wd - ob; k.o_bd = lay.dyn_bd - ob;
  {
    using namespace grp;
    const int grid = slode_dopri5_rows(s);
    const size_t fixed = (size_t)GroupLds<25>::floats(BTP) + (size_t)BTP * 32 * 3 + BTP * 8 + (size_t)BTP * ((s.L + 3) & ~3) + (size_t)((s.T + 3) & ~3);
    const size_t tile = (size_t)(s.H + 1) * (2 * s.S + 2) * TS, wz = 2 * (size_t)s.H * s.L, gxrows = (size_t)BTP * s.T * s.S;
    size_t big = tile > wz ? tile : wz;
    const size_t encb = g_pre ? 2 * (size_t)s.H * s.L + 2 * (size_t)BTP * s.L : 0;   // the rows' latent gradients (epilogue), behind W1 | W_z
    if (encb > big) big = encb;
    const size_t zwf = g_pre ? 2 * (size_t)s.L * s.Hc : 0;   // dedicated copy of the encoder head weights (fused encoder-head backward)
    k.stage_gx = sizeof(float) * (fixed + zwf + (gxrows > big ? gxrows : big)) <= 160 * 1024 ? 1 : 0;   // dL/dx rows in LDS when they fit
    if (k.stage_gx && gxrows > big) big = gxrows;
    big = (big + 3) & ~(size_t)3;
    k.zw_off = (int)big;
    const size_t lds = sizeof(float) * (fixed + big + zwf);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (s.H == 25 && s.S == 5) {
      (void)hipFuncSetAttribute((const void*)grp::dopri5_bwd_kernel<5, 25>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      SLODE_LAUNCH("dopri5_bwd", (grp::dopri5_bwd_kernel<5, 25>), dim3(grid), dim3(BNT), lds, stream, k);
    } else if (s.H == 25 && s.S == 8) {
      (void)hipFuncSetAttribute((const void*)grp::dopri5_bwd_kernel<8, 25>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      SLODE_LAUNCH("dopri5_bwd", (grp::dopri5_bwd_kernel<8, 25>), dim3(grid), dim3(BNT), lds, stream, k);
    } else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
