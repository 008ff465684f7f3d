// Auxiliary supervised loss of the reference's second SVI object, SVI(model_meta, guide_meta) (training_cvs.py:244-249):
//   models/mechanistic_cvs.py:240-276, mechanistic_proc.py:313-359, mechanistic_challenge.py:264-297.
// The group latents are sampled IN THE MODEL (the guide is empty), so with z_g = loc_g + scale_g * eps_g
//   -ELBO_aux = - [ sum_{l in label groups} log N(z_l; loc_l, scale_l)  +  aux_mult * sum_heads log p(label_h | MLP_h(z_g)) ]
// One workgroup handles one trajectory at a time (a half-wave per label head).  Outputs dLoss/dloc, dLoss/dscale (or, on the folded
// encoder path, directly g_pre / glat: the encoder-head backward runs here as it does inside ode_elbo_kernel) and one gradient slab
// row per workgroup, reduced by the common fixed-order reduction.
#include "slode_common.h"

namespace {

struct AuxK {
  int B, L, nu, n_aux, U, Hc;
  float aux_mult;
  slode_aux aux[SLODE_MAX_AUX];
  const float *w1[SLODE_MAX_AUX], *b1[SLODE_MAX_AUX], *w2[SLODE_MAX_AUX], *b2[SLODE_MAX_AUX], *cc[SLODE_MAX_AUX];   // label-head parameters (global)
  int o_w1[SLODE_MAX_AUX], o_b1[SLODE_MAX_AUX], o_w2[SLODE_MAX_AUX], o_b2[SLODE_MAX_AUX], o_c[SLODE_MAX_AUX];      // slab index of their gradients
  int row_floats;     // floats of a slab row this kernel is responsible for (the rest of the row, if any, is zero-filled)
  const float *loc, *scale, *eps, *u;
  float *g_loc, *g_scale, *slabs;
  int slab_stride, backward;
  // fused encoder-head backward (folded encoder path): see ode_elbo_kernel
  const float *enc_hid, *enc_zloc_w, *enc_zls_w;
  float *g_pre, *glat;
  RngK rng;       // on: the group latents' noise is drawn here (Philox, slode_common.h)
  LabelSrc lab;   // n > 0: label columns from separate tensors
};

// AUX_ZMAX (template parameter): latent dims one label head may read -- the lane's row of the hidden layer and its gradient live in
// registers: 16 for every reference config (z_*_dim 1..10), 64 (= SLODE_MAX_L) for wider heads (slode_launch_aux picks)
constexpr int AUX_QMAX = 8;    // label columns of one head

// Half-wave (32 lanes) = one label head, lane j = hidden unit j; a workgroup = the n_aux heads of one trajectory at a time.  Everything a
// head needs lives in its own 32 lanes: the sums over hidden units are xor-butterflies (offsets 16..1 stay inside the half-wave), the
// head's latent sample goes through a few LDS words written and read by the same wave (in program order: no barrier), and every
// gradient element has ONE owner lane that carries it in a register across the workgroup's trajectories -- the slab row is written
// once, at the end.  The only workgroup barriers are the two around the encoder-head backward (it needs every head's latent gradient).
__device__ __forceinline__ float half_sum(float v) { return half_wave_sum(v); }   // (DPP + v_permlane16_swap: slode_common.h)

template <int AUX_ZMAX>
__global__ void __launch_bounds__(128) aux_kernel(const AuxK k) {
  __shared__ float s_z[SLODE_MAX_AUX][AUX_ZMAX];
  __shared__ float s_gl[SLODE_MAX_L], s_gs[SLODE_MAX_L];   // dLoss/dloc, dLoss/dscale * scale of the current trajectory
  __shared__ float s_red[4];
  const int tid = threadIdx.x, L = k.L, hd = tid >> 5, j = tid & 31, U = k.U;
  const bool head_on = hd < k.n_aux;
  const slode_aux ax = k.aux[head_on ? hd : 0];
  const int zd = ax.z_dim, ud = ax.u_dim;
  const bool unit_on = head_on && j < U;
  // this lane's parameters (row j of the hidden layer, column j of the output layer): registers for the whole launch
  float w1r[AUX_ZMAX], w2c[AUX_QMAX];
  float b1v = 0.f;
#pragma unroll
  for (int l = 0; l < AUX_ZMAX; ++l) w1r[l] = (unit_on && l < zd) ? k.w1[hd][j * zd + min(l, zd - 1)] : 0.f;
#pragma unroll
  for (int q = 0; q < AUX_QMAX; ++q) w2c[q] = (unit_on && q < ud) ? k.w2[hd][min(q, ud - 1) * U + j] : 0.f;
  if (unit_on) b1v = k.b1[hd][j];
  float b2v = (head_on && j < ud) ? k.b2[hd][j] : 0.f;     // lane q holds b2[q]
  const float cpar = (head_on && ax.kind == SLODE_AUX_EXPEXP) ? k.cc[hd][0] : 0.f;
  // gradient accumulators of the elements this lane owns
  float a_w1[AUX_ZMAX], a_w2[AUX_QMAX], a_b1 = 0.f, a_b2 = 0.f, a_c = 0.f;
#pragma unroll
  for (int l = 0; l < AUX_ZMAX; ++l) a_w1[l] = 0.f;
#pragma unroll
  for (int q = 0; q < AUX_QMAX; ++q) a_w2[q] = 0.f;
  float loss_acc = 0.f;
  if (tid < L) { s_gl[tid] = 0.f; s_gs[tid] = 0.f; }   // latent dims outside every head keep a zero gradient
  // encoder head weights of hidden unit mm = tid (fused encoder-head backward): in registers from the start when the latent is short
  constexpr int ZW = AUX_ZMAX > 16 ? 1 : 16;   // (the wide instantiation spends its registers on the heads' rows: weights from memory)
  const bool zw_regs = AUX_ZMAX <= 16 && k.g_pre != nullptr && L <= ZW;
  float zw0[ZW], zw1[ZW];
#pragma unroll
  for (int l = 0; l < ZW; ++l) {
    const bool on = zw_regs && tid < k.Hc && l < L;
    zw0[l] = on ? k.enc_zloc_w[min(l, L - 1) * k.Hc + min(tid, k.Hc - 1)] : 0.f;
    zw1[l] = on ? k.enc_zls_w[min(l, L - 1) * k.Hc + min(tid, k.Hc - 1)] : 0.f;
  }
  __syncthreads();

  for (int b = blockIdx.x; b < k.B; b += gridDim.x) {
    // ---- the head's latent sample z_g = loc + scale * eps (lane l' < z_dim), -log N(z_g; loc, scale) ----
    float sc = 1.f, e = 0.f;
    if (head_on && j < zd) {
      const long long i = (long long)b * L + ax.z_off + j;
      const float loc = k.loc[i];
      sc = k.scale[i]; e = slode_eps_at(k.rng, k.eps, b, L, ax.z_off + j);
      const float z = fmaf(sc, e, loc);
      const float zq = (z - loc) / sc;
      loss_acc += logf(sc) + 0.91893853320467274178f + 0.5f * zq * zq;
      s_z[hd][j] = z;
    }
    float yv = (head_on && j < ud) ? slode_label_at(k.lab, k.u, k.nu, b, ax.u_off + j) : 0.f;   // lane q holds label column q
    const float enc_hv = (k.g_pre != nullptr && k.backward && tid < k.Hc) ? k.enc_hid[(long long)b * k.Hc + tid] : 0.f;   // in flight until the end
    // ---- hidden layer (Softplus) ----
    float pre = b1v;
#pragma unroll
    for (int l = 0; l < AUX_ZMAX; ++l)
      if (l < zd) pre = fmaf(w1r[l], s_z[hd][l], pre);
    const float hv = unit_on ? softplusf(pre) : 0.f;
    float dv = unit_on ? 1.f / (1.f + expf(-pre)) : 0.f;   // softplus'
    // ---- output layer: logit[q] on every lane of the head ----
    float lg[AUX_QMAX];
#pragma unroll
    for (int q = 0; q < AUX_QMAX; ++q) {
      lg[q] = 0.f;
      if (q < ud) lg[q] = half_sum(w2c[q] * hv) + __shfl(b2v, (tid & 32) + q, 64);   // (uniform per half-wave: ud belongs to the head)
    }
    // ---- log p(label | head) and dLoss/dlogit (go[q], every lane computes the head's few outputs: no divergence, no exchange) ----
    float go[AUX_QMAX], lp = 0.f, gcst = 0.f;
    float ys[AUX_QMAX];
#pragma unroll
    for (int q = 0; q < AUX_QMAX; ++q) { go[q] = 0.f; ys[q] = __shfl(yv, (tid & 32) + q, 64); }
    if (ax.kind == SLODE_AUX_SOFTMAX) {
      float mx = -3.0e38f, ysum = 0.f, se = 0.f;
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) mx = fmaxf(mx, lg[q]);
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) { se += expf(lg[q] - mx); ysum += ys[q]; }
      const float lse = mx + logf(se);
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) {
        const float lq = lg[q] - lse;
        lp = fmaf(ys[q], lq, lp);
        go[q] = k.aux_mult * (expf(lq) * ysum - ys[q]);
      }
    } else if (ax.kind == SLODE_AUX_SIGMOID) {
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) {
        const float o = lg[q], y = ys[q];
        const float sp_pos = (o > 0.f ? o : 0.f) + log1pf(expf(-fabsf(o)));   // softplus(o), stable
        lp += y * (o - sp_pos) + (1.f - y) * (-sp_pos);
        go[q] = k.aux_mult * (1.f / (1.f + expf(-o)) - y);
      }
    } else {   // EXPEXP: Laplace(loc = exp(head 0), b = softplus(constant_std_*)); the second Exp head is unused
      const float bsc = softplusf(cpar), ib = 1.f / bsc;
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) {
        const float lc = expf(lg[q]), r = ys[q] - lc, ar = fabsf(r);
        lp += -logf(2.f * bsc) - ar * ib;
        const float sg = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
        go[q] = -k.aux_mult * sg * ib * lc;
        gcst += k.aux_mult * (ib - ar * ib * ib);
      }
      gcst = gcst / (1.f + expf(-cpar));
    }
    if (head_on && j == 0) loss_acc -= k.aux_mult * lp;
    if (k.backward) {
      // ---- back through the output layer and the Softplus; the lane's own gradient elements ----
      float gh = 0.f;
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) { gh = fmaf(w2c[q], go[q], gh); a_w2[q] = fmaf(go[q], hv, a_w2[q]); }
      dv *= gh;                                   // dLoss/d(hidden pre-activation)
      a_b1 += dv;
#pragma unroll
      for (int l = 0; l < AUX_ZMAX; ++l) if (l < zd) a_w1[l] = fmaf(dv, s_z[hd][l], a_w1[l]);
      if (j < ud) {   // lane q owns b2[q]
        float gq = 0.f;
#pragma unroll
        for (int q = 0; q < AUX_QMAX; ++q) gq = (j == q) ? go[q] : gq;
        a_b2 += gq;
      }
      if (j == 0) a_c += gcst;
      // ---- latent gradient of the head's dims: sum over hidden units, lane l' < z_dim keeps dLoss/dz_l' ----
      float gz = 0.f;
#pragma unroll
      for (int l = 0; l < AUX_ZMAX; ++l)
        if (l < zd) { const float t = half_sum(w1r[l] * dv); gz = (j == l) ? t : gz; }
      if (head_on && j < zd) {
        const int l = ax.z_off + j;
        const float gsc = fmaf(gz, e, 1.0f / sc);               // d(-log N)/dloc = 0 (z moves with loc); + d(log scale)/dscale
        if (k.g_loc) { k.g_loc[(long long)b * L + l] = gz; k.g_scale[(long long)b * L + l] = gsc; }
        s_gl[l] = gz; s_gs[l] = gsc * sc;
      }
      if (k.g_loc && tid < L) {   // dims outside every head
        bool in = false;
        for (int h2 = 0; h2 < k.n_aux; ++h2) in = in || (tid >= k.aux[h2].z_off && tid < k.aux[h2].z_off + k.aux[h2].z_dim);
        if (!in) { k.g_loc[(long long)b * L + tid] = 0.f; k.g_scale[(long long)b * L + tid] = 0.f; }
      }
      if (k.g_pre) {
        // encoder heads + tanh, backward (models/encoder_conv.py:48-51): thread <-> hidden unit of the encoder
        __syncthreads();
        if (tid < L) { k.glat[(long long)b * 128 + tid] = s_gl[tid]; k.glat[(long long)b * 128 + 64 + tid] = s_gs[tid]; }
        if (zw_regs) {
          if (tid < k.Hc) {
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int l = 0; l < ZW; ++l)
              if (l < L) { g0 = fmaf(zw0[l], s_gl[l], g0); g1 = fmaf(zw1[l], s_gs[l], g1); }
            k.g_pre[(long long)b * 64 + tid] = (g0 + g1) * (1.f - enc_hv * enc_hv);
          }
        } else {
          for (int mm = tid; mm < k.Hc; mm += blockDim.x) {
            const float hvv = k.enc_hid[(long long)b * k.Hc + mm];
            float g0 = 0.f, g1 = 0.f;
#pragma unroll 4
            for (int l = 0; l < L; ++l) {
              g0 = fmaf(k.enc_zloc_w[l * k.Hc + mm], s_gl[l], g0);
              g1 = fmaf(k.enc_zls_w[l * k.Hc + mm], s_gs[l], g1);
            }
            k.g_pre[(long long)b * 64 + mm] = (g0 + g1) * (1.f - hvv * hvv);
          }
        }
        __syncthreads();   // s_gl / s_gs are rewritten by the next trajectory
      }
    }
  }
  // ---- fixed-order workgroup sum of the loss; the slab row: [loss | label-head gradients], zeros elsewhere ----
  const float v = wave_sum(loss_acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  float* slab = k.slabs + (long long)blockIdx.x * k.slab_stride;
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < ((int)blockDim.x >> 6); ++w) t += s_red[w];
    slab[0] = t;
  }
  if (k.backward) {
    for (int i = 1 + tid; i < k.row_floats; i += blockDim.x) slab[i] = 0.f;   // entries no head owns (e.g. the unused second Exp head)
    __syncthreads();                                                            // (same workgroup: the owners' stores below come later)
    if (unit_on) {
#pragma unroll
      for (int l = 0; l < AUX_ZMAX; ++l) if (l < zd) slab[1 + k.o_w1[hd] + j * zd + l] = a_w1[l];
#pragma unroll
      for (int q = 0; q < AUX_QMAX; ++q) if (q < ud) slab[1 + k.o_w2[hd] + q * U + j] = a_w2[q];
      slab[1 + k.o_b1[hd] + j] = a_b1;
    }
    if (head_on && j < ud) slab[1 + k.o_b2[hd] + j] = a_b2;
    if (head_on && j == 0 && ax.kind == SLODE_AUX_EXPEXP) slab[1 + k.o_c[hd]] = a_c;
  }
}

}  // namespace

// compact != 0: the slab row is [loss | gradients of the flat range [aux_lo, aux_hi)] (the fused tail reduces only that range);
// else the row has the layout of the whole ODE segment, zero outside the label heads (slode_launch_reduce).
hipError_t slode_launch_aux(const AuxLaunch& a, hipStream_t stream) {
  const slode_shape& s = a.s;
  const slode_layout& lay = a.lay;
  AuxK k{};
  k.B = s.B; k.L = s.L; k.nu = s.n_u; k.n_aux = s.n_aux; k.U = s.U; k.aux_mult = s.aux_mult; k.Hc = s.Hc;
  const int base = a.compact ? lay.aux_w1[0] : lay.ode_begin;
  for (int q = 0; q < SLODE_MAX_AUX; ++q) {
    k.aux[q] = s.aux[q];
    k.w1[q] = a.params + lay.aux_w1[q]; k.b1[q] = a.params + lay.aux_b1[q]; k.w2[q] = a.params + lay.aux_w2[q];
    k.b2[q] = a.params + lay.aux_b2[q]; k.cc[q] = a.params + lay.aux_c[q];
    k.o_w1[q] = lay.aux_w1[q] - base; k.o_b1[q] = lay.aux_b1[q] - base; k.o_w2[q] = lay.aux_w2[q] - base;
    k.o_b2[q] = lay.aux_b2[q] - base; k.o_c[q] = lay.aux_c[q] - base;
  }
  k.row_floats = 1 + (a.compact ? lay.cstd - lay.aux_w1[0] : lay.ode_end - lay.ode_begin);
  k.loc = a.loc; k.scale = a.scale; k.eps = a.eps; k.u = a.u;
  k.g_loc = a.g_loc; k.g_scale = a.g_scale; k.slabs = a.slabs; k.slab_stride = a.slab_stride; k.backward = a.backward;
  k.rng = a.rng; k.lab = a.lab;
  k.enc_hid = a.enc_hid; k.enc_zloc_w = a.params + lay.zloc_w; k.enc_zls_w = a.params + lay.zls_w; k.g_pre = a.g_pre; k.glat = a.glat;
  const int nthreads = s.n_aux <= 2 ? 64 : 128;
  int zmax = 0;
  for (int q = 0; q < s.n_aux; ++q) zmax = s.aux[q].z_dim > zmax ? s.aux[q].z_dim : zmax;
  if (zmax <= 16) SLODE_LAUNCH("aux", aux_kernel<16>, dim3(a.grid), dim3(nthreads), 0, stream, k);
  else SLODE_LAUNCH("aux", aux_kernel<SLODE_MAX_L>, dim3(a.grid), dim3(nthreads), 0, stream, k);
  return hipGetLastError();
}
