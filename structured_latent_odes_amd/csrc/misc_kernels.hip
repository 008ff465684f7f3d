// Small kernels around the hot path: stage-time table, fixed-order slab reduction, decoder heads, Adam.
#include "slode_common.h"

namespace {

// Times at which the fixed-grid solvers evaluate f, with torchdiffeq's fp32 arithmetic (t0 + dt * c, separate
// multiply and add: no FMA contraction).  Call site replaced: models/blackbox_ode.py:41-45.
__global__ void stage_times_kernel(const float* __restrict__ times, float* __restrict__ out, int T, int method) {
  const int R = method == SLODE_EULER ? 1 : (method == SLODE_MIDPOINT ? 2 : 3);
  const int nt = R * (T - 1) + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (method == SLODE_DOPRI5) { if (i == 0) out[0] = times[T - 1]; return; }
  if (i >= nt) return;
  if (i == nt - 1) { out[i] = times[T - 1]; return; }
  const int n = i / R, r = i - n * R;
  const float t0 = times[n], dt = __fsub_rn(times[n + 1], t0);
  float c = 0.f;
  if (method == SLODE_MIDPOINT) c = 0.5f;
  else if (method == SLODE_RK4) c = (r == 1) ? (float)(1.0 / 3.0) : (float)(2.0 / 3.0);
  out[i] = (r == 0) ? t0 : __fadd_rn(t0, __fmul_rn(dt, c));
}

struct ReduceK {
  const float* ode_slabs; int ode_stride, ode_n, nseg, ode_begin;
  const float* small_slabs; int small_stride, small_n, small_count, n_conv_part, lin_b_off;
  const float* lin_slabs; int lin_n, lin_count, lin_w_off;
  float* grads; float* loss_out;
  int n_params, zero_rest;
  // optional fused Adam (single-process training): applied to element i right after its gradient is final
  AdamK ad;     // ad.p == nullptr: no fused Adam
  int n_total;  // >= n_params: parameters appended by the caller (zero main-loss gradient) are stepped too
};

__global__ void __launch_bounds__(256) slab_stage1_kernel(const Stage1 fam0, const Stage1 fam1) {
  __shared__ __attribute__((aligned(16))) float s_p[4 * SLODE_S1_COLS];
  const Stage1 f = blockIdx.z == 0 ? fam0 : fam1;  // two slab families reduced by one launch
  if (f.slabs == nullptr || blockIdx.x * SLODE_S1_COLS >= f.count) return;
  slab_stage1_block(f, blockIdx.x, blockIdx.y, s_p);
}

// One thread per flat-gradient element; fixed summation order over slabs => bitwise reproducible.
__global__ void reduce_kernel(const ReduceK k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && k.loss_out && k.ode_slabs) {
    double acc = 0.0;
    for (int w = 0; w < k.ode_n; ++w) acc += (double)k.ode_slabs[(long long)w * k.ode_stride];
    k.loss_out[0] = (float)acc;
  }
  if (k.grads == nullptr) return;
  const int n_all = k.ad.p ? k.n_total : k.n_params;
  if (i >= n_all) return;
  float g = 0.f;
  bool write = false;
  if (i < k.n_params) {
    // which family owns flat element i?
    int j = -1;
    if (k.small_slabs) {
      if (i < k.n_conv_part) j = i;                                                     // conv_w, conv_b
      else if (i >= k.lin_b_off && i < k.lin_b_off + (k.small_count - k.n_conv_part)) j = k.n_conv_part + (i - k.lin_b_off);
    }
    if (k.ode_slabs && i >= k.ode_begin && i < k.ode_begin + k.nseg) {
      g = strided_sum(k.ode_slabs + 1 + (i - k.ode_begin), k.ode_stride, k.ode_n);
      write = true;
    } else if (k.lin_slabs && i >= k.lin_w_off && i < k.lin_w_off + k.lin_count) {
      g = strided_sum(k.lin_slabs + (i - k.lin_w_off), k.lin_count, k.lin_n);
      write = true;
    } else if (j >= 0) {
      g = strided_sum(k.small_slabs + j, k.small_stride, k.small_n);
      write = true;
    } else if (k.zero_rest) {
      write = true;
    } else {
      g = k.grads[i];  // already final (folded path: lin.weight gradient written by the chain kernel)
    }
    if (write) k.grads[i] = g;
  }
  if (k.ad.p) adam_apply(k.ad, i, g);
}

// Decoder heads on a given trajectory tensor: models/decoders.py:45-47 (ALD: q50, q75, q25) / :86 (Gauss: mean),
// and std = softplus(constant_std) (:52-53).  mu layout [Q][B][C][T].
__global__ void decode_heads_kernel(const float* __restrict__ x, const float* __restrict__ params, int B, int T, int C, int S,
                                    int Q, int h0, int h1, int h2, int cstd_off, float* __restrict__ mu, float* __restrict__ std_ct) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)C * T && std_ct) std_ct[i] = softplusf(params[cstd_off + i]);
  const long long total = (long long)Q * B * C * T;
  if (i >= total) return;
  const int t = i % T;
  long long r = i / T;
  const int c = r % C; r /= C;
  const int b = r % B;
  const int q = r / B;
  const float* W = params + (q == 0 ? h0 : (q == 1 ? h1 : h2)) + c * S;
  const float* xs = x + ((long long)b * T + t) * S;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc = fmaf(W[s], xs[s], acc);
  mu[i] = acc;
}

// Backward of decode_heads_kernel (the materialising recon API; autograd through Decoder.forward, models/decoders.py:42-54):
//   g_x[b,t,s] = sum_{q,c} g_mu[q,b,c,t] W_q[c,s]   (thread per (b,t));   g_cstd[c,t] = g_std[c,t] sigmoid(constant_std[c,t])
__global__ void decode_heads_bwd_x_kernel(const float* __restrict__ g_mu, const float* __restrict__ g_std, const float* __restrict__ params,
                                          int B, int T, int C, int S, int Q, int h0, int h1, int h2, int cstd_off, float* __restrict__ g_x,
                                          float* __restrict__ g_cstd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g_cstd && i < (long long)C * T) g_cstd[i] = g_std ? g_std[i] / (1.f + expf(-params[cstd_off + i])) : 0.f;
  if (i >= (long long)B * T) return;
  const int t = i % T, b = i / T;
  float acc[SLODE_MAX_S];
#pragma unroll
  for (int s = 0; s < SLODE_MAX_S; ++s) acc[s] = 0.f;
  for (int q = 0; q < Q; ++q) {
    const float* W = params + (q == 0 ? h0 : (q == 1 ? h1 : h2));
    for (int c = 0; c < C; ++c) {
      const float g = g_mu[(((long long)q * B + b) * C + c) * T + t];
#pragma unroll
      for (int s = 0; s < SLODE_MAX_S; ++s)
        if (s < S) acc[s] = fmaf(g, W[c * S + s], acc[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < SLODE_MAX_S; ++s)
    if (s < S) g_x[i * S + s] = acc[s];
}
//   g_W[q][c][s] = sum_{b,t} g_mu[q,b,c,t] x[b,t,s]: one block per entry, fixed per-thread order + fixed tree => bitwise reproducible
__global__ void __launch_bounds__(256) decode_heads_bwd_w_kernel(const float* __restrict__ g_mu, const float* __restrict__ x, int B, int T,
                                                                  int C, int S, float* __restrict__ g_w) {
  __shared__ float s_red[4];
  const int e = blockIdx.x, s = e % S, c = (e / S) % C, q = e / (S * C);
  float a0 = 0.f, a1 = 0.f;
  const long long n = (long long)B * T;
  for (long long i = threadIdx.x; i < n; i += 512) {
    const long long i1 = i + 256;
    const int b0 = i / T, t0 = i - (long long)b0 * T;
    a0 = fmaf(g_mu[(((long long)q * B + b0) * C + c) * T + t0], x[i * S + s], a0);
    if (i1 < n) {
      const int b1 = i1 / T, t1 = i1 - (long long)b1 * T;
      a1 = fmaf(g_mu[(((long long)q * B + b1) * C + c) * T + t1], x[i1 * S + s], a1);
    }
  }
  const float v = block_sum(a0 + a1, s_red);
  if (threadIdx.x == 0) g_w[e] = v;
}

// One evaluation of the dynamics net, OdeFunc.forward(t, state) -> dx/dt = a(t,z) - d(t,z) * state
// (models/blackbox_ode.py:57-61,97-109).  Thread per trajectory; API-completeness kernel (the solver never calls it).
__global__ void dynamics_eval_kernel(const float* __restrict__ params, int wh, int bh, int wg, int bg, int wd, int bd, int B, int L,
                                     int S, int H, float t, const float* __restrict__ state, const float* __restrict__ z,
                                     float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float h[SLODE_MAX_H];
  for (int j = 0; j < H; ++j) {
    float pre = fmaf(params[wh + j * (1 + L)], t, params[bh + j]);
    for (int l = 0; l < L; ++l) pre = fmaf(params[wh + j * (1 + L) + 1 + l], z[(long long)b * L + l], pre);
    h[j] = fmaxf(pre, 0.f);
  }
  for (int s = 0; s < S; ++s) {
    float xa = params[bg + s], xd = params[bd + s];
    for (int j = 0; j < H; ++j) {
      xa = fmaf(params[wg + s * H + j], h[j], xa);
      xd = fmaf(params[wd + s * H + j], h[j], xd);
    }
    out[(long long)b * S + s] = sigmoidf_fast(xa) - sigmoidf_fast(xd) * state[(long long)b * S + s];
  }
}

// Eval-side small nets, thread per trajectory (the training path evaluates the same nets inside the fused kernels).
// x0 = sigmoid(W2 relu(W1 z + b1) + b2): OdeModel.initialize_state (models/blackbox_ode.py:19-22, 32-34).
__global__ void init_state_kernel(const float* __restrict__ params, int w1, int b1, int w2, int b2, int B, int L, int S, int H,
                                  const float* __restrict__ z, float* __restrict__ x0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float h[SLODE_MAX_H];
  for (int j = 0; j < H; ++j) {
    float pre = params[b1 + j];
    for (int l = 0; l < L; ++l) pre = fmaf(params[w1 + j * L + l], z[(long long)b * L + l], pre);
    h[j] = fmaxf(pre, 0.f);
  }
  for (int s = 0; s < S; ++s) {
    float o = params[b2 + s];
    for (int j = 0; j < H; ++j) o = fmaf(params[w2 + s * H + j], h[j], o);
    x0[(long long)b * S + s] = sigmoidf_fast(o);
  }
}

// Conditional priors p(z_g | u_g) = N(W_loc u_g + b_loc, exp(W_ls u_g + b_ls)) (EncoderMLP([u_dim, [z_dim, z_dim]], [None, Exp]);
// models/mechanistic_cvs.py:88-100, 225-237); latent dims outside every group: N(0, 1).  loc, scale [B, L].
struct PriorK { int n_groups; slode_group grp[SLODE_MAX_GROUPS]; int ploc_w[SLODE_MAX_GROUPS], ploc_b[SLODE_MAX_GROUPS], pls_w[SLODE_MAX_GROUPS], pls_b[SLODE_MAX_GROUPS]; };
__global__ void prior_nets_kernel(const float* __restrict__ params, const PriorK k, int B, int L, int nu, const float* __restrict__ u,
                                  float* __restrict__ loc, float* __restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i - b * L;
  float pl = 0.f, sc = 1.f;
  for (int g = 0; g < k.n_groups; ++g) {
    const slode_group gr = k.grp[g];
    if (l >= gr.z_off && l < gr.z_off + gr.z_dim) {
      const int ll = l - gr.z_off;
      float a = params[k.ploc_b[g] + ll], c = params[k.pls_b[g] + ll];
      for (int q = 0; q < gr.u_dim; ++q) {
        const float uv = u[(long long)b * nu + gr.u_off + q];
        a = fmaf(params[k.ploc_w[g] + ll * gr.u_dim + q], uv, a);
        c = fmaf(params[k.pls_w[g] + ll * gr.u_dim + q], uv, c);
      }
      pl = a; sc = expf(c);
    }
  }
  loc[i] = pl; scale[i] = sc;
}

// Label heads q(label | z_g): Linear -> Softplus -> Linear -> Sigmoid | Softmax | Exp (the first of the two Exp heads), written into the
// label columns they score: out[B, n_u]  (classifier / pred_inputs: models/mechanistic_cvs.py:278-296, mechanistic_proc.py:361-380).
struct LabelK { int n_aux, U; slode_aux aux[SLODE_MAX_AUX]; int w1[SLODE_MAX_AUX], b1[SLODE_MAX_AUX], w2[SLODE_MAX_AUX], b2[SLODE_MAX_AUX]; };
__global__ void label_heads_kernel(const float* __restrict__ params, const LabelK k, int B, int L, int nu, const float* __restrict__ z,
                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * k.n_aux) return;
  const int b = i / k.n_aux, hd = i - b * k.n_aux;
  const slode_aux ax = k.aux[hd];
  float h[32];
  for (int j = 0; j < k.U; ++j) {
    float pre = params[k.b1[hd] + j];
    for (int l = 0; l < ax.z_dim; ++l) pre = fmaf(params[k.w1[hd] + j * ax.z_dim + l], z[(long long)b * L + ax.z_off + l], pre);
    h[j] = softplusf(pre);
  }
  float o[8], mx = -3.0e38f;
  for (int q = 0; q < ax.u_dim; ++q) {
    float a = params[k.b2[hd] + q];
    for (int j = 0; j < k.U; ++j) a = fmaf(params[k.w2[hd] + q * k.U + j], h[j], a);
    o[q] = a;
    mx = fmaxf(mx, a);
  }
  float se = 0.f;
  if (ax.kind == SLODE_AUX_SOFTMAX)
    for (int q = 0; q < ax.u_dim; ++q) se += expf(o[q] - mx);
  for (int q = 0; q < ax.u_dim; ++q) {
    float v;
    if (ax.kind == SLODE_AUX_SIGMOID) v = 1.f / (1.f + expf(-o[q]));
    else if (ax.kind == SLODE_AUX_SOFTMAX) v = expf(o[q] - mx) / se;
    else v = expf(o[q]);
    out[(long long)b * nu + ax.u_off + q] = v;
  }
}

// torch.optim.Adam single-tensor update (amsgrad=False, weight_decay=0, maximize=False) as applied per parameter by
// pyro.optim.Adam (training_cvs.py:226-227): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom.
__global__ void adam_kernel(long long n, const float* __restrict__ g, const AdamK a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  adam_apply(a, (int)i, g[i]);
}

}  // namespace

hipError_t slode_launch_stage_times(const slode_shape& s, const float* times, float* stage_t, hipStream_t stream) {
  const int R = s.method == SLODE_EULER ? 1 : (s.method == SLODE_MIDPOINT ? 2 : 3);
  const int nt = R * (s.T - 1) + 1;
  hipLaunchKernelGGL(stage_times_kernel, dim3((nt + 255) / 256), dim3(256), 0, stream, times, stage_t, s.T, s.method);
  return hipGetLastError();
}

hipError_t slode_launch_reduce(const ReduceLaunch& a_in, hipStream_t stream) {
  ReduceLaunch a = a_in;
  ReduceK k{};
  const slode_shape& s = a.s;
  const int n_conv = s.T - s.K + 1, FQ = s.F * (n_conv - s.P + 1);
  // two-stage: n slabs -> SLODE_REDUCE_GROUPS partial slabs -> final (keeps every thread's serial loop short)
  Stage1 fam[2] = {{nullptr, 0, 0, 0, 0, nullptr}, {nullptr, 0, 0, 0, 0, nullptr}};
  int maxcount = 0;
  auto stage1 = [&](int slot, const float*& slabs, int stride, int& n, int count, float* part) {
    if (!slabs || !part || n <= 2 * SLODE_REDUCE_GROUPS) return;
    const int per = (n + SLODE_REDUCE_GROUPS - 1) / SLODE_REDUCE_GROUPS;
    fam[slot] = Stage1{slabs, stride, n, count, per, part};
    if (count > maxcount) maxcount = count;
    slabs = part;
    n = (n + per - 1) / per;
  };
  stage1(0, a.ode_slabs, a.ode_stride, a.ode_n, (a.lay.ode_end - a.lay.ode_begin) + 1, a.ode_part);
  const int small_count = a.folded ? slode_fold_small_count(s) : slode_enc_small_count(s);
  stage1(1, a.small_slabs, a.small_stride, a.small_n, small_count, a.small_part);
  if (maxcount > 0)
    SLODE_LAUNCH("slab_stage1", slab_stage1_kernel, dim3((maxcount + SLODE_S1_COLS - 1) / SLODE_S1_COLS, SLODE_REDUCE_GROUPS, 2), dim3(256), 0, stream, fam[0], fam[1]);
  k.ode_slabs = a.ode_slabs; k.ode_stride = a.ode_stride; k.ode_n = a.ode_n;
  k.nseg = a.lay.ode_end - a.lay.ode_begin; k.ode_begin = a.lay.ode_begin;
  k.small_slabs = a.small_slabs; k.small_stride = a.small_stride; k.small_n = a.small_n;
  k.small_count = small_count; k.n_conv_part = a.folded ? 0 : s.F * s.C * s.K + s.F; k.lin_b_off = a.lay.lin_b;
  k.lin_slabs = a.lin_slabs; k.lin_n = a.lin_n;
  k.lin_count = a.folded ? s.F * s.C * s.K + s.F : s.Hc * FQ;
  k.lin_w_off = a.folded ? a.lay.conv_w : a.lay.lin_w;
  k.grads = a.grads; k.loss_out = a.loss_out; k.n_params = a.lay.n_params; k.zero_rest = a.zero_rest;
  int n = a.grads ? a.lay.n_params : 1;
  if (a.adam_p && a.grads) {
    AdamHost ah{a.adam_p, a.adam_m, a.adam_v, a.adam_lr, a.adam_b1, a.adam_b2, a.adam_eps, a.adam_step, a.adam_n};
    ah.lo2 = a.adam_lo2; ah.hi2 = a.adam_hi2; ah.delta2 = a.adam_delta2;
    k.ad = make_adamk(&ah);
    k.n_total = a.adam_n > a.lay.n_params ? (int)a.adam_n : a.lay.n_params;
    n = k.n_total;
  }
  SLODE_LAUNCH("reduce", reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, k);
  return hipGetLastError();
}

hipError_t slode_launch_decode_heads(const slode_shape& s, const slode_layout& lay, const float* params, const float* x,
                                     float* mu, float* std_ct, hipStream_t stream) {
  const int Q = s.likelihood == SLODE_GAUSS ? 1 : 3;
  long long total = (long long)Q * s.B * s.C * s.T;
  if (total < (long long)s.C * s.T) total = (long long)s.C * s.T;
  hipLaunchKernelGGL(decode_heads_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, params, s.B, s.T,
                     s.C, s.S, Q, lay.head_w[0], lay.head_w[1], lay.head_w[2], lay.cstd, mu, std_ct);
  return hipGetLastError();
}

hipError_t slode_launch_decode_heads_bwd(const slode_shape& s, const slode_layout& lay, const float* params, const float* x, const float* g_mu,
                                         const float* g_std, float* g_x, float* g_heads, float* g_cstd, hipStream_t stream) {
  const int Q = s.likelihood == SLODE_GAUSS ? 1 : 3;
  long long n = (long long)s.B * s.T;
  if (n < (long long)s.C * s.T) n = (long long)s.C * s.T;
  hipLaunchKernelGGL(decode_heads_bwd_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g_mu, g_std, params, s.B, s.T, s.C, s.S, Q,
                     lay.head_w[0], lay.head_w[1], lay.head_w[2], lay.cstd, g_x, g_cstd);
  hipLaunchKernelGGL(decode_heads_bwd_w_kernel, dim3(Q * s.C * s.S), dim3(256), 0, stream, g_mu, x, s.B, s.T, s.C, s.S, g_heads);
  return hipGetLastError();
}

hipError_t slode_launch_dynamics_eval(const slode_shape& s, const slode_layout& lay, const float* params, float t,
                                      const float* state, const float* z, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(dynamics_eval_kernel, dim3((s.B + 63) / 64), dim3(64), 0, stream, params, lay.dyn_wh, lay.dyn_bh, lay.dyn_wg,
                     lay.dyn_bg, lay.dyn_wd, lay.dyn_bd, s.B, s.L, s.S, s.H, t, state, z, out);
  return hipGetLastError();
}

// The noise of one drawing call, materialised (slode_rng_normal): the same device function the step kernels call
__global__ void rng_fill_kernel(const RngK r, int B, int L, float* __restrict__ eps_out, unsigned int* __restrict__ raw,
                                const float* __restrict__ loc, const float* __restrict__ scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = (L + 3) >> 2;
  if (eps_out && i < (long long)B * L) {
    const long long b = i / L;
    const float e = slode_rng_normal(r, b, (int)(i - b * L));
    eps_out[i] = loc ? fmaf(scale[i], e, loc[i]) : e;   // (loc given: the sample z = loc + scale * eps, torch.normal(loc, scale))
  }
  if (raw && i < (long long)B * nb) {
    const long long b = i / nb;
    unsigned int x[4];
    philox4x32_10((unsigned int)(b + r.b0), (unsigned int)(i - b * nb), r.c2, r.c3, r.k0, r.k1, x);
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[4 * i + q] = x[q];
  }
}
hipError_t slode_launch_rng_fill(const RngK& r, int B, int L, float* eps_out, unsigned int* raw, hipStream_t stream, const float* loc,
                                 const float* scale) {
  const long long n = (long long)B * L;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, r, B, L, eps_out, raw, loc, scale);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) pack_payload_kernel(const float* __restrict__ gslabs, const float* __restrict__ gslabs_loc,
                                                           const float* __restrict__ gslabs_ls, int gsplit, int n_g, int n_h,
                                                           const float* __restrict__ ode_part, int ode_stride, int ode_n, int ode_count,
                                                           float* __restrict__ out, int o_loc, int o_ls, int o_ode, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = 0.f;
  if (i < n_g) v = strided_sum(gslabs + i, n_g, gsplit);
  else if (i >= o_loc && i < o_loc + n_h) v = strided_sum(gslabs_loc + (i - o_loc), n_h, gsplit);
  else if (i >= o_ls && i < o_ls + n_h) v = strided_sum(gslabs_ls + (i - o_ls), n_h, gsplit);
  else if (i >= o_ode && i < o_ode + ode_count) {
    if (i == o_ode) {   // the loss: the same double-precision fixed-order sum the tail uses (tail_loss)
      double acc = 0.0;
      for (int w = 0; w < ode_n; ++w) acc += (double)ode_part[(long long)w * ode_stride];
      v = (float)acc;
    } else {
      v = strided_sum(ode_part + (i - o_ode), ode_stride, ode_n);
    }
  }
  out[i] = v;
}
hipError_t slode_launch_pack_payload(const float* gslabs, const float* gslabs_loc, const float* gslabs_ls, int gsplit, int Hc, int CT, int L,
                                     const float* ode_part, int ode_stride, int ode_n, int ode_count, float* out, int o_loc, int o_ls, int o_ode,
                                     int total, hipStream_t stream) {
  SLODE_LAUNCH("pack_payload", pack_payload_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, gslabs, gslabs_loc, gslabs_ls, gsplit,
               Hc * (CT + 1), L * (Hc + 1), ode_part, ode_stride, ode_n, ode_count, out, o_loc, o_ls, o_ode, total);
  return hipGetLastError();
}

hipError_t slode_launch_adam_k(int64_t n, const float* g, const AdamHost& a, hipStream_t stream) {
  SLODE_LAUNCH("adam", adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (long long)n, g, make_adamk(&a));
  return hipGetLastError();
}

hipError_t slode_launch_adam(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                             int64_t step, hipStream_t stream) {
  const AdamHost a{p, m, v, lr, b1, b2, eps, step, n};
  return slode_launch_adam_k(n, g, a, stream);
}

hipError_t slode_launch_init_state(const slode_shape& s, const slode_layout& lay, const float* params, const float* z, float* x0, hipStream_t stream) {
  hipLaunchKernelGGL(init_state_kernel, dim3((s.B + 63) / 64), dim3(64), 0, stream, params, lay.init_w1, lay.init_b1, lay.init_w2, lay.init_b2,
                     s.B, s.L, s.S, s.H, z, x0);
  return hipGetLastError();
}

hipError_t slode_launch_prior_nets(const slode_shape& s, const slode_layout& lay, const float* params, const float* u, float* loc, float* scale,
                                   hipStream_t stream) {
  PriorK k{};
  k.n_groups = s.n_groups;
  for (int g = 0; g < s.n_groups; ++g) {
    k.grp[g] = s.groups[g];
    k.ploc_w[g] = lay.ploc_w[g]; k.ploc_b[g] = lay.ploc_b[g]; k.pls_w[g] = lay.pls_w[g]; k.pls_b[g] = lay.pls_b[g];
  }
  const int n = s.B * s.L;
  hipLaunchKernelGGL(prior_nets_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, params, k, s.B, s.L, s.n_u, u, loc, scale);
  return hipGetLastError();
}

hipError_t slode_launch_label_heads(const slode_shape& s, const slode_layout& lay, const float* params, const float* z, float* out, hipStream_t stream) {
  LabelK k{};
  k.n_aux = s.n_aux; k.U = s.U;
  for (int a = 0; a < s.n_aux; ++a) {
    k.aux[a] = s.aux[a];
    k.w1[a] = lay.aux_w1[a]; k.b1[a] = lay.aux_b1[a]; k.w2[a] = lay.aux_w2[a]; k.b2[a] = lay.aux_b2[a];
  }
  const int n = s.B * s.n_aux;
  hipLaunchKernelGGL(label_heads_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, params, k, s.B, s.L, s.n_u, z, out);
  return hipGetLastError();
}
