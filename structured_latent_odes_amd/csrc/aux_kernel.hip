// Auxiliary supervised loss of the reference's second SVI object, SVI(model_meta, guide_meta) (training_cvs.py:244-249):
//   models/mechanistic_cvs.py:240-276, mechanistic_proc.py:313-359, mechanistic_challenge.py:264-297.
// The group latents are sampled IN THE MODEL (the guide is empty), so with z_g = loc_g + scale_g * eps_g
//   -ELBO_aux = - [ sum_{l in label groups} log N(z_l; loc_l, scale_l)  +  aux_mult * sum_heads log p(label_h | MLP_h(z_g)) ]
// One workgroup handles one trajectory at a time (persistent loop); the label-head code is the same as phases P0/P7 of
// ode_elbo_kernel.  Outputs dLoss/dloc, dLoss/dscale (fed to the encoder backward kernels) and a gradient slab in the layout of
// the ODE segment (only the label-head entries are non-zero), reduced by the common fixed-order reduction.
#include "slode_common.h"

namespace {

struct AuxK {
  int B, L, nu, n_aux, U;
  float aux_mult;
  slode_aux aux[SLODE_MAX_AUX];
  int o_w1[SLODE_MAX_AUX], o_b1[SLODE_MAX_AUX], o_w2[SLODE_MAX_AUX], o_b2[SLODE_MAX_AUX], o_c[SLODE_MAX_AUX];
  int npar, nseg;
  const float *pseg, *loc, *scale, *eps, *u;
  float *g_loc, *g_scale, *slabs;
  int slab_stride, backward;
};

constexpr int ANT = 128;

__global__ void __launch_bounds__(ANT) aux_kernel(const AuxK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, L = k.L;
  float* s_par = smem;                         // [npar]
  float* s_acc = s_par + ((k.npar + 3) & ~3);  // [npar + 1]
  float* s_z = s_acc + ((k.npar + 4) & ~3);    // [L]
  float* s_eps = s_z + SLODE_MAX_L;            // [L]
  float* s_sc = s_eps + SLODE_MAX_L;           // [L]
  float* s_gz = s_sc + SLODE_MAX_L;            // [L]
  float* s_uu = s_gz + SLODE_MAX_L;            // [n_u]
  float* s_h = s_uu + SLODE_MAX_NU;            // [n_aux][32]
  float* s_d = s_h + SLODE_MAX_AUX * 32;       // [n_aux][32]
  float* s_go = s_d + SLODE_MAX_AUX * 32;      // [n_aux][12]
  float* s_red = s_go + SLODE_MAX_AUX * 12;    // [4]
  for (int i = tid; i < k.npar; i += ANT) s_par[i] = k.pseg[i];
  for (int i = tid; i < k.npar + 1; i += ANT) s_acc[i] = 0.f;
  float loss_acc = 0.f;
  __syncthreads();
  for (int b = blockIdx.x; b < k.B; b += gridDim.x) {
    if (tid < L) {
      const float loc = k.loc[(long long)b * L + tid], sc = k.scale[(long long)b * L + tid], e = k.eps[(long long)b * L + tid];
      const float z = fmaf(sc, e, loc);
      bool in_aux = false;
      for (int hd = 0; hd < k.n_aux; ++hd) in_aux = in_aux || (tid >= k.aux[hd].z_off && tid < k.aux[hd].z_off + k.aux[hd].z_dim);
      if (in_aux) {
        const float zq = (z - loc) / sc;
        loss_acc += logf(sc) + 0.91893853320467274178f + 0.5f * zq * zq;   // - log N(z; loc, scale)
      }
      s_z[tid] = z;
      s_eps[tid] = in_aux ? e : 0.f;
      s_sc[tid] = in_aux ? sc : 0.f;    // 0 marks "not a label-group dim"
      s_gz[tid] = 0.f;
    }
    if (tid < k.nu) s_uu[tid] = k.u[(long long)b * k.nu + tid];
    __syncthreads();
    {  // hidden layer (Softplus), thread (head, j)
      const int hd = tid >> 5, j = tid & 31;
      if (hd < k.n_aux && j < k.U) {
        const slode_aux ax = k.aux[hd];
        float pre = s_par[k.o_b1[hd] + j];
        for (int l = 0; l < ax.z_dim; ++l) pre = fmaf(s_par[k.o_w1[hd] + j * ax.z_dim + l], s_z[ax.z_off + l], pre);
        s_h[hd * 32 + j] = softplusf(pre);
        s_d[hd * 32 + j] = 1.f / (1.f + expf(-pre));
      }
    }
    __syncthreads();
    if (tid < k.n_aux) {  // outputs + log-prob, one thread per head (same arithmetic as ode_elbo_kernel P0c)
      const int hd = tid;
      const slode_aux ax = k.aux[hd];
      const float* w2 = s_par + k.o_w2[hd];
      const float* b2 = s_par + k.o_b2[hd];
      const float* hv = s_h + hd * 32;
      auto logit = [&](int q) {
        float o = b2[q];
        for (int j = 0; j < k.U; ++j) o = fmaf(w2[q * k.U + j], hv[j], o);
        return o;
      };
      float lp = 0.f;
      if (ax.kind == SLODE_AUX_SOFTMAX) {
        float mx = -3.0e38f, ysum = 0.f, se = 0.f;
        for (int q = 0; q < ax.u_dim; ++q) mx = fmaxf(mx, logit(q));
        for (int q = 0; q < ax.u_dim; ++q) { se += expf(logit(q) - mx); ysum += s_uu[ax.u_off + q]; }
        const float lse = mx + logf(se);
        for (int q = 0; q < ax.u_dim; ++q) {
          const float lq = logit(q) - lse, y = s_uu[ax.u_off + q];
          lp = fmaf(y, lq, lp);
          s_go[hd * 12 + q] = k.aux_mult * (expf(lq) * ysum - y);
        }
      } else if (ax.kind == SLODE_AUX_SIGMOID) {
        for (int q = 0; q < ax.u_dim; ++q) {
          const float o = logit(q), y = s_uu[ax.u_off + q];
          const float sp_pos = (o > 0.f ? o : 0.f) + log1pf(expf(-fabsf(o)));
          lp += y * (o - sp_pos) + (1.f - y) * (-sp_pos);
          s_go[hd * 12 + q] = k.aux_mult * (1.f / (1.f + expf(-o)) - y);
        }
      } else {
        const float c = s_par[k.o_c[hd]];
        const float bsc = softplusf(c), ib = 1.f / bsc;
        float gc = 0.f;
        for (int q = 0; q < ax.u_dim; ++q) {
          const float loc = expf(logit(q)), y = s_uu[ax.u_off + q];
          const float r = y - loc, ar = fabsf(r);
          lp += -logf(2.f * bsc) - ar * ib;
          const float sg = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
          s_go[hd * 12 + q] = -k.aux_mult * sg * ib * loc;
          gc += k.aux_mult * (ib - ar * ib * ib);
        }
        s_go[hd * 12 + 8] = gc / (1.f + expf(-c));
      }
      loss_acc -= k.aux_mult * lp;
    }
    if (k.backward) {
      __syncthreads();
      {
        const int hd = tid >> 5, j = tid & 31;
        if (hd < k.n_aux && j < k.U) {
          const slode_aux ax = k.aux[hd];
          float gh = 0.f;
          for (int q = 0; q < ax.u_dim; ++q) gh = fmaf(s_par[k.o_w2[hd] + q * k.U + j], s_go[hd * 12 + q], gh);
          s_d[hd * 32 + j] *= gh;
        }
      }
      __syncthreads();
      if (tid < L) {
        const int l = tid;
        float gz = 0.f;
        for (int hd = 0; hd < k.n_aux; ++hd) {
          const slode_aux ax = k.aux[hd];
          if (l >= ax.z_off && l < ax.z_off + ax.z_dim)
            for (int j = 0; j < k.U; ++j) gz = fmaf(s_par[k.o_w1[hd] + j * ax.z_dim + (l - ax.z_off)], s_d[hd * 32 + j], gz);
        }
        const float sc = s_sc[l];
        k.g_loc[(long long)b * L + l] = gz;                                       // d(-log N)/dloc = 0 (z moves with loc)
        k.g_scale[(long long)b * L + l] = (sc > 0.f) ? fmaf(gz, s_eps[l], 1.0f / sc) : 0.f;   // + d(log scale)/dscale
      }
      float* acc = s_acc + 1;
      for (int hd = 0; hd < k.n_aux; ++hd) {
        const slode_aux ax = k.aux[hd];
        for (int e = tid; e < k.U * ax.z_dim; e += ANT) {
          const int j = e / ax.z_dim, l = e - j * ax.z_dim;
          acc[k.o_w1[hd] + e] += s_d[hd * 32 + j] * s_z[ax.z_off + l];
        }
        for (int e = tid; e < ax.u_dim * k.U; e += ANT) {
          const int q = e / k.U, j = e - q * k.U;
          acc[k.o_w2[hd] + e] += s_go[hd * 12 + q] * s_h[hd * 32 + j];
        }
        if (tid < k.U) acc[k.o_b1[hd] + tid] += s_d[hd * 32 + tid];
        if (tid < ax.u_dim) acc[k.o_b2[hd] + tid] += s_go[hd * 12 + tid];
        if (tid == 0 && ax.kind == SLODE_AUX_EXPEXP) acc[k.o_c[hd]] += s_go[hd * 12 + 8];
      }
    }
    __syncthreads();
  }
  // fixed-order workgroup sum of the loss, then the slab (zeros outside the label-head entries)
  float v = wave_sum(loss_acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  float* slab = k.slabs + (long long)blockIdx.x * k.slab_stride;
  if (tid == 0) s_acc[0] = s_red[0] + s_red[1];
  __syncthreads();
  if (k.backward) {
    for (int i = tid; i < k.nseg + 1; i += ANT) slab[i] = (i < k.npar + 1) ? s_acc[i] : 0.f;
  } else if (tid == 0) {
    slab[0] = s_acc[0];
  }
}

}  // namespace

hipError_t slode_launch_aux(const AuxLaunch& a, hipStream_t stream) {
  const slode_shape& s = a.s;
  const slode_layout& lay = a.lay;
  AuxK k{};
  k.B = s.B; k.L = s.L; k.nu = s.n_u; k.n_aux = s.n_aux; k.U = s.U; k.aux_mult = s.aux_mult;
  const int ob = lay.ode_begin;
  for (int q = 0; q < SLODE_MAX_AUX; ++q) {
    k.aux[q] = s.aux[q];
    k.o_w1[q] = lay.aux_w1[q] - ob; k.o_b1[q] = lay.aux_b1[q] - ob; k.o_w2[q] = lay.aux_w2[q] - ob;
    k.o_b2[q] = lay.aux_b2[q] - ob; k.o_c[q] = lay.aux_c[q] - ob;
  }
  k.npar = lay.cstd - ob; k.nseg = lay.ode_end - ob;
  k.pseg = a.params + ob; k.loc = a.loc; k.scale = a.scale; k.eps = a.eps; k.u = a.u;
  k.g_loc = a.g_loc; k.g_scale = a.g_scale; k.slabs = a.slabs; k.slab_stride = a.slab_stride; k.backward = a.backward;
  const size_t lds = sizeof(float) * (2 * (size_t)((k.npar + 4) & ~3) + 4 * SLODE_MAX_L + SLODE_MAX_NU + 2 * SLODE_MAX_AUX * 32 +
                                      SLODE_MAX_AUX * 12 + 8);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)aux_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  SLODE_LAUNCH("aux", aux_kernel, dim3(a.grid), dim3(ANT), lds, stream, k);
  return hipGetLastError();
}
