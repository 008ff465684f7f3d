// C ABI of libslode.so (include/slode.h): validation, parameter layout, workspace carving, launch orchestration.
#include "slode_common.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";
thread_local ClockTable* g_slode_clock = nullptr;

// scope of one profiled entry point: the kernels launched inside it fill the handle's clock table from slot 0
struct ClockScope {
  explicit ClockScope(slode_handle h, bool on) {
    if (on && h->profile && h->ev_ready) { h->clk.n = 0; g_slode_clock = &h->clk; }
  }
  ~ClockScope() { g_slode_clock = nullptr; }
};

static int fail(slode_handle h, int code, const char* fmt, ...) {
  char* dst = h ? h->err : g_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  if (h) snprintf(g_err, sizeof(g_err), "%s", h->err);
  return code;
}
#define HIP_TRY(h, expr)                                                                             \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return fail(h, SLODE_EHIP, "%s: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

// Lanes per trajectory of the forward adaptive solve: sixteen while that leaves every wave a SIMD of its own (B / 4 waves <= 4 SIMDs per CU;
// the two lane groups of a trajectory share the stage evaluations: DESIGN 3.3, 94 against 101 us at B = 4096), eight beyond (the groups
// repeat the Runge-Kutta combination: more instructions in total, which is what counts once the SIMDs hold several waves).
static int dp5_lanes(const slode_ctx* h, int B) {
  if (h->dp5_w64) return h->dp5_w64;
  return (B + 3) / 4 <= 4 * h->num_cu ? 16 : 8;
}

static int stages_per_step(int method) { return method == SLODE_EULER ? 1 : (method == SLODE_MIDPOINT ? 2 : 3); }

static const char* check_shape(const slode_shape* s) {
  if (!s) return "shape is NULL";
  if (s->B < 1) return "B < 1";
  if (s->T < 2 || s->T > SLODE_MAX_T) return "T out of range [2, 1024]";
  if (s->C < 1 || s->C > SLODE_MAX_C) return "C (obs_dim) out of range [1, 4]";
  if (s->L < 1 || s->L > SLODE_MAX_L) return "L (latent dim) out of range [1, 64]";
  if (s->S < 1 || s->S > SLODE_MAX_S) return "S (ode_state_dim) out of range [1, 8]";
  if (s->H < 1 || s->H > SLODE_MAX_H) return "H (ode_hidden_dim) out of range [1, 32]";
  if (s->F < 1 || s->F > SLODE_MAX_F) return "F (n_filters) out of range [1, 16]";
  if (s->K < 1 || s->K > SLODE_MAX_K) return "K (filter_size) out of range [1, 16]";
  if (s->P < 1 || s->P > SLODE_MAX_P) return "P (pool_size) out of range [1, 8]";
  if (s->Hc < 1 || s->Hc > SLODE_MAX_HC) return "Hc (cnn_hidden_dim) out of range [1, 64]";
  if (s->T - s->K + 1 - s->P + 1 < 1) return "T too short for the conv/pool stack";
  if (s->n_u < 0 || s->n_u > SLODE_MAX_NU) return "n_u out of range [0, 16]";
  if (s->n_groups < 0 || s->n_groups > SLODE_MAX_GROUPS) return "n_groups out of range [0, 4]";
  for (int g = 0; g < s->n_groups; ++g) {
    const slode_group& gr = s->groups[g];
    if (gr.z_off < 0 || gr.z_dim < 1 || gr.z_off + gr.z_dim > s->L) return "prior group latent range outside [0, L)";
    if (gr.u_off < 0 || gr.u_dim < 1 || gr.u_off + gr.u_dim > s->n_u) return "prior group label range outside [0, n_u)";
    for (int g2 = 0; g2 < g; ++g2) {
      const slode_group& o = s->groups[g2];
      if (gr.z_off < o.z_off + o.z_dim && o.z_off < gr.z_off + gr.z_dim) return "prior groups overlap";
    }
  }
  if (s->method < SLODE_EULER || s->method > SLODE_DOPRI5) return "unknown method";
  if (s->likelihood != SLODE_ALD && s->likelihood != SLODE_GAUSS) return "likelihood must be ALD or GAUSS";
  if (s->n_aux < 0 || s->n_aux > SLODE_MAX_AUX) return "n_aux out of range [0, 4]";
  if (s->n_aux > 0 && (s->U < 1 || s->U > 32)) return "U (u_hidden_dim) out of range [1, 32]";
  for (int a = 0; a < s->n_aux; ++a) {
    const slode_aux& x = s->aux[a];
    if (x.kind < SLODE_AUX_SIGMOID || x.kind > SLODE_AUX_EXPEXP) return "unknown aux head kind";
    if (x.z_off < 0 || x.z_dim < 1 || x.z_off + x.z_dim > s->L) return "aux head latent range outside [0, L)";
    if (x.u_off < 0 || x.u_dim < 1 || x.u_dim > 8 || x.u_off + x.u_dim > s->n_u) return "aux head label range outside [0, n_u) or wider than 8";
  }
  if (s->grad_mode != SLODE_GRAD_EXACT && s->grad_mode != SLODE_GRAD_REFERENCE_ADJOINT) return "grad_mode must be SLODE_GRAD_EXACT or SLODE_GRAD_REFERENCE_ADJOINT";
  return nullptr;
}

extern "C" {

int slode_version(void) { return SLODE_VERSION; }

const char* slode_last_error(slode_handle h) { return h ? h->err : g_err; }

int slode_create(slode_handle* out, int device_id) {
  if (!out) return fail(nullptr, SLODE_EINVAL, "handle pointer is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, SLODE_EHIP, "no HIP device visible (%s); libslode has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device_id < 0 || device_id >= n) return fail(nullptr, SLODE_EINVAL, "device_id %d outside [0, %d)", device_id, n);
  hipDeviceProp_t prop;
  HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SLODE_EHIP, "device %d is %s; libslode is built for gfx950 only", device_id, prop.gcnArchName);
  slode_ctx* c = new slode_ctx();
  c->device = device_id;
  c->num_cu = prop.multiProcessorCount;
  c->err[0] = 0;
  c->profile = 0; c->ev_ready = 0; c->clk.n = 0;
  c->adam_lo2 = c->adam_hi2 = 0; c->adam_delta2 = 0;
  c->rng_seed = 0; c->rng_counter = 0; c->rng_b0 = 0;
  // the in-launch fold is a measured arm, off by default: it removes the 5.5 us fold launch and adds 6.0 us to the chain launch
  // (profiles/r04_c_ab*_fold_next_*.log)
  c->fold_on = getenv("SLODE_FOLD_NEXT") ? atoi(getenv("SLODE_FOLD_NEXT")) : 0;
  c->fold_valid = 0; c->fold_tmajor = 0; c->fold_ws = nullptr; c->fold_params = nullptr; c->fold_gen = 0;
  // (measured arms: 16 / 32 / 64 lanes per trajectory give the same bits and the same time, profiles/r04_f_ab11_*: the shipped form stays 8)
  c->dp5_w64 = getenv("SLODE_DP5_LPT") ? atoi(getenv("SLODE_DP5_LPT")) : 0;   // 0: chosen per batch (dp5_lanes)
  c->chain_resident = 0; memset(c->chain_resident_sig, 0, sizeof(c->chain_resident_sig));
  // diagnostics and test hooks: the environment is read here, once per handle, never at launch time
  c->no_fold = getenv("SLODE_NO_FOLD") != nullptr;   // force the layer-by-layer encoder kernels
  c->ode_loop = getenv("SLODE_ODE_LOOP") != nullptr;
  c->ode_generic = getenv("SLODE_ODE_GENERIC") != nullptr;
  c->ode_alg = getenv("SLODE_ODE_ALG") ? atoi(getenv("SLODE_ODE_ALG")) : 0;
  c->enc_fuse = getenv("SLODE_ENC_FUSE") ? atoi(getenv("SLODE_ENC_FUSE")) : 1;
  c->ode_pack = getenv("SLODE_ODE_PACK") ? atoi(getenv("SLODE_ODE_PACK")) : 0;
  c->ode_grid_cap = getenv("SLODE_ODE_GRID") ? atoi(getenv("SLODE_ODE_GRID")) : 0;
  *out = c;
  return SLODE_OK;
}

int slode_destroy(slode_handle h) {
  if (h && h->ev_ready)
    for (int i = 0; i < SLODE_CLOCK_MAX; ++i) { (void)hipEventDestroy(h->clk.ev[i][0]); (void)hipEventDestroy(h->clk.ev[i][1]); }
  delete h;
  return SLODE_OK;
}

int slode_layout_init(const slode_shape* s, slode_layout* lay) {
  const char* why = check_shape(s);
  if (why) return fail(nullptr, SLODE_EINVAL, "%s", why);
  if (!lay) return fail(nullptr, SLODE_EINVAL, "layout pointer is NULL");
  memset(lay, 0, sizeof(*lay));
  const int n_conv = s->T - s->K + 1, FQ = s->F * (n_conv - s->P + 1);
  const int Q = s->likelihood == SLODE_GAUSS ? 1 : 3;
  int o = 0;
  lay->conv_w = o; o += s->F * s->C * s->K;
  lay->conv_b = o; o += s->F;
  lay->lin_w = o; o += s->Hc * FQ;
  lay->lin_b = o; o += s->Hc;
  lay->zloc_w = o; o += s->L * s->Hc;
  lay->zloc_b = o; o += s->L;
  lay->zls_w = o; o += s->L * s->Hc;
  lay->zls_b = o; o += s->L;
  lay->ode_begin = o;
  for (int g = 0; g < s->n_groups; ++g) {
    const slode_group& gr = s->groups[g];
    lay->ploc_w[g] = o; o += gr.z_dim * gr.u_dim;
    lay->ploc_b[g] = o; o += gr.z_dim;
    lay->pls_w[g] = o; o += gr.z_dim * gr.u_dim;
    lay->pls_b[g] = o; o += gr.z_dim;
  }
  lay->init_w1 = o; o += s->H * s->L;
  lay->init_b1 = o; o += s->H;
  lay->init_w2 = o; o += s->S * s->H;
  lay->init_b2 = o; o += s->S;
  lay->dyn_wh = o; o += s->H * (1 + s->L);
  lay->dyn_bh = o; o += s->H;
  lay->dyn_wg = o; o += s->S * s->H;
  lay->dyn_bg = o; o += s->S;
  lay->dyn_wd = o; o += s->S * s->H;
  lay->dyn_bd = o; o += s->S;
  for (int q = 0; q < SLODE_MAX_HEADS; ++q) {
    lay->head_w[q] = o;
    if (q < Q) o += s->C * s->S;
  }
  for (int a = 0; a < s->n_aux; ++a) {
    const slode_aux& x = s->aux[a];
    lay->aux_w1[a] = o; o += s->U * x.z_dim;
    lay->aux_b1[a] = o; o += s->U;
    lay->aux_w2[a] = o; o += x.u_dim * s->U;
    lay->aux_b2[a] = o; o += x.u_dim;
    if (x.kind == SLODE_AUX_EXPEXP) {
      lay->aux_w3[a] = o; o += x.u_dim * s->U;
      lay->aux_b3[a] = o; o += x.u_dim;
      lay->aux_c[a] = o; o += 1;
    }
  }
  lay->cstd = o; o += s->C * s->T;
  lay->ode_end = o;
  lay->n_params = o;
  return SLODE_OK;
}

int slode_num_stage_times(const slode_shape* s) {
  if (!s || s->T < 2) return SLODE_EINVAL;
  if (s->method == SLODE_DOPRI5) return 1;  // adaptive: no table (a 1-element dummy keeps callers uniform)
  return stages_per_step(s->method) * (s->T - 1) + 1;
}

}  // extern "C"

// ---- workspace carving -------------------------------------------------------------------------------------
struct Workspace {
  float *loc, *scale, *pooled, *hid, *g_loc, *g_scale, *g_pre, *ode_slabs, *ode_part, *small_slabs, *small_part, *lin_slabs;
  float *weff, *rowsum, *wprime, *beff, *gslabs, *conv_slabs, *glat, *gslabs2, *gslabs3;  // folded encoder path
  unsigned int* counter;
  // dopri5 training: solution, dLoss/dx, external latent gradient, latent sample, step records
  float *dp_x, *dp_gx, *dp_gz, *dp_z, *dp_rec, *dp_snap, *dp_eps, *dp_tabs;
  float* sigtab;   // [4][C*T] likelihood-scale table of the step (OdeLaunch::sigtab)
  int* dp_nrec;
  int dp_kmax, dp_rows;
  int gsplit;
  int ode_grid, ode_stride, small_grid, small_stride, lin_splitk;
  size_t bytes;
};

static int ode_grid_for(slode_handle h, const slode_shape& s) {
  const int nthreads = slode_ode_threads(s);
  const size_t lds = slode_ode_lds_bytes(s, nthreads);
  int occ = lds ? (int)((160 * 1024) / lds) : 1;
  const int by_waves = 32 / (nthreads / 64);
  if (occ > by_waves) occ = by_waves;
  if (occ > 8) occ = 8;
  if (occ < 1) occ = 1;
  const int cus = h ? h->num_cu : 256;
  long long g = (long long)cus * occ;
  // One workgroup per trajectory up to 65,536 trajectories (the hardware queues the workgroups; one slab per trajectory);
  // beyond that (and under the SLODE_ODE_LOOP handle flag, which the persistent-loop tests set) a resident grid loops over them.
  if (s.B <= 65536 && !(h && h->ode_loop)) g = s.B;
  else if (h && h->ode_grid_cap > 0 && g > h->ode_grid_cap) g = h->ode_grid_cap;
  if (g > s.B) g = s.B;
  return (int)g;
}

// the Philox key / counter words of drawing call n on this handle (slode_common.h: RngK)
static RngK rng_of(const slode_ctx* h, uint64_t n) {
  RngK r{};
  r.k0 = (unsigned int)h->rng_seed; r.k1 = (unsigned int)(h->rng_seed >> 32);
  r.c2 = (unsigned int)n; r.c3 = (unsigned int)(n >> 32);
  r.b0 = h->rng_b0; r.on = 1;
  return r;
}

static size_t align_up(size_t v) { return (v + 63) & ~(size_t)63; }  // in floats: 256-byte alignment

// dopri5 training runs the fixed-grid ELBO kernel as the scorer of the adaptive solution (explicit Euler on the data grid as its
// placeholder solver: nothing flows through it) -- the shape that kernel, the grid and the workspace are sized for
static slode_shape scorer_shape(const slode_shape& s) {
  slode_shape e = s;
  if (s.method == SLODE_DOPRI5) { e.method = SLODE_EULER; e.grad_mode = SLODE_GRAD_EXACT; }
  return e;
}

static Workspace carve(slode_handle h, const slode_shape& s_in, const slode_layout& lay, void* base) {
  Workspace w{};
  const slode_shape s = scorer_shape(s_in);
  const bool dp5 = s_in.method == SLODE_DOPRI5;
  w.dp_rows = dp5 ? slode_dopri5_rows(s) : 0;
  const int n_conv = s.T - s.K + 1, FQ = s.F * (n_conv - s.P + 1);
  w.ode_grid = ode_grid_for(h, s);
  w.ode_stride = (int)align_up((size_t)(lay.ode_end - lay.ode_begin) + 1);
  w.small_grid = slode_enc_bwd_grid(s);
  w.small_stride = (int)align_up((size_t)slode_enc_small_count(s));
  w.lin_splitk = slode_enc_lin_splitk(s);
  size_t o = 0;
  float* b = (float*)base;
  auto take = [&](size_t n) { float* p = b ? b + o : nullptr; o += align_up(n); return p; };
  w.loc = take((size_t)s.B * s.L);
  w.scale = take((size_t)s.B * s.L);
  w.pooled = take((size_t)s.B * FQ);
  w.hid = take((size_t)s.B * s.Hc);
  w.g_loc = take((size_t)s.B * s.L);
  w.g_scale = take((size_t)s.B * s.L);
  w.g_pre = take((size_t)s.B * 64);
  w.ode_slabs = take((size_t)(w.ode_grid + w.dp_rows) * w.ode_stride);   // dopri5: its backward kernel's rows follow the scorer's
  w.ode_part = take((size_t)SLODE_REDUCE_GROUPS * w.ode_stride);
  w.small_slabs = take((size_t)w.small_grid * w.small_stride);
  w.small_part = take((size_t)SLODE_REDUCE_GROUPS * w.small_stride);
  w.lin_slabs = take((size_t)w.lin_splitk * s.Hc * FQ);
  w.gsplit = w.lin_splitk;
  w.weff = take((size_t)s.Hc * s.C * s.T);
  w.rowsum = take((size_t)s.Hc * s.F);
  w.wprime = take((size_t)s.F * s.C * (s.K + s.P));
  w.beff = take(64);
  w.gslabs = take((size_t)w.gsplit * s.Hc * (s.C * s.T + 1));
  w.conv_slabs = take((size_t)s.Hc * (s.F * s.C * s.K + s.F));
  w.glat = take((size_t)s.B * 128);
  w.gslabs2 = take((size_t)w.gsplit * s.L * (s.Hc + 1));
  w.gslabs3 = take((size_t)w.gsplit * s.L * (s.Hc + 1));
  w.counter = reinterpret_cast<unsigned int*>(take(32 * (16 + SLODE_MAX_HC)));   // arrival counters 128 B apart: one per conv-filter pair (0..7), the in-launch fold's (8, 9, 16 + m)
  w.sigtab = take(4 * (size_t)s.C * s.T);
  if (dp5) {
    w.dp_kmax = slode_dopri5_kmax(s);
    w.dp_x = take((size_t)s.B * s.T * s.S);
    w.dp_gx = take((size_t)s.B * s.T * s.S);
    w.dp_gz = take((size_t)s.B * s.L);
    w.dp_z = take((size_t)s.B * s.L);
    w.dp_nrec = reinterpret_cast<int*>(take((size_t)s.B));
    w.dp_rec = take((size_t)w.dp_kmax * s.B * (s.S + 2));
    w.dp_tabs = take((size_t)slode_dopri5_rows(s) * slode_dopri5_tab_floats(s));   // the forward kernel's set-up tables, handed to the reverse sweep
    w.dp_snap = take((size_t)s.B * 2 * s.H * 4 * s.S);   // running sums parked at the hidden units' switching times, per lane group (dopri5_kernel.hip)
    w.dp_eps = take((size_t)s.B * s.L);              // the noise the forward kernel drew (eps == NULL), for the scorer and the reverse sweep
  }
  w.bytes = o * sizeof(float);
  return w;
}

// Data-parallel payload (slode_grad_partial -> all-reduce -> slode_grad_apply): everything the chain rule + tail need of the batch,
// [G = g_pre^T [X | 1]: Hc x (CT + 1)] [glat_loc^T [hid | 1]: L x (Hc + 1)] [glat_ls^T [hid | 1]: L x (Hc + 1)] [loss | ODE-half row],
// each piece starting on a 16-byte boundary.  The chain rule is linear in G, so reducing G over the ranks and chain-ruling once gives the
// gradient of the global batch: 34 k floats instead of the 96 k of the flat gradient at the metric shape.
struct PayloadMap { int g_loc, g_ls, ode, total; };
static PayloadMap payload_map(const slode_shape& s, int part_floats) {
  auto a4 = [](int v) { return (v + 3) & ~3; };
  PayloadMap m;
  m.g_loc = a4(s.Hc * (s.C * s.T + 1));
  m.g_ls = m.g_loc + a4(s.L * (s.Hc + 1));
  m.ode = m.g_ls + a4(s.L * (s.Hc + 1));
  m.total = m.ode + a4(part_floats + 1);
  return m;
}

static int batch_labels(slode_handle h, const slode_shape* s, const slode_batch* batch, LabelSrc* lab) {
  if (batch->n_labels < 0 || batch->n_labels > SLODE_MAX_LABELS) return fail(h, SLODE_EINVAL, "n_labels out of range [0, %d]", SLODE_MAX_LABELS);
  lab->n = batch->n_labels;
  int cols = 0;
  for (int i = 0; i < batch->n_labels; ++i) {
    if (!batch->labels[i] || batch->label_width[i] < 1) return fail(h, SLODE_EINVAL, "label tensor %d is NULL or has width < 1", i);
    lab->p[i] = batch->labels[i]; lab->off[i] = cols; cols += batch->label_width[i];
  }
  for (int i = batch->n_labels; i <= SLODE_MAX_LABELS; ++i) lab->off[i] = cols;
  if (batch->n_labels > 0 && cols != s->n_u)
    return fail(h, SLODE_EINVAL, "the label tensors have %d columns in all, the shape's n_u is %d", cols, s->n_u);
  return SLODE_OK;
}

static const char* check_common(slode_handle h, const slode_shape* s, const slode_layout* lay, const void* params) {
  if (!h) return "handle is NULL";
  const char* why = check_shape(s);
  if (why) return why;
  if (!lay) return "layout is NULL";
  if (!params) return "params is NULL";
  return nullptr;
}

extern "C" {

size_t slode_workspace_bytes(slode_handle h, const slode_shape* s) {
  slode_layout lay;
  if (slode_layout_init(s, &lay) != SLODE_OK) return 0;
  return carve(h, *s, lay, nullptr).bytes;
}

int slode_stage_times(slode_handle h, const slode_shape* s, const float* times, float* stage_t, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  const char* why = check_shape(s);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!times || !stage_t) return fail(h, SLODE_EINVAL, "times / stage_t is NULL");
  HIP_TRY(h, slode_launch_stage_times(*s, times, stage_t, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_encoder_conv_fwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                           const float* obs, const int64_t obs_strides[3], float* loc, float* scale, float* pooled,
                           float* hid, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!obs || !obs_strides || !loc || !scale) return fail(h, SLODE_EINVAL, "obs / obs_strides / loc / scale is NULL");
  EncLaunch a{*s, *lay, params, obs, obs_strides[0], obs_strides[1], obs_strides[2], loc, scale, pooled, hid};
  hipError_t e = slode_launch_enc_fwd(a, (hipStream_t)stream);
  if (e == hipErrorInvalidValue)
    return fail(h, SLODE_EINVAL, "encoder kernels are instantiated for (obs_dim, filter_size) in {(3,10),(4,10)} and need "
                                 "the tile to fit 160 KiB of LDS; got C=%d K=%d T=%d", s->C, s->K, s->T);
  HIP_TRY(h, e);
  return SLODE_OK;
}

int slode_encoder_conv_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                           const float* obs, const int64_t obs_strides[3], const float* scale, const float* pooled,
                           const float* hid, const float* g_loc, const float* g_scale, float* grads, void* workspace,
                           size_t workspace_bytes, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!obs || !obs_strides || !scale || !pooled || !hid || !g_loc || !g_scale || !grads || !workspace)
    return fail(h, SLODE_EINVAL, "a required pointer is NULL");
  Workspace w = carve(h, *s, *lay, workspace);
  if (workspace_bytes < w.bytes) return fail(h, SLODE_ENOSPC, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  EncBwdLaunch a{*s, *lay, params, obs, obs_strides[0], obs_strides[1], obs_strides[2], scale, pooled, hid, g_loc, g_scale,
                 w.g_pre, w.small_slabs, w.small_stride, w.small_grid, w.lin_slabs, w.lin_splitk};
  hipError_t e = slode_launch_enc_bwd(a, (hipStream_t)stream);
  if (e == hipErrorInvalidValue) return fail(h, SLODE_EINVAL, "unsupported encoder shape C=%d K=%d T=%d", s->C, s->K, s->T);
  HIP_TRY(h, e);
  ReduceLaunch r{*s, *lay, nullptr, 0, 0, w.small_slabs, w.small_stride, w.small_grid, w.lin_slabs, w.lin_splitk, grads, nullptr, 0, nullptr, w.small_part, 0};
  HIP_TRY(h, slode_launch_reduce(r, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_ode_solve_fwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                        const float* times, const float* stage_t, const float* z, float* x, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!times || !z || !x) return fail(h, SLODE_EINVAL, "times / z / x is NULL");
  if (s->method == SLODE_DOPRI5) {  // adaptive solve: per-trajectory controller, no stage-time table
    DopriRec plain{};   // (no guide sample, no records: a bare solve) -- carries the handle's choice of forward kernel
    plain.w64 = dp5_lanes(h, s->B);
    hipError_t e5 = slode_launch_dopri5(*s, *lay, params, times, z, x, (hipStream_t)stream, &plain);
    if (e5 == hipErrorInvalidValue) return fail(h, SLODE_EINVAL, "dopri5 kernel is instantiated for (S,H) in {(5,25),(8,25)}");
    HIP_TRY(h, e5);
    return SLODE_OK;
  }
  if (!stage_t) return fail(h, SLODE_EINVAL, "stage_t is NULL");
  const int grid = ode_grid_for(h, *s);
  OdeLaunch a{};
  a.s = *s; a.lay = *lay; a.params = params; a.times = times; a.stage_t = stage_t; a.z_in = z; a.x_out = x;
  a.slabs = nullptr; a.slab_stride = 0; a.grid = grid; a.backward = 0; a.with_ll = 0;   // pure solve: no loss slot, nothing allocated
  a.force_loop = h->ode_loop; a.force_generic = h->ode_generic;
  hipError_t e = slode_launch_ode(a, (hipStream_t)stream, h->err, sizeof(h->err));
  if (e == hipErrorInvalidValue) return SLODE_EINVAL;
  HIP_TRY(h, e);
  return SLODE_OK;
}

int slode_ode_solve_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                        const float* times, const float* stage_t, const float* z, const float* g_x, float* g_z,
                        float* grads, void* workspace, size_t workspace_bytes, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!times || !stage_t || !z || !g_x || !g_z || !grads || !workspace) return fail(h, SLODE_EINVAL, "a required pointer is NULL");
  if (s->method == SLODE_DOPRI5) return fail(h, SLODE_EINVAL, "dopri5 is forward-only (slode_ode_solve_fwd); gradients need a fixed-grid method");
  Workspace w = carve(h, *s, *lay, workspace);
  if (workspace_bytes < w.bytes) return fail(h, SLODE_ENOSPC, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  OdeLaunch a{};
  a.s = *s; a.lay = *lay; a.params = params; a.times = times; a.stage_t = stage_t; a.z_in = z; a.gx_in = g_x;
  a.g_loc = g_z; a.slabs = w.ode_slabs; a.slab_stride = w.ode_stride; a.grid = w.ode_grid; a.backward = 1; a.with_ll = 0;
  a.force_loop = h->ode_loop; a.force_generic = h->ode_generic;
  hipError_t e = slode_launch_ode(a, (hipStream_t)stream, h->err, sizeof(h->err));
  if (e == hipErrorInvalidValue) return SLODE_EINVAL;
  HIP_TRY(h, e);
  ReduceLaunch r{*s, *lay, w.ode_slabs, w.ode_stride, w.ode_grid, nullptr, 0, 0, nullptr, 0, grads, nullptr, 0, w.ode_part, nullptr, 0};
  HIP_TRY(h, slode_launch_reduce(r, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_decode_heads(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* x,
                       float* mu, float* std_ct, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!x || !mu) return fail(h, SLODE_EINVAL, "x / mu is NULL");
  HIP_TRY(h, slode_launch_decode_heads(*s, *lay, params, x, mu, std_ct, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_decode_heads_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* x,
                           const float* g_mu, const float* g_std, float* g_x, float* g_heads, float* g_cstd, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!x || !g_mu || !g_x || !g_heads) return fail(h, SLODE_EINVAL, "x / g_mu / g_x / g_heads is NULL");
  HIP_TRY(h, slode_launch_decode_heads_bwd(*s, *lay, params, x, g_mu, g_std, g_x, g_heads, g_cstd, (hipStream_t)stream));
  return SLODE_OK;
}

struct AdamArgs { float *p, *m, *v; float lr, b1, b2, eps; int64_t step, n; };

static int elbo_step_impl(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* times,
                          const float* stage_t, const float* obs, const int64_t obs_strides[3], const float* u, const float* eps,
                          float* loss_out, float* grads, float* x_out, float* z_out, void* workspace, size_t workspace_bytes,
                          void* stream, const AdamArgs* adam, int aux_mode = 0, const LabelSrc* labels = nullptr, int phase = 0,
                          float* payload = nullptr) {
  // phase 0: the whole step.  Data parallel with the small payload (slode_grad_partial / slode_grad_apply): phase 1 = everything up to the
  // split-K products, then the partials packed into `payload` = [G | G_loc | G_ls | loss | ODE-half row]; phase 2 = chain rule + tail
  // (+ Adam) from the (all-reduced) payload.
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (phase == 2) {
    if (!obs_strides || !payload || !grads || !workspace) return fail(h, SLODE_EINVAL, "a required pointer is NULL");
  } else if ((!aux_mode && (!times || !stage_t)) || !obs || !obs_strides || (!loss_out && phase == 0) || !workspace || (phase == 1 && !payload))
    return fail(h, SLODE_EINVAL, "a required pointer is NULL");
  // eps == NULL: this call draws the guide's noise inside its kernels -- call number rng_counter of the handle's Philox stream
  RngK rng{};
  if (!eps && phase != 2) rng = rng_of(h, h->rng_counter++);
  LabelSrc lab{};
  if (labels) lab = *labels;
  if (lab.n > 0) u = u ? u : lab.p[0];   // (non-null = "labels present"; the kernels read through the accessor)
  if (aux_mode && (s->n_aux < 1 || (!u && phase != 2))) return fail(h, SLODE_EINVAL, "the auxiliary loss needs label heads (n_aux >= 1) and labels u");
  if (aux_mode) {
    // aux_kernel only: every head owns the latent-gradient slots of the dims it reads (one writer per slot).  The reference's heads read
    // disjoint groups (z_iext / z_rtpr, z_aR / z_aS / z_C12 / z_C6, ...); the main step, the solves and the eval-side entry points
    // have no such limit and are not affected.  Heads wider than 16 dims take the kernel's wide instantiation.
    for (int a = 0; a < s->n_aux; ++a)
      for (int a2 = 0; a2 < a; ++a2) {
        const slode_aux &x = s->aux[a], &o = s->aux[a2];
        if (x.z_off < o.z_off + o.z_dim && o.z_off < x.z_off + x.z_dim)
          return fail(h, SLODE_EINVAL, "slode_aux_step: label heads %d and %d read overlapping latent ranges", a2, a);
      }
  }
  if (s->n_groups > 0 && !u && phase != 2) return fail(h, SLODE_EINVAL, "u is NULL but the shape has conditional prior groups");
  const bool dp5 = !aux_mode && s->method == SLODE_DOPRI5;
  if (dp5 && !(s->H == 25 && (s->S == 5 || s->S == 8)))
    return fail(h, SLODE_EINVAL, "dopri5 kernels are instantiated for (S,H) in {(5,25),(8,25)}");
  if (dp5 && (s->B > 65536 || h->ode_loop))
    return fail(h, SLODE_EINVAL, "the dopri5 ELBO step takes at most 65,536 trajectories per call");
  Workspace w = carve(h, *s, *lay, workspace);
  if (workspace_bytes < w.bytes) return fail(h, SLODE_ENOSPC, "workspace %zu B < required %zu B", workspace_bytes, w.bytes);
  hipStream_t st = (hipStream_t)stream;
  const bool bwd = grads != nullptr || phase == 1;
  ClockScope clock_scope(h, true);

  // Folded encoder (encoder_fused.hip) when every trajectory's C*T observations are one dense block; else layer by layer.
  const long long CT = (long long)s->C * s->T;
  const bool t_major = obs_strides[1] == 1 && obs_strides[2] == s->C;          // [B,T,C] contiguous (cvs / challenge batches)
  const bool c_major = obs_strides[2] == 1 && obs_strides[1] == s->T;          // [B,C,T] contiguous (proc batches)
  const bool folded = !h->no_fold && obs_strides[0] == CT && (t_major || c_major) && (s->C == 3 || s->C == 4);
  FoldLaunch fl{};
  hipError_t e;
  bool enc_fused = false;
  if (phase != 0 && !folded)
    return fail(h, SLODE_EINVAL, "slode_grad_partial / slode_grad_apply need the folded encoder path (dense [B,T,C] or [B,C,T] observations, C in {3,4})");
  if (phase == 2) {   // no forward work: the fold launch of phase 1 left w' / rowsum / the zeroed arrival counters in this workspace
    fl.s = *s; fl.lay = *lay; fl.params = params; fl.x = obs; fl.t_major = t_major ? 1 : 0;
    fl.weff = w.weff; fl.rowsum = w.rowsum; fl.wprime = w.wprime; fl.beff = w.beff; fl.loc = w.loc; fl.scale = w.scale; fl.hid = w.hid;
    fl.g_loc = w.g_loc; fl.g_scale = w.g_scale; fl.g_pre = w.g_pre; fl.small_slabs = w.small_slabs; fl.small_stride = w.small_stride;
    fl.g_lin_w = grads + lay->lin_w; fl.conv_slabs = w.conv_slabs; fl.counter = w.counter;
  } else if (folded) {
    fl.s = *s; fl.lay = *lay; fl.params = params; fl.x = obs; fl.t_major = t_major ? 1 : 0;
    fl.weff = w.weff; fl.rowsum = w.rowsum; fl.wprime = w.wprime; fl.beff = w.beff; fl.loc = w.loc; fl.scale = w.scale; fl.hid = w.hid;
    fl.g_loc = w.g_loc; fl.g_scale = w.g_scale; fl.g_pre = w.g_pre; fl.small_slabs = w.small_slabs; fl.small_stride = w.small_stride;
    fl.gslabs = w.gslabs; fl.n_gslabs = w.gsplit; fl.g_lin_w = grads ? grads + lay->lin_w : nullptr; fl.conv_slabs = w.conv_slabs;
    fl.counter = w.counter;
    fl.sigtab = w.sigtab;   // (the auxiliary step does not read it, but the fold it leaves in the workspace serves the next main step too)
    // the loop-free ODE kernel of the metric shape runs the encoder forward of its own trajectories (ode_kernel.hip, ENCF): fold only
    enc_fused = !aux_mode && !dp5 && bwd && h->enc_fuse && !h->ode_loop && !h->ode_generic && h->ode_alg == 0 && (h->ode_pack == 0 || h->ode_pack >= 10) && !x_out &&
                slode_ode_can_fuse_encoder(*s, bwd, w.ode_grid);
    fl.skip_enc = enc_fused ? 1 : 0;
    // The previous weight-updating step on this (workspace, params) left W_eff / b_eff / rowsum / w' / the likelihood-scale table of the
    // CURRENT weights behind (enc_chain_kernel, FOLD-NEXT): no fold launch then.  Anything else -- first step, another workspace, weights
    // changed outside this handle (slode_fold_invalidate) -- folds here, which also zeroes the arrival counters.
    const bool have_fold = h->fold_on && h->fold_valid && h->fold_ws == workspace && h->fold_params == (const void*)params &&
                           h->fold_tmajor == (t_major ? 1 : 0);
    if (have_fold && enc_fused) {
      // (nothing to launch)
    } else if (have_fold) {
      fl.fold_skip = 1;                  // the encoder forward launch alone
      e = slode_launch_fold_fwd(fl, st);
      HIP_TRY(h, e);
    } else {
      e = slode_launch_fold_fwd(fl, st);
      HIP_TRY(h, e);
      h->fold_gen = 0;                   // (the fold launch zeroed the in-launch fold's arrival counter)
      h->fold_valid = 1; h->fold_ws = workspace; h->fold_params = params; h->fold_tmajor = t_major ? 1 : 0;
    }
  } else {
    EncLaunch ef{*s, *lay, params, obs, obs_strides[0], obs_strides[1], obs_strides[2], w.loc, w.scale, w.pooled, w.hid};
    e = slode_launch_enc_fwd(ef, st);
    if (e == hipErrorInvalidValue) return fail(h, SLODE_EINVAL, "unsupported encoder shape C=%d K=%d T=%d", s->C, s->K, s->T);
    HIP_TRY(h, e);
  }

  int n_slabs = w.ode_grid;
  int part_lo = lay->ode_begin, part_hi = lay->n_params;   // flat range the slab rows carry (after the loss slot)
  int zr_rows = 0, zr_lo = 0, zr_hi = 0;                   // rows [0, zr_rows) carry nothing in slab columns [zr_lo, zr_hi) (dopri5 scorer)
  if (phase == 2) {
    if (aux_mode) { part_lo = lay->aux_w1[0]; part_hi = lay->cstd; }
  } else if (aux_mode) {
    // one workgroup per trajectory up to 2,048 of them, then a loop; on the folded path the kernel also runs the encoder-head backward and
    // its slab rows carry only the label-head range (the fused tail below reduces exactly that)
    AuxLaunch al{*s, *lay, params, w.loc, w.scale, eps, u, w.g_loc, w.g_scale, w.ode_slabs, w.ode_stride,
                 w.ode_grid < 2048 ? w.ode_grid : 2048, bwd ? 1 : 0};
    al.rng = rng; al.lab = lab;
    // compact rows carry the flat range [aux_w1[0], cstd): every label-head tensor must lie inside it (a caller-made layout may not)
    bool aux_contig = lay->aux_w1[0] <= lay->cstd;
    for (int a = 0; a < s->n_aux; ++a) {
      const int hi = s->aux[a].kind == SLODE_AUX_EXPEXP ? lay->aux_c[a] + 1 : lay->aux_b2[a] + s->aux[a].u_dim;
      aux_contig = aux_contig && lay->aux_w1[a] >= lay->aux_w1[0] && hi <= lay->cstd;
    }
    if (bwd && folded && !aux_contig) return fail(h, SLODE_EINVAL, "slode_aux_step: the label heads must lie in [aux_w1[0], cstd) of the layout (slode_layout_init's order)");
    if (bwd && folded) {
      al.compact = 1; al.enc_hid = w.hid; al.g_pre = w.g_pre; al.glat = w.glat; al.g_loc = nullptr; al.g_scale = nullptr;
      part_lo = lay->aux_w1[0]; part_hi = lay->cstd;
    }
    n_slabs = al.grid;
    HIP_TRY(h, slode_launch_aux(al, st));
  } else {
    OdeLaunch a{};
    a.s = scorer_shape(*s); a.lay = *lay; a.params = params; a.times = times; a.stage_t = dp5 ? times : stage_t;
    a.obs = obs; a.sb = obs_strides[0]; a.sc = obs_strides[1]; a.st = obs_strides[2];
    a.u = u; a.eps = eps; a.loc = w.loc; a.scale = w.scale; a.x_out = x_out; a.z_out = z_out;
    a.g_loc = w.g_loc; a.g_scale = w.g_scale; a.slabs = w.ode_slabs; a.slab_stride = w.ode_stride; a.grid = w.ode_grid;
    a.backward = bwd ? 1 : 0; a.with_ll = 1;
    a.rng = rng; a.lab = lab;
    a.sigtab = folded ? w.sigtab : nullptr;   // written by the fold launch above
    a.force_loop = h->ode_loop; a.force_generic = h->ode_generic; a.alg = h->ode_alg; a.pack = h->ode_pack;
    if (bwd && folded) { a.enc_hid = w.hid; a.g_pre = w.g_pre; a.glat = w.glat; a.g_loc = nullptr; a.g_scale = nullptr; }
    if (enc_fused) { a.enc_fuse = 1; a.enc_weff = w.weff; a.enc_beff = w.beff; a.enc_hid_out = w.hid; }
    // ext_skip assumes slode_layout_init's order: the ten solver-side tensors tile [init_w1, dyn_bd + S) exactly and nothing else (prior
    // nets, decoder heads, label heads, constant_std) lies inside; a caller-made layout that does not keeps the zeros written and read
    bool solver_contig = lay->init_b1 == lay->init_w1 + s->H * s->L && lay->init_w2 == lay->init_b1 + s->H && lay->init_b2 == lay->init_w2 + s->S * s->H &&
                         lay->dyn_wh == lay->init_b2 + s->S && lay->dyn_bh == lay->dyn_wh + s->H * (1 + s->L) && lay->dyn_wg == lay->dyn_bh + s->H &&
                         lay->dyn_bg == lay->dyn_wg + s->S * s->H && lay->dyn_wd == lay->dyn_bg + s->S && lay->dyn_bd == lay->dyn_wd + s->S * s->H;
    {
      const int lo = lay->init_w1, hi = lay->dyn_bd + s->S;
      auto inside = [&](int off) { return off >= lo && off < hi; };
      for (int g = 0; g < s->n_groups; ++g)
        solver_contig = solver_contig && !inside(lay->ploc_w[g]) && !inside(lay->ploc_b[g]) && !inside(lay->pls_w[g]) && !inside(lay->pls_b[g]);
      for (int q = 0; q < (s->likelihood == SLODE_GAUSS ? 1 : 3); ++q) solver_contig = solver_contig && !inside(lay->head_w[q]);
      for (int a2 = 0; a2 < s->n_aux; ++a2)
        solver_contig = solver_contig && !inside(lay->aux_w1[a2]) && !inside(lay->aux_b1[a2]) && !inside(lay->aux_w2[a2]) && !inside(lay->aux_b2[a2]) &&
                        (s->aux[a2].kind != SLODE_AUX_EXPEXP || (!inside(lay->aux_w3[a2]) && !inside(lay->aux_b3[a2]) && !inside(lay->aux_c[a2])));
      solver_contig = solver_contig && !inside(lay->cstd);
    }
    if (dp5 && bwd && folded && solver_contig && w.ode_grid + w.dp_rows > 2 * SLODE_REDUCE_GROUPS) {
      // the scorer's rows carry nothing in the solver-side range [init net | dynamics] (the reverse sweep's rows do): the scorer does not
      // write those zeros and stage 1 of the fused tail (the only reader of the rows) does not read them
      a.ext_skip = 1;
      zr_rows = w.ode_grid; zr_lo = 1 + (lay->init_w1 - lay->ode_begin); zr_hi = 1 + (lay->dyn_bd + s->S - lay->ode_begin);
    }
    if (dp5) {
      // adaptive solve (per-trajectory controller, accepted steps recorded) -> ONE scorer pass (loss terms, dLoss/dx, every gradient
      // that does not flow through the solver, its own share of the latent gradient into g_loc / g_scale) -> reverse mode over the
      // records: solver-side gradients as extra slab rows, the latent gradient through the solver added to g_loc / g_scale -> the
      // unfused encoder tail below
      DopriRec rc{w.loc, w.scale, eps, w.dp_z, bwd ? w.dp_rec : nullptr, w.dp_nrec, w.dp_kmax};
      rc.w64 = dp5_lanes(h, s->B);
      const bool hand_over = bwd && (rc.w64 == 8 || rc.w64 == 16);   // forward workgroups of sixteen trajectories, as the reverse sweep's
      rc.tabs = hand_over ? w.dp_tabs : nullptr;
      if (rng.on) {   // the forward kernel draws the noise once and materialises it: the scorer and the reverse sweep read the same values
        rc.rng = rng; rc.eps_out = w.dp_eps;
        a.rng = RngK{}; a.eps = w.dp_eps; eps = w.dp_eps;
      }
      HIP_TRY(h, slode_launch_dopri5(*s, *lay, params, times, nullptr, w.dp_x, st, &rc));
      a.x_ext = w.dp_x;
      if (bwd) {
        a.gx_out = w.dp_gx;
        a.enc_hid = nullptr; a.g_pre = nullptr; a.glat = nullptr; a.g_loc = w.g_loc; a.g_scale = w.g_scale;
        n_slabs = w.ode_grid + w.dp_rows;
      }
    }
    e = slode_launch_ode(a, st, h->err, sizeof(h->err));
    if (e == hipErrorInvalidValue) return SLODE_EINVAL;
    HIP_TRY(h, e);
    if (dp5 && bwd) {
      DopriRec rc{w.loc, w.scale, eps, w.dp_z, w.dp_rec, w.dp_nrec, w.dp_kmax};
      { const int l5 = dp5_lanes(h, s->B); rc.tabs = (l5 == 8 || l5 == 16) ? w.dp_tabs : nullptr; }
      HIP_TRY(h, slode_launch_dopri5_bwd(*s, *lay, params, times, rc, w.dp_gx, w.g_loc, w.g_scale, w.ode_slabs + (size_t)w.ode_grid * w.ode_stride,
                                         w.ode_stride, s->grad_mode == SLODE_GRAD_REFERENCE_ADJOINT ? 1 : 0, w.dp_snap, st,
                                         folded ? w.hid : nullptr, folded ? w.g_pre : nullptr, folded ? w.glat : nullptr));
    }
  }

  if (phase != 0 && !(bwd && folded)) return fail(h, SLODE_EINVAL, "the payload split needs a backward step on the folded encoder path");
  if (bwd && folded) {
    // Fused tail.  The ODE kernel (auxiliary step: the aux kernel; dopri5: its reverse sweep) has already run the encoder heads + tanh
    // backward (g_pre, glat): two launches remain --
    // split-K MFMA GEMMs (+ rider blocks: stage 1 of the ODE-slab reduction), chain rule, one final reduction (+ Adam).
    AdamHost ah{};
    if (adam) {
      ah = AdamHost{adam->p, adam->m, adam->v, adam->lr, adam->b1, adam->b2, adam->eps, adam->step, adam->n};
      ah.lo2 = h->adam_lo2; ah.hi2 = h->adam_hi2; ah.delta2 = h->adam_delta2;
    }
    const float* ode_part = nullptr;
    int ode_pn = 0;
    const PayloadMap pm = payload_map(*s, part_hi - part_lo);
    if (phase != 2)
      HIP_TRY(h, slode_launch_gemm_tail(w.g_pre, obs, w.gslabs, s->Hc, (int)CT, w.glat, w.hid, w.gslabs2, w.gslabs3, s->L, s->B, w.gsplit,
                                        w.ode_slabs, w.ode_stride, n_slabs, (part_hi - part_lo) + 1, w.ode_part, &ode_part, &ode_pn, st,
                                        zr_rows, zr_lo, zr_hi));
    if (phase == 1) {   // split-K partials and partial slab rows, summed in fixed order, into the contiguous payload
      HIP_TRY(h, slode_launch_pack_payload(w.gslabs, w.gslabs2, w.gslabs3, w.gsplit, s->Hc, (int)CT, s->L, ode_part, w.ode_stride, ode_pn,
                                           (part_hi - part_lo) + 1, payload, pm.g_loc, pm.g_ls, pm.ode, pm.total, st));
      return SLODE_OK;
    }
    TailK tl{};
    tl.gslabs = w.gslabs; tl.gslabs_loc = w.gslabs2; tl.gslabs_ls = w.gslabs3; tl.conv_slabs = w.conv_slabs;
    tl.ode_part = ode_part; tl.ode_stride = w.ode_stride; tl.ode_n = ode_pn; tl.loss_out = loss_out;
    int gsplit_eff = w.gsplit;
    if (phase == 2) {   // the (reduced) payload stands for ONE split / ONE partial row
      fl.gslabs = payload; fl.n_gslabs = 1; gsplit_eff = 1;
      tl.gslabs = payload; tl.gslabs_loc = payload + pm.g_loc; tl.gslabs_ls = payload + pm.g_ls;
      tl.ode_part = payload + pm.ode; tl.ode_stride = 0; tl.ode_n = 1;
    }
    tl.part_lo = part_lo; tl.part_hi = part_hi;
    tl.gsplit = gsplit_eff; tl.Hc = s->Hc; tl.L = s->L; tl.CT = (int)CT; tl.n_cv = s->F * s->C * s->K + s->F;
    tl.conv_w = lay->conv_w; tl.lin_w = lay->lin_w; tl.lin_b = lay->lin_b; tl.zloc_w = lay->zloc_w; tl.zloc_b = lay->zloc_b;
    tl.zls_w = lay->zls_w; tl.zls_b = lay->zls_b; tl.ode_begin = lay->ode_begin; tl.n_params = lay->n_params;
    tl.n_total = (adam && adam->n > lay->n_params) ? (int)adam->n : lay->n_params;
    tl.grads = grads; tl.ad = make_adamk(adam ? &ah : nullptr); tl.counter = w.counter;
    // FOLD-NEXT: this launch updates the weights (Adam inside) => it also folds them for the next step, provided every block of the launch
    // is resident at once (the blocks wait for each other) and the handle's W_eff bookkeeping covers this workspace
    int n_chain = 0;
    int nblk = slode_chain_blocks(*s, tl.n_total, lay->lin_b, &n_chain);
    int resident = 0;
    if (h->fold_on && adam) {   // (the occupancy query is asked once per shape and handle)
      const int sig[8] = {s->T, s->C, s->F, s->K, s->P, s->Hc, s->L, lay->n_params};
      if (memcmp(sig, h->chain_resident_sig, sizeof(sig)) != 0) {
        h->chain_resident = slode_chain_resident_blocks(*s, h->num_cu);
        memcpy(h->chain_resident_sig, sig, sizeof(sig));
      }
      resident = h->chain_resident;
    }
    tl.n_riders = 0;
    if (nblk > resident && resident > n_chain && nblk - n_chain <= 4 * (resident - n_chain)) {   // fewer, looping riders: the grid fits
      tl.n_riders = resident - n_chain;
      nblk = resident;
    }
    const bool fold_next = h->fold_on && adam && adam->p == params && nblk <= resident && h->fold_valid && h->fold_ws == workspace &&
                           h->fold_params == (const void*)params;
    tl.fold_next = fold_next ? 1 : 0;
    if (!fold_next) tl.n_riders = 0;
    tl.done = w.counter; tl.done_target = 0; tl.cstd_off = lay->cstd; tl.gauss = s->likelihood == SLODE_GAUSS ? 1 : 0;
    tl.sigtab = w.sigtab;
    if (fold_next) {
      ++h->fold_gen;
      tl.done_target = h->fold_gen;   // (the generation: every counter's target is gen x its number of arrivals per launch)
      if (tl.n_riders == 0) tl.n_riders = nblk - n_chain;
    }
    fl.tail = &tl;
    if (adam) h->fold_valid = fold_next ? 1 : 0;   // the weights change now: what the workspace holds is current only if this launch re-folds
    HIP_TRY(h, slode_launch_fold_chain(fl, st));   // + rider blocks and the last-block conv reduction: the flat gradient is complete
  } else if (bwd) {
    if (adam) h->fold_valid = 0;
    EncBwdLaunch eb{*s, *lay, params, obs, obs_strides[0], obs_strides[1], obs_strides[2], w.scale, w.pooled, w.hid,
                    w.g_loc, w.g_scale, w.g_pre, w.small_slabs, w.small_stride, w.small_grid, w.lin_slabs, w.lin_splitk};
    HIP_TRY(h, slode_launch_enc_bwd(eb, st));
    ReduceLaunch r{*s, *lay, w.ode_slabs, w.ode_stride, n_slabs, w.small_slabs, w.small_stride, w.small_grid,
                   w.lin_slabs, w.lin_splitk, grads, loss_out, 1, w.ode_part, w.small_part, 0};
    if (adam) { r.adam_p = adam->p; r.adam_m = adam->m; r.adam_v = adam->v; r.adam_lr = adam->lr; r.adam_b1 = adam->b1;
                r.adam_b2 = adam->b2; r.adam_eps = adam->eps; r.adam_step = adam->step; r.adam_n = adam->n;
                r.adam_lo2 = h->adam_lo2; r.adam_hi2 = h->adam_hi2; r.adam_delta2 = h->adam_delta2; }
    HIP_TRY(h, slode_launch_reduce(r, st));
  } else {
    ReduceLaunch r{*s, *lay, w.ode_slabs, w.ode_stride, n_slabs, nullptr, 0, 0, nullptr, 0, nullptr, loss_out, 0, w.ode_part, nullptr, 0};
    HIP_TRY(h, slode_launch_reduce(r, st));
  }
  return SLODE_OK;
}

int slode_elbo_step(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* times,
                    const float* stage_t, const float* obs, const int64_t obs_strides[3], const float* u, const float* eps,
                    float* loss_out, float* grads, float* x_out, float* z_out, void* workspace, size_t workspace_bytes,
                    void* stream) {
  return elbo_step_impl(h, s, lay, params, times, stage_t, obs, obs_strides, u, eps, loss_out, grads, x_out, z_out, workspace,
                        workspace_bytes, stream, nullptr);
}

int slode_elbo_adam_step(slode_handle h, const slode_shape* s, const slode_layout* lay, float* params, const float* times,
                         const float* stage_t, const float* obs, const int64_t obs_strides[3], const float* u, const float* eps,
                         float* loss_out, float* grads, void* workspace, size_t workspace_bytes, int64_t n_total, float* exp_avg,
                         float* exp_avg_sq, float lr, float beta1, float beta2, float adam_eps, int64_t step, void* stream) {
  if (!grads || !exp_avg || !exp_avg_sq || step < 1 || !lay || n_total < lay->n_params)
    return fail(h, SLODE_EINVAL, "slode_elbo_adam_step needs grads, Adam moments, step >= 1 and n_total >= layout n_params");
  const AdamArgs ad{params, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, step, n_total};
  return elbo_step_impl(h, s, lay, params, times, stage_t, obs, obs_strides, u, eps, loss_out, grads, nullptr, nullptr, workspace,
                        workspace_bytes, stream, &ad);
}

int slode_aux_step(slode_handle h, const slode_shape* s, const slode_layout* lay, float* params, const float* obs,
                   const int64_t obs_strides[3], const float* u, const float* eps, float* loss_out, float* grads, void* workspace,
                   size_t workspace_bytes, int64_t n_total, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                   float adam_eps, int64_t step, void* stream) {
  const bool with_adam = exp_avg != nullptr;
  if (with_adam && (!grads || !exp_avg_sq || step < 1 || !lay || n_total < lay->n_params))
    return fail(h, SLODE_EINVAL, "slode_aux_step with Adam needs grads, both moments, step >= 1 and n_total >= layout n_params");
  const AdamArgs ad{params, exp_avg, exp_avg_sq, lr, beta1, beta2, adam_eps, step, n_total};
  return elbo_step_impl(h, s, lay, params, nullptr, nullptr, obs, obs_strides, u, eps, loss_out, grads, nullptr, nullptr, workspace,
                        workspace_bytes, stream, with_adam ? &ad : nullptr, 1);
}

int slode_svi_step(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, float* params, const float* times,
                   const float* stage_t, const slode_batch* batch, float* loss_out, float* grads, void* workspace, size_t workspace_bytes,
                   const slode_adam* adam, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (!s || !batch) return fail(h, SLODE_EINVAL, "shape / batch is NULL");
  if (kind != SLODE_SVI_MAIN && kind != SLODE_SVI_AUX) return fail(h, SLODE_EINVAL, "kind must be SLODE_SVI_MAIN or SLODE_SVI_AUX");
  LabelSrc lab{};
  const int rc_lab = batch_labels(h, s, batch, &lab);
  if (rc_lab != SLODE_OK) return rc_lab;
  AdamArgs ad{};
  if (adam) {
    if (!grads || !adam->exp_avg || !adam->exp_avg_sq || adam->step < 1 || !lay || adam->n_total < lay->n_params)
      return fail(h, SLODE_EINVAL, "slode_svi_step with Adam needs grads, both moments, step >= 1 and n_total >= layout n_params");
    ad = AdamArgs{params, adam->exp_avg, adam->exp_avg_sq, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->n_total};
  }
  return elbo_step_impl(h, s, lay, params, kind == SLODE_SVI_AUX ? nullptr : times, kind == SLODE_SVI_AUX ? nullptr : stage_t, batch->obs,
                        batch->obs_strides, nullptr, batch->eps, loss_out, grads, nullptr, nullptr, workspace, workspace_bytes, stream,
                        adam ? &ad : nullptr, kind == SLODE_SVI_AUX ? 1 : 0, batch->n_labels > 0 ? &lab : nullptr);
}

size_t slode_grad_payload_floats(const slode_shape* s, const slode_layout* lay, int kind) {
  if (check_shape(s) || !lay) return 0;
  const int part = kind == SLODE_SVI_AUX ? lay->cstd - lay->aux_w1[0] : lay->n_params - lay->ode_begin;
  return (size_t)payload_map(*s, part).total;
}

int slode_grad_partial(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, const float* params, const float* times,
                       const float* stage_t, const slode_batch* batch, float* payload, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (!s || !batch) return fail(h, SLODE_EINVAL, "shape / batch is NULL");
  if (kind != SLODE_SVI_MAIN && kind != SLODE_SVI_AUX) return fail(h, SLODE_EINVAL, "kind must be SLODE_SVI_MAIN or SLODE_SVI_AUX");
  LabelSrc lab{};
  const int rc = batch_labels(h, s, batch, &lab);
  if (rc != SLODE_OK) return rc;
  const bool aux = kind == SLODE_SVI_AUX;
  return elbo_step_impl(h, s, lay, params, aux ? nullptr : times, aux ? nullptr : stage_t, batch->obs, batch->obs_strides, nullptr, batch->eps,
                        nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream, nullptr, aux ? 1 : 0,
                        batch->n_labels > 0 ? &lab : nullptr, 1, payload);
}

int slode_grad_apply(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, float* params, const int64_t obs_strides[3],
                     const float* payload, float* loss_out, float* grads, void* workspace, size_t workspace_bytes, const slode_adam* adam,
                     void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (kind != SLODE_SVI_MAIN && kind != SLODE_SVI_AUX) return fail(h, SLODE_EINVAL, "kind must be SLODE_SVI_MAIN or SLODE_SVI_AUX");
  AdamArgs ad{};
  if (adam) {
    if (!grads || !adam->exp_avg || !adam->exp_avg_sq || adam->step < 1 || !lay || adam->n_total < lay->n_params)
      return fail(h, SLODE_EINVAL, "slode_grad_apply with Adam needs grads, both moments, step >= 1 and n_total >= layout n_params");
    ad = AdamArgs{params, adam->exp_avg, adam->exp_avg_sq, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->n_total};
  }
  return elbo_step_impl(h, s, lay, params, nullptr, nullptr, nullptr, obs_strides, nullptr, nullptr, loss_out, grads, nullptr, nullptr, workspace,
                        workspace_bytes, stream, adam ? &ad : nullptr, kind == SLODE_SVI_AUX ? 1 : 0, nullptr, 2, const_cast<float*>(payload));
}

int slode_fold_invalidate(slode_handle h) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  h->fold_valid = 0;
  return SLODE_OK;
}

int slode_rng_seed(slode_handle h, uint64_t seed, int64_t first_trajectory) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (first_trajectory < 0) return fail(h, SLODE_EINVAL, "first_trajectory < 0");
  h->rng_seed = seed; h->rng_b0 = first_trajectory; h->rng_counter = 0;
  return SLODE_OK;
}

int slode_rng_set_counter(slode_handle h, uint64_t n) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  h->rng_counter = n;
  return SLODE_OK;
}

int slode_rng_get(slode_handle h, uint64_t* seed, int64_t* first_trajectory, uint64_t* n) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (seed) *seed = h->rng_seed;
  if (first_trajectory) *first_trajectory = h->rng_b0;
  if (n) *n = h->rng_counter;
  return SLODE_OK;
}

int slode_rng_normal(slode_handle h, uint64_t n, int32_t B, int32_t L, float* eps_out, uint32_t* raw_out, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (B < 1 || L < 1 || L > SLODE_MAX_L || (!eps_out && !raw_out)) return fail(h, SLODE_EINVAL, "slode_rng_normal: B >= 1, 1 <= L <= 64 and an output required");
  HIP_TRY(h, slode_launch_rng_fill(rng_of(h, n), B, L, eps_out, raw_out, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_sample_normal(slode_handle h, int32_t B, int32_t L, const float* loc, const float* scale, float* z_out, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (B < 1 || L < 1 || L > SLODE_MAX_L || !loc || !scale || !z_out) return fail(h, SLODE_EINVAL, "slode_sample_normal: B >= 1, 1 <= L <= 64, loc / scale / z_out required");
  HIP_TRY(h, slode_launch_rng_fill(rng_of(h, h->rng_counter++), B, L, z_out, nullptr, (hipStream_t)stream, loc, scale));
  return SLODE_OK;
}

int slode_dynamics_eval(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, float t,
                        const float* state, const float* z, float* out, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!state || !z || !out) return fail(h, SLODE_EINVAL, "state / z / out is NULL");
  HIP_TRY(h, slode_launch_dynamics_eval(*s, *lay, params, t, state, z, out, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_initialize_state(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* z, float* x0,
                           void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!z || !x0) return fail(h, SLODE_EINVAL, "z / x0 is NULL");
  HIP_TRY(h, slode_launch_init_state(*s, *lay, params, z, x0, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_prior_nets(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* u, float* loc,
                     float* scale, void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!loc || !scale || (s->n_groups > 0 && !u)) return fail(h, SLODE_EINVAL, "u / loc / scale is NULL");
  HIP_TRY(h, slode_launch_prior_nets(*s, *lay, params, u, loc, scale, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_label_heads(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* z, float* out,
                      void* stream) {
  const char* why = check_common(h, s, lay, params);
  if (why) return fail(h, SLODE_EINVAL, "%s", why);
  if (!z || !out) return fail(h, SLODE_EINVAL, "z / out is NULL");
  if (s->n_aux < 1) return fail(h, SLODE_EINVAL, "the shape has no label heads");
  HIP_TRY(h, slode_launch_label_heads(*s, *lay, params, z, out, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_dopri5_step_counts(slode_handle h, const slode_shape* s, const slode_layout* lay, const void* workspace, size_t workspace_bytes,
                             int* counts, void* stream) {
  if (!h) return SLODE_EINVAL;
  if (!s || !lay || !workspace || !counts || s->method != SLODE_DOPRI5) return fail(h, SLODE_EINVAL, "slode_dopri5_step_counts: dopri5 shape, workspace and output required");
  if (workspace_bytes < slode_workspace_bytes(h, s)) return fail(h, SLODE_EINVAL, "slode_dopri5_step_counts: workspace too small");
  const Workspace w = carve(h, *s, *lay, const_cast<void*>(workspace));
  HIP_TRY(h, hipMemcpyAsync(counts, w.dp_nrec, sizeof(int) * (size_t)s->B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SLODE_OK;
}

int slode_adam_region(slode_handle h, int64_t lo, int64_t hi, int64_t step_delta) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (lo < 0 || hi < lo || hi > 0x7fffffff) return fail(h, SLODE_EINVAL, "bad Adam region [%lld, %lld)", (long long)lo, (long long)hi);
  h->adam_lo2 = (int)lo; h->adam_hi2 = (int)hi; h->adam_delta2 = step_delta;
  return SLODE_OK;
}

int slode_profile_enable(slode_handle h, int on) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (on != 0 && on != 1) return fail(h, SLODE_EINVAL, "profile mode %d: 0 (off) or 1 (per-kernel timestamps)", on);
  if (on && !h->ev_ready) {
    for (int i = 0; i < SLODE_CLOCK_MAX; ++i) {
      HIP_TRY(h, hipEventCreate(&h->clk.ev[i][0]));
      HIP_TRY(h, hipEventCreate(&h->clk.ev[i][1]));
    }
    h->ev_ready = 1;
  }
  h->profile = on;
  h->clk.n = 0;
  return SLODE_OK;
}

int slode_profile_read(slode_handle h, int max_kernels, const char** names, float* us) {
  if (!h || !names || !us || max_kernels < 1) return fail(h, SLODE_EINVAL, "handle / names / us is NULL or max_kernels < 1");
  if (!h->profile) return fail(h, SLODE_EINVAL, "profiling is off");
  if (h->clk.n < 1) return fail(h, SLODE_EINVAL, "no profiled step has been recorded on this handle");
  const int n = h->clk.n < max_kernels ? h->clk.n : max_kernels;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    HIP_TRY(h, hipEventSynchronize(h->clk.ev[i][1]));
    HIP_TRY(h, hipEventElapsedTime(&ms, h->clk.ev[i][0], h->clk.ev[i][1]));
    names[i] = h->clk.name[i];
    us[i] = 1e3f * ms;
  }
  return n;
}

int slode_adam_step(slode_handle h, int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float lr,
                    float beta1, float beta2, float eps, int64_t step, void* stream) {
  if (!h) return fail(nullptr, SLODE_EINVAL, "handle is NULL");
  if (n < 0 || step < 1 || !params || !grads || !exp_avg || !exp_avg_sq) return fail(h, SLODE_EINVAL, "bad Adam arguments");
  if (n == 0) return SLODE_OK;
  AdamHost a{params, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, n};
  a.lo2 = h->adam_lo2; a.hi2 = h->adam_hi2; a.delta2 = h->adam_delta2;
  h->fold_valid = 0;   // the weights change outside a step's chain launch: the next step folds again
  ClockScope clock_scope(h, true);
  HIP_TRY(h, slode_launch_adam_k(n, grads, a, (hipStream_t)stream));
  return SLODE_OK;
}

}  // extern "C"
