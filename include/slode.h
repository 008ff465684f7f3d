/*
 * slode.h -- C ABI of libslode.so: the MI355X (gfx950) engine for the latent-ODE solve + ELBO path of
 * paidamoyo/structured_latent_ODEs.
 *
 * The reference has no FFI of its own (its only API is Python classes); every entry point below
 * names the reference Python interface it replaces (file:line relative to the reference repo).  The Python host
 * side (structured_latent_odes_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - all tensors are fp32, device (HBM) pointers owned by the CALLER; the library never allocates device
 *     memory, never synchronises the stream, and launches on the hipStream_t passed as `void* stream`;
 *   - the environment is read once per handle, in slode_create (diagnostic switches SLODE_NO_FOLD, SLODE_ODE_LOOP, SLODE_ODE_GRID,
 *     SLODE_ODE_GENERIC, SLODE_ODE_ALG, SLODE_ODE_PACK, SLODE_ENC_FUSE, SLODE_DP5_LPT); nothing about a launch depends on the environment at call time;
 *   - `times` must be strictly monotone (torchdiffeq's precondition); a table that is not turns the fused kernel's loss into NaN;
 *   - every call returns SLODE_OK (0) or a negative slode_status; slode_last_error() gives the text;
 *   - all model parameters live in ONE flat fp32 vector whose segment offsets are given by slode_layout
 *     (filled by slode_layout_init); gradients use the same layout;
 *   - observations are addressed as the logical [B, C, T] tensor through explicit element strides, so the
 *     reference's permuted view of a contiguous [B, T, C] batch (training_cvs.py:25) is consumed without a copy.
 */
#ifndef SLODE_H
#define SLODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLODE_VERSION 110 /* 0.1.1: slode_svi_step, slode_rng_*, slode_grad_* */

#define SLODE_MAX_GROUPS 4
#define SLODE_MAX_HEADS 3
#define SLODE_MAX_AUX 4
#define SLODE_MAX_LABELS 4 /* label tensors of one minibatch (proc: aR, aS, C12, C6) */

typedef enum slode_status {
  SLODE_OK = 0,
  SLODE_EINVAL = -1, /* bad shape / stride / null pointer / unsupported dimension */
  SLODE_EHIP = -2,   /* a HIP runtime call failed (text from hipGetErrorString)      */
  SLODE_ENOSPC = -3  /* caller workspace too small                                   */
} slode_status;

/* torchdiffeq method strings accepted by OdeModel.init_with_params(solver=...), models/blackbox_ode.py:7-17,41-45 */
typedef enum slode_method { SLODE_EULER = 0, SLODE_MIDPOINT = 1, SLODE_RK4 = 2, SLODE_DOPRI5 = 3 } slode_method;

/* Decoder (models/decoders.py:8-54, asymmetric-Laplace, 3 heads q50/q75/q25) or GaussianDecoder (:57-91, 1 head) */
typedef enum slode_likelihood { SLODE_ALD = 0, SLODE_GAUSS = 1 } slode_likelihood;
/* SLODE_GRAD_EXACT: exact gradient of the discrete scheme (== reference with adjoint_solver=False).
 * SLODE_GRAD_REFERENCE_ADJOINT: what torchdiffeq.odeint_adjoint returns, the reference default (models/blackbox_ode.py:40-42,
 * adjoint_solver = True in all three configs): the continuous adjoint stepped backwards with the same fixed-grid method, and NO
 * gradient to z through the dynamics (OdeFunc.constants is not a parameter, :55).  Fixed-grid methods, ELBO / solve backward. */
typedef enum slode_grad_mode { SLODE_GRAD_EXACT = 0, SLODE_GRAD_REFERENCE_ADJOINT = 1 } slode_grad_mode;

/* One conditional prior net p(z_g | u_g): EncoderMLP([u_dim, [z_dim, z_dim]], [None, Exp]);
 * models/mechanistic_cvs.py:88-100, mechanistic_proc.py:107-114, mechanistic_challenge.py:88-95 */
typedef struct slode_group {
  int32_t z_off, z_dim; /* latent dims [z_off, z_off + z_dim) */
  int32_t u_off, u_dim; /* label columns [u_off, u_off + u_dim) of u[B, n_u] */
} slode_group;

/* A label head q(label | z_g): EncoderMLP([z_dim, U, u_dim]) with one Softplus hidden layer.  Scored at `aux_mult` x by the
 * auxiliary loss (model_meta: mechanistic_cvs.py:240-270, mechanistic_proc.py:313-353, mechanistic_challenge.py:264-291) and,
 * when `aux_in_main` is set (the proc family: q_label / q_continous on the replayed z, mechanistic_proc.py:145-146), also
 * inside the main loss.
 *   SLODE_AUX_SIGMOID: Bernoulli(probs = sigmoid(.));  SLODE_AUX_SOFTMAX: OneHotCategorical(probs = softmax(.));
 *   SLODE_AUX_EXPEXP:  two Exp heads [loc, unused], Laplace(loc, softplus(constant_std_*)) on the label. */
typedef enum slode_aux_kind { SLODE_AUX_SIGMOID = 0, SLODE_AUX_SOFTMAX = 1, SLODE_AUX_EXPEXP = 2 } slode_aux_kind;
typedef struct slode_aux {
  int32_t kind;
  int32_t z_off, z_dim; /* latent dims the head reads */
  int32_t u_off, u_dim; /* label columns it scores   */
} slode_aux;

typedef struct slode_shape {
  int32_t B;  /* trajectories in this call (per GPU)                                   */
  int32_t T;  /* time points (len(times))                                              */
  int32_t C;  /* observed channels, config.obs_dim                                      */
  int32_t L;  /* latent dim = sum of z_*_dim                                           */
  int32_t S;  /* config.ode_state_dim                                                  */
  int32_t H;  /* config.ode_hidden_dim                                                 */
  int32_t F;  /* config.n_filters                                                      */
  int32_t K;  /* config.filter_size                                                    */
  int32_t P;  /* config.pool_size                                                      */
  int32_t Hc; /* config.cnn_hidden_dim                                                 */
  int32_t n_u;      /* label columns of u                                              */
  int32_t n_groups; /* conditional prior groups; remaining latent dims are N(0, 1)     */
  slode_group groups[SLODE_MAX_GROUPS];
  int32_t method;     /* slode_method                                                  */
  int32_t likelihood; /* slode_likelihood                                              */
  float quantile_diff; /* config.quantile_diff (ALD only), data/cvs/config_cvs.py:48    */
  float rtol, atol;    /* dopri5 only (torchdiffeq defaults 1e-7 / 1e-9)                */
  int32_t n_aux;       /* label heads (auxiliary loss; also the main loss iff aux_in_main)  */
  int32_t U;           /* config.u_hidden_dim (<= 32)                                      */
  float aux_mult;      /* config.aux_loss_multiplier                                       */
  slode_aux aux[SLODE_MAX_AUX];
  int32_t aux_in_main; /* 1: the main model scores the label heads too (proc family)       */
  int32_t grad_mode;   /* slode_grad_mode: which gradient the backward pass returns         */
} slode_shape;

/* Offsets (in floats) of each parameter tensor inside the flat parameter / gradient vector.
 * Names are the reference state_dict keys they hold. */
typedef struct slode_layout {
  int32_t conv_w, conv_b;   /* encoder.conv.{weight[F,C,K], bias[F]}          encoder_conv.py:31 */
  int32_t lin_w, lin_b;     /* encoder.lin.{weight[Hc,F*n_pool], bias[Hc]}    encoder_conv.py:34 */
  int32_t zloc_w, zloc_b;   /* encoder.z_loc.{weight[L,Hc], bias[L]}          encoder_conv.py:37 */
  int32_t zls_w, zls_b;     /* encoder.z_scale.0.{weight[L,Hc], bias[L]}      encoder_conv.py:38 */
  int32_t ode_begin;        /* start of the segment the fused ODE/ELBO kernel differentiates      */
  int32_t ploc_w[SLODE_MAX_GROUPS], ploc_b[SLODE_MAX_GROUPS]; /* <prior>.sequential_mlp.1.0.0.*      */
  int32_t pls_w[SLODE_MAX_GROUPS], pls_b[SLODE_MAX_GROUPS];   /* <prior>.sequential_mlp.1.1.0.*      */
  int32_t init_w1, init_b1; /* decoder.ode_model.latent_to_ode_net.0.{weight[H,L], bias[H]}          */
  int32_t init_w2, init_b2; /* decoder.ode_model.latent_to_ode_net.2.{weight[S,H], bias[S]}          */
  int32_t dyn_wh, dyn_bh;   /* ...dynamics.dynamics_hidden.{weight[H,1+L], bias[H]} (col 0 = time)   */
  int32_t dyn_wg, dyn_bg;   /* ...dynamics.dyanamics_growth.{weight[S,H], bias[S]}                   */
  int32_t dyn_wd, dyn_bd;   /* ...dynamics.dyanmics_degradation.{weight[S,H], bias[S]}               */
  int32_t head_w[SLODE_MAX_HEADS]; /* decoder.output_{q50,q75,q25}.0.weight[C,S] | output_mean (Gauss) */
  /* label heads of the main loss: <head>.sequential_mlp.1.module.{weight[U,z_dim],bias[U]}, .3.{weight[u_dim,U],bias}
   * (EXPEXP: .3.0.0.* then .3.1.0.*), and for EXPEXP the scalar constant_std_C_* */
  int32_t aux_w1[SLODE_MAX_AUX], aux_b1[SLODE_MAX_AUX], aux_w2[SLODE_MAX_AUX], aux_b2[SLODE_MAX_AUX];
  int32_t aux_w3[SLODE_MAX_AUX], aux_b3[SLODE_MAX_AUX], aux_c[SLODE_MAX_AUX];
  int32_t cstd;             /* decoder.constant_std[C,T]                                             */
  int32_t ode_end;          /* end of that segment                                                   */
  int32_t n_params;         /* total floats covered by this layout; callers may append their own      */
} slode_layout;

typedef struct slode_ctx* slode_handle;

int slode_version(void);
/* device_id >= 0: HIP device ordinal.  There is no CPU backend: a process without a gfx950 device gets SLODE_EHIP. */
int slode_create(slode_handle* h, int device_id);
int slode_destroy(slode_handle h);
const char* slode_last_error(slode_handle h); /* valid until the next call on h; h may be NULL (global text) */

/* Validates `s` (SLODE_EINVAL on unsupported dimensions) and fills the canonical layout. */
int slode_layout_init(const slode_shape* s, slode_layout* lay);
/* Number of stage times the fixed-grid method evaluates: R*(T-1)+1 (R = 1/2/3 for euler/midpoint/rk4). */
int slode_num_stage_times(const slode_shape* s);
/* Bytes of caller workspace needed by slode_elbo_step / the *_bwd ops for this shape. */
size_t slode_workspace_bytes(slode_handle h, const slode_shape* s);

/* Stage-time table: the distinct times at which the fixed-grid solver evaluates f, computed on device with the
 * same fp32 arithmetic as torchdiffeq's step functions (t0 + dt/3, ...).  Replaces the time bookkeeping inside
 * torchdiffeq.odeint for the call at models/blackbox_ode.py:41-45.  times[T] -> stage_t[slode_num_stage_times]. */
int slode_stage_times(slode_handle h, const slode_shape* s, const float* times, float* stage_t, void* stream);

/* EncoderCONV.forward (models/encoder_conv.py:43-51): obs (logical [B,C,T], strides in elements) -> loc, scale [B,L].
 * `pooled` [B, F*n_pool] and `hid` [B, Hc] are saved for the backward (either may be NULL for inference). */
int slode_encoder_conv_fwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                           const float* obs, const int64_t obs_strides[3], float* loc, float* scale,
                           float* pooled, float* hid, void* stream);

/* Backward of EncoderCONV.forward: (g_loc, g_scale) [B,L] -> grads[conv_w .. zls_b] (overwritten, not accumulated). */
int slode_encoder_conv_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                           const float* obs, const int64_t obs_strides[3], const float* scale,
                           const float* pooled, const float* hid, const float* g_loc, const float* g_scale,
                           float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* OdeModel.solve_ODE (models/blackbox_ode.py:36-47): z[B,L] -> x[B,T,S] (contiguous; the reference returns the
 * same logical tensor as a permuted view).  stage_t from slode_stage_times. */
int slode_ode_solve_fwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                        const float* times, const float* stage_t, const float* z, float* x, void* stream);

/* Exact discrete adjoint of slode_ode_solve_fwd (== autograd through torchdiffeq.odeint, adjoint_solver=False):
 * g_x[B,T,S] -> g_z[B,L] and grads[init_w1 .. dyn_bd] (overwritten). */
int slode_ode_solve_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                        const float* times, const float* stage_t, const float* z, const float* g_x,
                        float* g_z, float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* OdeFunc.forward(t, state) (models/blackbox_ode.py:57-61 -> Dynamics.forward :97-109): one evaluation of
 * dx/dt = a(t,z) - d(t,z) * state for state[B,S], z[B,L] -> out[B,S].  API completeness; the solver does not use it. */
int slode_dynamics_eval(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, float t,
                        const float* state, const float* z, float* out, void* stream);

/* OdeModel.initialize_state (models/blackbox_ode.py:19-22, 32-34): z[B,L] -> x0[B,S] = sigmoid(W2 relu(W1 z + b1) + b2). */
int slode_initialize_state(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* z, float* x0,
                           void* stream);

/* The conditional prior nets p(z_g | u_g) (EncoderMLP([u_dim, [z_dim, z_dim]], [None, Exp]).forward; call sites
 * models/mechanistic_cvs.py:225-237, 304-311 (prior reconstructions)): u[B,n_u] -> loc, scale [B,L]; latent dims outside every group get
 * loc 0, scale 1 (the N(0,1) prior of z_epsilon). */
int slode_prior_nets(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* u, float* loc,
                     float* scale, void* stream);

/* The label heads q(label | z_g) (EncoderMLP([z_dim, U, u_dim]).forward; call sites: classifier / pred_inputs,
 * models/mechanistic_cvs.py:278-296, mechanistic_proc.py:361-380): z[B,L] -> out[B,n_u], every head writing the label columns it scores:
 * Bernoulli probabilities (SIGMOID), class probabilities (SOFTMAX) or the Laplace location exp(.) (EXPEXP). */
int slode_label_heads(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* z, float* out,
                      void* stream);

/* Decoder.forward / GaussianDecoder.forward heads (models/decoders.py:45-53, 86-89) on a given trajectory:
 * x[B,T,S] -> mu[Q][B,C,T] (Q = 3: mu_50, mu_75, mu_25 in that order; Q = 1: mean) and std[C,T] = softplus(constant_std). */
int slode_decode_heads(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                       const float* x, float* mu, float* std_ct, void* stream);

/* Backward of slode_decode_heads (autograd through Decoder.forward / GaussianDecoder.forward, models/decoders.py:42-54, 84-91, as the
 * reference's recon-style callers would differentiate it): g_mu[Q][B,C,T] (zeros for heads without a gradient), g_std[C,T] (NULL: none) ->
 * g_x[B,T,S], g_heads[Q][C,S] (the head weights' gradients, in slode_decode_heads' head order), g_cstd[C,T] (NULL to skip). */
int slode_decode_heads_bwd(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params, const float* x,
                           const float* g_mu, const float* g_std, float* g_x, float* g_heads, float* g_cstd, void* stream);

/* One SVI step's arithmetic for the main loss (pyro SVI.step on (model, guide); call sites training_cvs.py:152,236;
 * models/mechanistic_cvs.py:105-238 and the proc/challenge equivalents):
 *   encoder -> z = loc + scale*eps -> log q, log p -> ODE solve -> heads -> ALD/Gauss likelihood -> -ELBO (summed
 *   over the batch) -> exact gradient wrt every parameter of the layout.
 * Outputs: loss_out[0] = -ELBO (float, device); grads[0 .. lay->n_params) overwritten.
 * With grads == NULL only the loss is computed (SVI.evaluate_loss, training_cvs.py:81).
 * Optional outputs (NULL to skip): x_out[B,T,S] latent trajectories, z_out[B,L].
 * method == SLODE_DOPRI5 (solver="dopri5", models/blackbox_ode.py:41-45): adaptive solve with one controller per trajectory
 * (rtol / atol of the shape); the gradient is the reverse mode of the accepted steps and of the dense output, step sizes held fixed
 * (grad_mode SLODE_GRAD_REFERENCE_ADJOINT: without the z -> dynamics path, as odeint_adjoint).  stage_t is ignored.  At most 65,536
 * trajectories per call; a trajectory whose accepted steps exceed the record capacity (256 MB / (B*(S+2)) floats, clamped to
 * [64, 2048] steps) or that exhausts 20,000 attempted steps turns the loss into NaN.  slode_workspace_bytes accounts for the records, for
 * the running sums the reverse sweep parks at the hidden units' switching times ([B][2][H][4S]: one set per lane group of a trajectory) and
 * for the forward kernel's set-up tables of every sixteen trajectories, which the reverse sweep reads back instead of rebuilding them. */
int slode_elbo_step(slode_handle h, const slode_shape* s, const slode_layout* lay, const float* params,
                    const float* times, const float* stage_t, const float* obs, const int64_t obs_strides[3],
                    const float* u, const float* eps, float* loss_out, float* grads, float* x_out, float* z_out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* slode_elbo_step immediately followed by slode_adam_step, with the Adam update applied by the final gradient-reduction
 * kernel (one launch and one pass over the gradient less; identical arithmetic).  For single-process training: data-parallel
 * runs need the gradient materialised for the all-reduce between the two, so they call the two entry points separately.
 * params / exp_avg / exp_avg_sq hold n_total >= lay->n_params floats; entries beyond the layout (caller-appended parameters the
 * main loss does not touch) are stepped with a zero gradient, as pyro's shared optimizer does.  grads is still
 * written ([0, n_params)). */
int slode_elbo_adam_step(slode_handle h, const slode_shape* s, const slode_layout* lay, float* params, const float* times,
                         const float* stage_t, const float* obs, const int64_t obs_strides[3], const float* u, const float* eps,
                         float* loss_out, float* grads, void* workspace, size_t workspace_bytes, int64_t n_total, float* exp_avg,
                         float* exp_avg_sq, float lr, float beta1, float beta2, float adam_eps, int64_t step, void* stream);

/* One step of the reference's SECOND SVI object, SVI(model_meta, guide_meta) (training_cvs.py:244-249,152): encoder ->
 * group latents z_g = loc_g + scale_g * eps_g sampled in the model -> -[sum log N(z_g; loc_g, scale_g) + aux_mult * sum_heads
 * log p(label | head(z_g))], summed over the batch, and its exact gradient (encoder and label-head parameters; every other
 * entry of grads[0, n_params) is written as 0).  grads == NULL: loss only (evaluate_loss).  If exp_avg != NULL the Adam update
 * of all n_total >= n_params parameters is applied by the final reduction kernel (as slode_elbo_adam_step). */
int slode_aux_step(slode_handle h, const slode_shape* s, const slode_layout* lay, float* params, const float* obs,
                   const int64_t obs_strides[3], const float* u, const float* eps, float* loss_out, float* grads, void* workspace,
                   size_t workspace_bytes, int64_t n_total, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                   float adam_eps, int64_t step, void* stream);

/* ---- one SVI.step(**batch) as ONE call (training_cvs.py:147-157: `losses[i].step(**d)`) ------------------------------------------------
 * The minibatch as the reference's loader and batch_to_device hand it over (training_cvs.py:18-27, training_proc.py:25-33,
 * training_challenge.py:27-33): the observation tensor with its strides and the label tensors ONE BY ONE -- dense [B, width] each, in the
 * order in which the model concatenates them into u (cvs: iext, rtpr; proc: aR, aS, C12, C6; challenge: symptoms, shedding) -- no host-side
 * concatenation.  eps == NULL: the reparameterisation noise of the guide's sample sites (mechanistic_cvs.py:225-237 `pyro.sample(...,
 * dist.Normal(loc, scale).to_event(1))`; model_meta :256-262 for the auxiliary loss) is drawn INSIDE the kernels from the handle's
 * counter-based generator (slode_rng_seed); eps != NULL ([B, L], the explicit-noise parity path) is used as is. */
typedef struct slode_batch {
  const float* obs;            /* logical [B, C, T] */
  int64_t obs_strides[3];      /* element strides */
  int32_t n_labels;            /* 0: no labels (shapes without conditional priors / label heads) */
  int32_t label_width[SLODE_MAX_LABELS];
  const float* labels[SLODE_MAX_LABELS];
  const float* eps;            /* [B, L] or NULL */
} slode_batch;
/* Adam hyper-parameters and state for the update applied by the step's last kernel (NULL: gradient only) */
typedef struct slode_adam {
  int64_t n_total;             /* floats in params / exp_avg / exp_avg_sq (>= lay->n_params) */
  float *exp_avg, *exp_avg_sq;
  float lr, beta1, beta2, eps;
  int64_t step;                /* 1-based */
} slode_adam;
typedef enum slode_svi_kind { SLODE_SVI_MAIN = 0 /* SVI(model, guide) */, SLODE_SVI_AUX = 1 /* SVI(model_meta, guide_meta) */ } slode_svi_kind;
/* = slode_elbo_step / slode_elbo_adam_step (kind MAIN) or slode_aux_step (kind AUX) on a slode_batch.  grads == NULL: loss only
 * (SVI.evaluate_loss); adam != NULL: the update is applied by the final reduction kernel (grads must be given).  times / stage_t are
 * ignored for kind AUX. */
int slode_svi_step(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, float* params, const float* times,
                   const float* stage_t, const slode_batch* batch, float* loss_out, float* grads, void* workspace, size_t workspace_bytes,
                   const slode_adam* adam, void* stream);

/* ---- data parallel with the small payload (one collective per step) -----------------------------------------------------
 * The encoder's chain rule is linear in G = g_pre^T [X | 1] (and the head layers' gradients in glat^T [hid | 1]): a rank only has to
 * contribute its shard's G, its head-layer products and its ODE-half gradient row with the loss scalar --
 *   payload = [G: Hc x (C T + 1) | G_loc: L x (Hc + 1) | G_ls: L x (Hc + 1) | loss | gradient of flat range [ode_begin, n_params)]
 * (kind AUX: the row holds [loss | label-head range]), slode_grad_payload_floats floats (34,204 = 137 KB at BASELINE config[1] / [3]
 * against the 96,463 of [flat gradient | loss]).  One step on N ranks =
 *   slode_grad_partial (fold, [encoder,] ODE / aux kernel, split-K products, pack)  ->  SUM all-reduce of `payload` (RCCL; the caller's,
 *   torch.distributed.all_reduce in svi.py)  ->  slode_grad_apply (chain rule, final reduction, Adam) on every rank.
 * Both calls must use the SAME workspace with no other step on it in between (the fold's w' / row sums stay there).  Folded encoder
 * path only (dense [B,T,C] or [B,C,T] observations): otherwise SLODE_EINVAL, and the caller reduces the flat gradient of slode_svi_step
 * instead.  Replaces, like slode_svi_step, `losses[i].step(**d)` of training_cvs.py:152 on each rank of a data-parallel job. */
size_t slode_grad_payload_floats(const slode_shape* s, const slode_layout* lay, int kind);
int slode_grad_partial(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, const float* params, const float* times,
                       const float* stage_t, const slode_batch* batch, float* payload, void* workspace, size_t workspace_bytes, void* stream);
/* obs_strides: the batch's observation strides (they select the fold's column order).  loss_out[0] = the payload's loss slot (the global
 * -ELBO after the all-reduce); grads[0, n_params) written; adam != NULL: update applied by the same launch. */
int slode_grad_apply(slode_handle h, const slode_shape* s, const slode_layout* lay, int kind, float* params, const int64_t obs_strides[3],
                     const float* payload, float* loss_out, float* grads, void* workspace, size_t workspace_bytes, const slode_adam* adam,
                     void* stream);

/* Measured arm, OFF by default (SLODE_FOLD_NEXT=1 in the environment of slode_create turns it on): the launch that applies a step's Adam
 * update also folds the UPDATED encoder weights (W_eff & co.: csrc/encoder_fused.hip) and leaves them in the workspace, so consecutive
 * training steps on one (workspace, params) pair -- slode_svi_step / slode_elbo_adam_step / slode_aux_step / slode_grad_apply with Adam --
 * start without a fold launch.  Bitwise the same results; on MI355X the hand-offs inside the launch cost 6.0 us where the fold launch costs
 * 5.5, hence off.  When it is on, the workspace carries state from step to step and the library notices every weight change IT
 * makes; a caller that writes the parameter vector (or the workspace) itself between two steps -- checkpoint load, `load_state_dict`, its
 * own optimizer -- calls slode_fold_invalidate first.  With the arm off the call is a no-op.  The waits inside the launch are bounded: when
 * one fires, W_eff is NaN-poisoned and the next loss is NaN (one bench run of a B = 128 shape averaged 308 instead of 46 us per step --
 * profiles/r04_h_ab24_fold_next_all_configs.log); a diagnostic form, not for production runs. */
int slode_fold_invalidate(slode_handle h);

/* The handle's noise generator (replaces torch's global generator behind `rsample`): Philox-4x32-10 keyed by `seed`; the draw of
 * (drawing call n, trajectory b, latent index l) is word l & 3 -> Box-Muller of block [b + first_trajectory | l >> 2 | n] -- stateless,
 * so results do not depend on the grid or on how a global batch is sharded (data parallel: every rank passes the global index of its
 * shard's first trajectory).  slode_rng_seed resets the call counter n to 0; every step / evaluate call with eps == NULL uses the
 * current n and then increments it.  slode_rng_get reads (seed, first_trajectory, n) -- checkpoint / resume. */
int slode_rng_seed(slode_handle h, uint64_t seed, int64_t first_trajectory);
int slode_rng_set_counter(slode_handle h, uint64_t n);
int slode_rng_get(slode_handle h, uint64_t* seed, int64_t* first_trajectory, uint64_t* n);
/* The noise drawing call `n` would use for B trajectories of latent dim L, without running a step: eps_out[B, L] (NULL to skip) and the
 * raw Philox words raw_out[B, ceil(L / 4), 4] (uint32; NULL to skip).  Tests, and callers that want the explicit-eps path to reproduce
 * an in-kernel draw. */
int slode_rng_normal(slode_handle h, uint64_t n, int32_t B, int32_t L, float* eps_out, uint32_t* raw_out, void* stream);

/* torch.normal(loc, scale) of the eval-side callers (recon / classifier / pred_inputs: models/mechanistic_cvs.py:285, 300, 309):
 * z_out[B, L] = loc + scale * eps with eps from the handle's generator (one drawing call: uses the current counter n, then n + 1). */
int slode_sample_normal(slode_handle h, int32_t B, int32_t L, const float* loc, const float* scale, float* z_out, void* stream);

/* torch.optim.Adam step as pyro.optim.Adam applies it per parameter (training_cvs.py:226-227): in-place on flat
 * buffers.  step = 1-based step count. */
int slode_adam_step(slode_handle h, int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* Per-parameter step counts of pyro.optim.Adam (one torch.optim.Adam per parameter, state created at the first non-None gradient):
 * the two SVI objects of the reference share one optimizer (training_cvs.py:226-249) and alternate main, aux, main, ...; every
 * parameter is registered by both (pyro.module(..., self)), so all are stepped twice per minibatch -- except that the label heads of the
 * cvs / challenge families have no gradient yet in the very first main step and are skipped there.  Elements [lo, hi) of the flat
 * vector therefore use step + step_delta (skipped while that is < 1) in every Adam this handle applies (slode_adam_step,
 * slode_elbo_adam_step, slode_aux_step).  Default: empty region. */
int slode_adam_region(slode_handle h, int64_t lo, int64_t hi, int64_t step_delta);

/* Diagnostic (no reference counterpart; torchdiffeq does not report it): accepted steps per trajectory of the last dopri5 training
 * step run on this workspace -> counts[B] (int32, device).  -1: 20,000 attempted steps exhausted; > capacity: record overflow. */
int slode_dopri5_step_counts(slode_handle h, const slode_shape* s, const slode_layout* lay, const void* workspace,
                             size_t workspace_bytes, int* counts, void* stream);

/* Measurement aid for bench.py's roofline block (no reference counterpart).  on = 1: every kernel that slode_elbo_step /
 * slode_elbo_adam_step / slode_aux_step / slode_adam_step launch from now on carries its own start / stop event pair (hipExtLaunchKernelGGL):
 * the begin -> end device timestamps of that dispatch -- the duration rocprofv3 --kernel-trace reports for it -- without any extra
 * packet on `stream`; on = 0: off.  slode_profile_read waits for the kernels of the LAST such call on this handle and returns their
 * number n (<= max_kernels; a negative slode_status on error), their names (static strings: "weff", "enc_fwd2", "ode_elbo", "enc_bwd_lin",
 * "enc_chain", "dopri5_fwd", "dopri5_bwd", "aux", "enc_bwd2", "slab_stage1", "reduce", "adam", ...) in launch order and their durations in
 * microseconds. */
#define SLODE_PROFILE_MAX_KERNELS 16
int slode_profile_enable(slode_handle h, int on);
int slode_profile_read(slode_handle h, int max_kernels, const char** names, float* us);

#ifdef __cplusplus
}
#endif
#endif /* SLODE_H */
