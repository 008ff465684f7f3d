"""Host-side engine: owns the libslode handle, the parameter layout and the workspaces; every method is one call
through the C ABI (include/slode.h) on the caller's current HIP stream.  PyTorch tensors are storage only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L


@dataclass
class PriorGroup:
    """p(z_g | u_g): latent dims [z_off, z_off+z_dim), label columns [u_off, u_off+u_dim) of u.
    `prefix` is the reference attribute name of the EncoderMLP (e.g. 'p_z_iext_given_iext')."""
    prefix: str
    z_off: int
    z_dim: int
    u_off: int
    u_dim: int


@dataclass
class AuxHead:
    """A label head q(label | z_g) (auxiliary loss; also the main loss for the proc family): `prefix` = reference attribute
    (e.g. 'q_aR_given_z_aR'), kind in sigmoid | softmax | expexp; `std_key` = the scalar std parameter of an expexp head."""
    prefix: str
    kind: str
    z_off: int
    z_dim: int
    u_off: int
    u_dim: int
    std_key: str = ""


@dataclass
class ModelSpec:
    """Shape of one of the reference's three model families (models/mechanistic_{cvs,proc,challenge}[_Gauss].py)."""
    name: str
    gauss: bool
    n_channels: int
    latent_dim: int
    z_eps_dim: int
    n_u: int
    prior_groups: List[PriorGroup]
    ode_state_dim: int = 5
    ode_hidden_dim: int = 25
    n_filters: int = 10
    filter_size: int = 10
    pool_size: int = 5
    cnn_hidden_dim: int = 50
    solver: str = "midpoint"
    quantile_diff: float = 0.475
    aux_heads: List[AuxHead] = field(default_factory=list)
    labels_in_main: bool = False   # proc: the main model also scores the label heads (mechanistic_proc.py:145-146)
    u_hidden_dim: int = 25
    aux_mult: float = 46.0
    rtol: float = 1e-7      # dopri5 only (torchdiffeq defaults)
    atol: float = 1e-9
    # "exact": gradient of the discrete scheme (== adjoint_solver=False); "reference_adjoint": torchdiffeq.odeint_adjoint's backward,
    # the reference default (config.adjoint_solver = True; models/blackbox_ode.py:40-42) -- no gradient to z through the dynamics
    grad_mode: str = "exact"

    @property
    def head_names(self) -> List[str]:
        return ["output_mean"] if self.gauss else ["output_q50", "output_q75", "output_q25"]


def _check(lib, handle, rc):
    if rc != 0:
        msg = lib.slode_last_error(handle)
        raise L.SlodeError("libslode call failed (%d): %s" % (rc, msg.decode() if msg else "?"))


PROFILE_MAX_KERNELS = 16   # include/slode.h, SLODE_PROFILE_MAX_KERNELS


class Engine:
    def __init__(self, spec: ModelSpec, n_time: int, device: Optional[torch.device] = None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.SlodeError("no HIP device visible: the slode engine has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        if self.device.type != "cuda":
            raise L.SlodeError("the slode engine runs on a HIP device only, got %s" % self.device)
        self.spec, self.T = spec, int(n_time)
        self.handle = C.c_void_p()
        _check(self.lib, None, self.lib.slode_create(C.byref(self.handle), self.device.index or 0))
        self._shapes: Dict[int, L.Shape] = {}
        self.layout = L.Layout()
        _check(self.lib, None, self.lib.slode_layout_init(C.byref(self.shape(1)), C.byref(self.layout)))
        self.n_params = int(self.layout.n_params)
        self._ws: Dict[int, torch.Tensor] = {}
        self._stage_t: Optional[torch.Tensor] = None
        self._times: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.slode_destroy(self.handle)
        except Exception:
            pass

    # ---- shapes / layout ---------------------------------------------------------------------------------
    def shape(self, B: int) -> L.Shape:
        s = self._shapes.get(B)
        if s is None:
            sp = self.spec
            if sp.solver not in L.METHODS:
                raise ValueError("unknown solver %r" % sp.solver)
            s = L.Shape(B=B, T=self.T, C=sp.n_channels, L=sp.latent_dim, S=sp.ode_state_dim, H=sp.ode_hidden_dim,
                        F=sp.n_filters, K=sp.filter_size, P=sp.pool_size, Hc=sp.cnn_hidden_dim, n_u=sp.n_u,
                        n_groups=len(sp.prior_groups), method=L.METHODS[sp.solver],
                        likelihood=L.GAUSS if sp.gauss else L.ALD, quantile_diff=sp.quantile_diff, rtol=sp.rtol, atol=sp.atol,
                        n_aux=len(sp.aux_heads), U=sp.u_hidden_dim, aux_mult=sp.aux_mult, aux_in_main=int(sp.labels_in_main),
                        grad_mode=L.GRAD_MODES[sp.grad_mode])
            for i, a in enumerate(sp.aux_heads):
                s.aux[i] = L.Aux(L.AUX_KINDS[a.kind], a.z_off, a.z_dim, a.u_off, a.u_dim)
            for i, g in enumerate(sp.prior_groups):
                s.groups[i] = L.Group(g.z_off, g.z_dim, g.u_off, g.u_dim)
            self._shapes[B] = s
        return s

    def param_table(self) -> List[Tuple[str, int, Tuple[int, ...]]]:
        """(reference state_dict key, offset, shape) of every tensor of the flat layout, in layout order."""
        sp, lay, T = self.spec, self.layout, self.T
        C_, Ld, S, H, F, K, Hc = sp.n_channels, sp.latent_dim, sp.ode_state_dim, sp.ode_hidden_dim, sp.n_filters, sp.filter_size, sp.cnn_hidden_dim
        FQ = F * (T - K + 1 - sp.pool_size + 1)
        t = [("encoder.conv.weight", lay.conv_w, (F, C_, K)), ("encoder.conv.bias", lay.conv_b, (F,)),
             ("encoder.lin.weight", lay.lin_w, (Hc, FQ)), ("encoder.lin.bias", lay.lin_b, (Hc,)),
             ("encoder.z_loc.weight", lay.zloc_w, (Ld, Hc)), ("encoder.z_loc.bias", lay.zloc_b, (Ld,)),
             ("encoder.z_scale.0.weight", lay.zls_w, (Ld, Hc)), ("encoder.z_scale.0.bias", lay.zls_b, (Ld,))]
        for i, g in enumerate(sp.prior_groups):
            t += [(g.prefix + ".sequential_mlp.1.0.0.weight", lay.ploc_w[i], (g.z_dim, g.u_dim)),
                  (g.prefix + ".sequential_mlp.1.0.0.bias", lay.ploc_b[i], (g.z_dim,)),
                  (g.prefix + ".sequential_mlp.1.1.0.weight", lay.pls_w[i], (g.z_dim, g.u_dim)),
                  (g.prefix + ".sequential_mlp.1.1.0.bias", lay.pls_b[i], (g.z_dim,))]
        o = "decoder.ode_model."
        t += [(o + "latent_to_ode_net.0.weight", lay.init_w1, (H, Ld)), (o + "latent_to_ode_net.0.bias", lay.init_b1, (H,)),
              (o + "latent_to_ode_net.2.weight", lay.init_w2, (S, H)), (o + "latent_to_ode_net.2.bias", lay.init_b2, (S,)),
              (o + "dynamics.dynamics_hidden.weight", lay.dyn_wh, (H, 1 + Ld)), (o + "dynamics.dynamics_hidden.bias", lay.dyn_bh, (H,)),
              (o + "dynamics.dyanamics_growth.weight", lay.dyn_wg, (S, H)), (o + "dynamics.dyanamics_growth.bias", lay.dyn_bg, (S,)),
              (o + "dynamics.dyanmics_degradation.weight", lay.dyn_wd, (S, H)), (o + "dynamics.dyanmics_degradation.bias", lay.dyn_bd, (S,))]
        for i, hn in enumerate(sp.head_names):
            t.append(("decoder.%s.0.weight" % hn, lay.head_w[i], (C_, S)))
        U = sp.u_hidden_dim
        for i, a in enumerate(sp.aux_heads):
            t += [(a.prefix + ".sequential_mlp.1.module.weight", lay.aux_w1[i], (U, a.z_dim)),
                  (a.prefix + ".sequential_mlp.1.module.bias", lay.aux_b1[i], (U,))]
            if a.kind == "expexp":
                t += [(a.prefix + ".sequential_mlp.3.0.0.weight", lay.aux_w2[i], (a.u_dim, U)), (a.prefix + ".sequential_mlp.3.0.0.bias", lay.aux_b2[i], (a.u_dim,)),
                      (a.prefix + ".sequential_mlp.3.1.0.weight", lay.aux_w3[i], (a.u_dim, U)), (a.prefix + ".sequential_mlp.3.1.0.bias", lay.aux_b3[i], (a.u_dim,)),
                      (a.std_key, lay.aux_c[i], (1,))]
            else:
                t += [(a.prefix + ".sequential_mlp.3.weight", lay.aux_w2[i], (a.u_dim, U)), (a.prefix + ".sequential_mlp.3.bias", lay.aux_b2[i], (a.u_dim,))]
        t.append(("decoder.constant_std", lay.cstd, (C_, T)))
        return t

    def pack(self, params: Dict[str, torch.Tensor], flat: Optional[torch.Tensor] = None, extra: int = 0) -> torch.Tensor:
        """Copy a reference-keyed parameter dict into a flat device vector (n_params + extra floats)."""
        if flat is None:
            flat = torch.zeros(self.n_params + extra, dtype=torch.float32, device=self.device)
        for key, off, shp in self.param_table():
            n = 1
            for d in shp:
                n *= d
            flat[off:off + n].copy_(params[key].reshape(-1).to(torch.float32))
        return flat

    def unpack(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for key, off, shp in self.param_table():
            n = 1
            for d in shp:
                n *= d
            out[key] = flat[off:off + n].view(*shp)
        return out

    # ---- plumbing ----------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def _f32(self, t: torch.Tensor, name: str, contiguous: bool = True) -> torch.Tensor:
        if t.device != self.device or t.dtype != torch.float32:
            raise ValueError("%s must be a float32 tensor on %s (got %s on %s)" % (name, self.device, t.dtype, t.device))
        if contiguous and not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
        return t

    def _check_batch(self, obs, u, eps):
        """The kernels read raw device pointers: a CPU / float64 / strided label or noise tensor must not get that far."""
        B = obs.shape[0]
        self._f32(obs, "observations", contiguous=False)
        if u is not None:
            self._f32(u, "u")
            if tuple(u.shape) != (B, self.spec.n_u):
                raise ValueError("u must be [%d, %d], got %s" % (B, self.spec.n_u, tuple(u.shape)))
        self._f32(eps, "eps")
        if tuple(eps.shape) != (B, self.spec.latent_dim):
            raise ValueError("eps must be [%d, %d], got %s" % (B, self.spec.latent_dim, tuple(eps.shape)))

    def workspace(self, B: int) -> torch.Tensor:
        w = self._ws.get(B)
        if w is None:
            nbytes = int(self.lib.slode_workspace_bytes(self.handle, C.byref(self.shape(B))))
            if nbytes == 0:
                _check(self.lib, None, -1)
            w = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws[B] = w
        return w

    def set_times(self, times: torch.Tensor) -> torch.Tensor:
        """Bind the time grid (len T) and build the stage-time table on device."""
        times = self._f32(times.to(self.device, torch.float32).contiguous(), "times")
        if times.numel() != self.T:
            raise ValueError("times has %d points, engine built for T=%d" % (times.numel(), self.T))
        d = times[1:] - times[:-1]
        if not (bool((d > 0).all()) or bool((d < 0).all())):    # torchdiffeq odeint's own precondition (misc._check_timelike)
            raise ValueError("t must be strictly increasing or decreasing")
        if self.spec.solver == "dopri5" and not bool((d > 0).all()):
            # torchdiffeq integrates a decreasing grid in s = -t; the adaptive kernels here only walk forward in time
            # (dopri5_kernel.hip: dt > 0, outputs emitted while tj <= t1) and answer such a grid with NaN trajectories
            raise ValueError("solver='dopri5' needs a strictly increasing time grid (decreasing grids: fixed-grid solvers only)")
        n = int(self.lib.slode_num_stage_times(C.byref(self.shape(1))))
        st = torch.empty(n, dtype=torch.float32, device=self.device)
        _check(self.lib, self.handle, self.lib.slode_stage_times(self.handle, C.byref(self.shape(1)), self._p(times), self._p(st), self._stream()))
        self._times, self._stage_t = times, st
        return st

    def _obs_strides(self, obs: torch.Tensor):
        if obs.dim() != 3 or obs.shape[1] != self.spec.n_channels or obs.shape[2] != self.T:
            raise ValueError("observations must be [B, %d, %d], got %s" % (self.spec.n_channels, self.T, tuple(obs.shape)))
        return (C.c_int64 * 3)(*obs.stride())

    # ---- ops (one C-ABI call each) ---------------------------------------------------------------------------
    def encoder_fwd(self, params, obs, save: bool = True):
        B = obs.shape[0]
        sp = self.spec
        self._f32(obs, "observations", contiguous=False)
        FQ = sp.n_filters * (self.T - sp.filter_size + 1 - sp.pool_size + 1)
        loc = torch.empty(B, sp.latent_dim, dtype=torch.float32, device=self.device)
        scale = torch.empty_like(loc)
        pooled = torch.empty(B, FQ, dtype=torch.float32, device=self.device) if save else None
        hid = torch.empty(B, sp.cnn_hidden_dim, dtype=torch.float32, device=self.device) if save else None
        _check(self.lib, self.handle, self.lib.slode_encoder_conv_fwd(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(obs), self._obs_strides(obs),
            self._p(loc), self._p(scale), self._p(pooled), self._p(hid), self._stream()))
        return loc, scale, pooled, hid

    def encoder_bwd(self, params, obs, scale, pooled, hid, g_loc, g_scale, grads):
        B = obs.shape[0]
        ws = self.workspace(B)
        _check(self.lib, self.handle, self.lib.slode_encoder_conv_bwd(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(obs), self._obs_strides(obs),
            self._p(scale), self._p(pooled), self._p(hid), self._p(self._f32(g_loc, "g_loc")), self._p(self._f32(g_scale, "g_scale")),
            self._p(grads), self._p(ws), ws.numel() * 4, self._stream()))
        return grads

    def ode_solve(self, params, z):
        B = z.shape[0]
        x = torch.empty(B, self.T, self.spec.ode_state_dim, dtype=torch.float32, device=self.device)
        _check(self.lib, self.handle, self.lib.slode_ode_solve_fwd(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._times), self._p(self._stage_t),
            self._p(self._f32(z, "z")), self._p(x), self._stream()))
        return x

    def ode_solve_bwd(self, params, z, g_x, grads):
        B = z.shape[0]
        ws = self.workspace(B)
        g_z = torch.empty_like(z)
        _check(self.lib, self.handle, self.lib.slode_ode_solve_bwd(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._times), self._p(self._stage_t),
            self._p(self._f32(z, "z")), self._p(self._f32(g_x, "g_x")), self._p(g_z), self._p(grads), self._p(ws), ws.numel() * 4, self._stream()))
        return g_z

    def dynamics_eval(self, params, t: float, state, z):
        B = z.shape[0]
        out = torch.empty_like(state)
        _check(self.lib, self.handle, self.lib.slode_dynamics_eval(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), float(t), self._p(self._f32(state, "state")),
            self._p(self._f32(z, "z")), self._p(out), self._stream()))
        return out

    def initialize_state(self, params, z):
        """OdeModel.initialize_state: z [B, L] -> x0 [B, S] (one small HIP kernel, no solve)."""
        B = z.shape[0]
        x0 = torch.empty(B, self.spec.ode_state_dim, dtype=torch.float32, device=self.device)
        _check(self.lib, self.handle, self.lib.slode_initialize_state(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._f32(z, "z")), self._p(x0), self._stream()))
        return x0

    def prior_nets(self, params, u):
        """Conditional priors: u [B, n_u] -> (loc, scale) [B, L]; dims outside every conditional group: (0, 1)."""
        B = u.shape[0]
        loc = torch.empty(B, self.spec.latent_dim, dtype=torch.float32, device=self.device)
        scale = torch.empty_like(loc)
        _check(self.lib, self.handle, self.lib.slode_prior_nets(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._f32(u, "u")), self._p(loc), self._p(scale),
            self._stream()))
        return loc, scale

    def label_heads(self, params, z):
        """Label heads q(label | z_g): z [B, L] -> [B, n_u] probabilities / Laplace locations in the label columns they score."""
        B = z.shape[0]
        out = torch.zeros(B, self.spec.n_u, dtype=torch.float32, device=self.device)
        _check(self.lib, self.handle, self.lib.slode_label_heads(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._f32(z, "z")), self._p(out), self._stream()))
        return out

    def dopri5_step_counts(self, B: int) -> torch.Tensor:
        """Accepted steps per trajectory of the last dopri5 training step at batch size B (diagnostic; int32 [B])."""
        out = torch.empty(B, dtype=torch.int32, device=self.device)
        w = self.workspace(B)
        _check(self.lib, self.handle, self.lib.slode_dopri5_step_counts(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(w), w.numel() * 4, self._p(out), self._stream()))
        return out

    def decode_heads(self, params, x):
        B = x.shape[0]
        sp = self.spec
        Q = 1 if sp.gauss else 3
        mu = torch.empty(Q, B, sp.n_channels, self.T, dtype=torch.float32, device=self.device)
        std = torch.empty(sp.n_channels, self.T, dtype=torch.float32, device=self.device)
        _check(self.lib, self.handle, self.lib.slode_decode_heads(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._f32(x, "x")), self._p(mu), self._p(std), self._stream()))
        return mu, std

    def heads_snapshot(self, params):
        """Copy of the flat range [first decoder head, n_params) -- the decoder heads and constant_std (with the label heads, if any,
        between them) -- and its start: what decode_heads_bwd reads of the parameters, frozen at forward time."""
        lo = int(self.layout.head_w[0])
        return params[lo:self.n_params].clone(), lo

    def decode_heads_bwd(self, params, x, g_mu, g_std=None, snapshot=None):
        """Backward of decode_heads: g_mu [Q, B, C, T] (+ g_std [C, T]) -> (g_x [B, T, S], g_heads [Q, C, S], g_cstd [C, T]).
        `snapshot` = heads_snapshot(...) of the forward: the kernel then reads the head weights / constant_std from it (it touches no
        parameter below the first decoder head, so the base pointer is the snapshot's, moved back by its start)."""
        B = x.shape[0]
        sp = self.spec
        Q = 1 if sp.gauss else 3
        g_x = torch.empty(B, self.T, sp.ode_state_dim, dtype=torch.float32, device=self.device)
        g_heads = torch.empty(Q, sp.n_channels, sp.ode_state_dim, dtype=torch.float32, device=self.device)
        g_cstd = torch.empty(sp.n_channels, self.T, dtype=torch.float32, device=self.device)
        if snapshot is not None:
            snap, lo = snapshot
            if snap.numel() != self.n_params - lo or lo != int(self.layout.head_w[0]):
                raise ValueError("snapshot does not match this engine's layout")
            p_ptr = C.c_void_p(self._f32(snap, "snapshot").data_ptr() - 4 * lo)
        else:
            p_ptr = self._p(params)
        _check(self.lib, self.handle, self.lib.slode_decode_heads_bwd(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), p_ptr, self._p(self._f32(x, "x")), self._p(self._f32(g_mu, "g_mu")),
            self._p(self._f32(g_std, "g_std") if g_std is not None else None), self._p(g_x), self._p(g_heads), self._p(g_cstd), self._stream()))
        return g_x, g_heads, g_cstd

    def elbo_step(self, params, obs, u, eps, loss_out, grads=None, x_out=None, z_out=None):
        """-ELBO (summed over the batch) into loss_out[0]; exact gradient into grads (flat) unless grads is None."""
        B = obs.shape[0]
        self._check_batch(obs, u, eps)
        ws = self.workspace(B)
        self._guard(params, ws)
        _check(self.lib, self.handle, self.lib.slode_elbo_step(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._times), self._p(self._stage_t),
            self._p(obs), self._obs_strides(obs), self._p(u), self._p(eps), self._p(loss_out), self._p(grads), self._p(x_out), self._p(z_out),
            self._p(ws), ws.numel() * 4, self._stream()))
        return loss_out

    def elbo_adam_step(self, params, obs, u, eps, loss_out, grads, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), adam_eps=1e-8):
        """elbo_step + Adam with the update fused into the gradient reduction (single-process training)."""
        B = obs.shape[0]
        self._check_batch(obs, u, eps)
        ws = self.workspace(B)
        self._guard(params, ws)
        _check(self.lib, self.handle, self.lib.slode_elbo_adam_step(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(self._times), self._p(self._stage_t),
            self._p(obs), self._obs_strides(obs), self._p(u), self._p(eps), self._p(loss_out), self._p(grads), self._p(ws), ws.numel() * 4,
            params.numel(), self._p(exp_avg), self._p(exp_avg_sq), float(lr), float(betas[0]), float(betas[1]), float(adam_eps), int(step),
            self._stream()))
        return loss_out

    def aux_step(self, params, obs, u, eps, loss_out, grads=None, adam=None):
        """-ELBO of the auxiliary loss (model_meta) and its gradient; `adam` = (exp_avg, exp_avg_sq, lr, step, betas, eps) fuses
        the Adam update into the final reduction."""
        B = obs.shape[0]
        self._check_batch(obs, u, eps)
        ws = self.workspace(B)
        self._guard(params, ws)
        m, v, lr, step, betas, aeps = adam if adam is not None else (None, None, 0.0, 1, (0.9, 0.999), 1e-8)
        _check(self.lib, self.handle, self.lib.slode_aux_step(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), self._p(params), self._p(obs), self._obs_strides(obs), self._p(u),
            self._p(eps), self._p(loss_out), self._p(grads), self._p(ws), ws.numel() * 4, params.numel(), self._p(m), self._p(v), float(lr),
            float(betas[0]), float(betas[1]), float(aeps), int(step), self._stream()))
        return loss_out

    # ---- SVI.step(**batch) as one call: labels as the loader yields them, noise drawn in the kernels -----------------------------
    def make_batch(self, obs, labels, eps=None) -> L.Batch:
        """slode_batch for `obs` [B, C, T] (any strides) and the label tensors in the model's concatenation order (each [B, width] or [B],
        float32, contiguous, on the device).  eps [B, L] or None (None: drawn in-kernel from the handle's Philox stream, rng_seed)."""
        B = obs.shape[0]
        self._f32(obs, "observations", contiguous=False)
        bt = L.Batch()
        bt.obs = obs.data_ptr()
        st = self._obs_strides(obs)
        for i in range(3):
            bt.obs_strides[i] = st[i]
        if len(labels) > L.MAX_LABELS:
            raise ValueError("at most %d label tensors, got %d" % (L.MAX_LABELS, len(labels)))
        cols = 0
        for i, t in enumerate(labels):
            self._f32(t, "label %d" % i)
            if t.shape[0] != B:
                raise ValueError("label %d has %d rows, the batch %d" % (i, t.shape[0], B))
            w = t.numel() // B
            bt.labels[i], bt.label_width[i] = t.data_ptr(), w
            cols += w
        if labels and cols != self.spec.n_u:
            raise ValueError("the label tensors have %d columns in all, the model's u has %d" % (cols, self.spec.n_u))
        bt.n_labels = len(labels)
        if eps is not None:
            self._f32(eps, "eps")
            if tuple(eps.shape) != (B, self.spec.latent_dim):
                raise ValueError("eps must be [%d, %d], got %s" % (B, self.spec.latent_dim, tuple(eps.shape)))
            bt.eps = eps.data_ptr()
        return bt

    def svi_step(self, kind: int, params, batch: L.Batch, B: int, loss_out, grads=None, adam=None):
        """slode_svi_step: kind L.SVI_MAIN | L.SVI_AUX; adam = (exp_avg, exp_avg_sq, lr, step, betas, eps) or None."""
        ws = self.workspace(B)
        self._guard(params, ws)
        ad = None
        if adam is not None:
            m, v, lr, step, betas, aeps = adam
            ad = L.AdamArgs(params.numel(), m.data_ptr(), v.data_ptr(), float(lr), float(betas[0]), float(betas[1]), float(aeps), int(step))
        _check(self.lib, self.handle, self.lib.slode_svi_step(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), int(kind), self._p(params), self._p(self._times), self._p(self._stage_t),
            C.byref(batch), self._p(loss_out), self._p(grads), self._p(ws), ws.numel() * 4, C.byref(ad) if ad is not None else None, self._stream()))
        return loss_out

    # ---- data parallel with the small payload: grad_partial -> all-reduce(payload) -> grad_apply (include/slode.h) ------------------
    def payload_floats(self, kind: int) -> int:
        return int(self.lib.slode_grad_payload_floats(C.byref(self.shape(1)), C.byref(self.layout), int(kind)))

    def grad_partial(self, kind: int, params, batch: L.Batch, B: int, payload):
        ws = self.workspace(B)
        self._guard(params, ws)
        _check(self.lib, self.handle, self.lib.slode_grad_partial(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), int(kind), self._p(params), self._p(self._times), self._p(self._stage_t),
            C.byref(batch), self._p(self._f32(payload, "payload")), self._p(ws), ws.numel() * 4, self._stream()))

    def grad_apply(self, kind: int, params, batch: L.Batch, B: int, payload, loss_out, grads, adam=None):
        ws = self.workspace(B)
        ad = None
        if adam is not None:
            m, v, lr, step, betas, aeps = adam
            ad = L.AdamArgs(params.numel(), m.data_ptr(), v.data_ptr(), float(lr), float(betas[0]), float(betas[1]), float(aeps), int(step))
        _check(self.lib, self.handle, self.lib.slode_grad_apply(
            self.handle, C.byref(self.shape(B)), C.byref(self.layout), int(kind), self._p(params), batch.obs_strides, self._p(payload),
            self._p(loss_out), self._p(grads), self._p(ws), ws.numel() * 4, C.byref(ad) if ad is not None else None, self._stream()))
        return loss_out

    def fold_invalidate(self):
        """Tell the engine that the parameter vector was written outside its own steps (checkpoint load, another optimizer): the next step
        folds the encoder weights again instead of trusting the fold the previous step left in the workspace (slode_fold_invalidate)."""
        _check(self.lib, self.handle, self.lib.slode_fold_invalidate(self.handle))

    def _guard(self, params, ws):
        """torch-side writes to the flat vector itself, or to the workspace of this call (it carries the kept fold), bump the tensor's
        version counter: such a write between two steps invalidates the kept fold.  (Writes through re-pointed nn.Parameters do not show
        here: models call fold_invalidate from load_state_dict.)"""
        pk = (params.data_ptr(), params._version)
        wk = ws.data_ptr()
        if (getattr(self, "_pk", None) not in (None, pk)) or getattr(self, "_wv", {}).get(wk, ws._version) != ws._version:
            self.fold_invalidate()
        self._pk = pk
        if not hasattr(self, "_wv"):
            self._wv = {}
        self._wv[wk] = ws._version

    def rng_seed(self, seed: int, first_trajectory: int = 0):
        """Key of the in-kernel noise generator (Philox-4x32-10); resets its call counter.  Data parallel: every rank passes the global
        index of its shard's first trajectory, so the draws do not depend on the sharding."""
        _check(self.lib, self.handle, self.lib.slode_rng_seed(self.handle, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_trajectory)))

    def rng_state(self):
        """(seed, first_trajectory, calls drawn so far)."""
        a, b, c = C.c_uint64(), C.c_int64(), C.c_uint64()
        _check(self.lib, self.handle, self.lib.slode_rng_get(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def rng_set_counter(self, n: int):
        _check(self.lib, self.handle, self.lib.slode_rng_set_counter(self.handle, int(n)))

    def rng_normal(self, n: int, B: int, raw: bool = False):
        """The noise drawing call `n` uses for B trajectories: eps [B, L] (and the raw Philox words [B, ceil(L/4), 4] as int64 if raw)."""
        Ld = self.spec.latent_dim
        eps = torch.empty(B, Ld, dtype=torch.float32, device=self.device)
        words = torch.empty(B, (Ld + 3) // 4, 4, dtype=torch.int32, device=self.device) if raw else None
        _check(self.lib, self.handle, self.lib.slode_rng_normal(self.handle, int(n), B, Ld, self._p(eps), self._p(words), self._stream()))
        return (eps, words.to(torch.int64) & 0xFFFFFFFF) if raw else eps

    def draw_normal(self, B: int):
        """eps [B, L] of the next drawing call of the engine's generator (the call counter moves on)."""
        _, _, n = self.rng_state()
        eps = self.rng_normal(n, B)
        self.rng_set_counter(n + 1)
        return eps

    def sample_normal(self, loc, scale):
        """torch.normal(loc, scale) on the engine's generator: z = loc + scale * eps, one HIP kernel, one drawing call."""
        z = torch.empty_like(self._f32(loc, "loc"))
        _check(self.lib, self.handle, self.lib.slode_sample_normal(self.handle, loc.shape[0], loc.shape[1], self._p(loc),
                                                                   self._p(self._f32(scale, "scale")), self._p(z), self._stream()))
        return z

    def adam_step(self, params, grads, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-8):
        _check(self.lib, self.handle, self.lib.slode_adam_step(
            self.handle, params.numel(), self._p(params), self._p(grads), self._p(exp_avg), self._p(exp_avg_sq),
            float(lr), float(betas[0]), float(betas[1]), float(eps), int(step), self._stream()))


    def adam_region(self, lo: int, hi: int, step_delta: int):
        """Flat-vector elements [lo, hi) use Adam step count `step + step_delta` (skipped while < 1): include/slode.h, slode_adam_region."""
        _check(self.lib, self.handle, self.lib.slode_adam_region(self.handle, int(lo), int(hi), int(step_delta)))

    def aux_only_region(self):
        """[lo, hi) of the label-head parameters when only the auxiliary loss uses them (cvs / challenge), else (0, 0)."""
        lay, sp = self.layout, self.spec
        if not sp.aux_heads or sp.labels_in_main:
            return 0, 0
        return int(lay.aux_w1[0]), int(lay.cstd)

    def profile_enable(self, on: bool):
        """Per-kernel device timestamps for the step entry points (include/slode.h, slode_profile_enable)."""
        _check(self.lib, self.handle, self.lib.slode_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self) -> List[Tuple[str, float]]:
        """[(kernel name, microseconds)] of the last profiled step call on this engine, in launch order."""
        names = (C.c_char_p * PROFILE_MAX_KERNELS)()
        us = (C.c_float * PROFILE_MAX_KERNELS)()
        n = self.lib.slode_profile_read(self.handle, PROFILE_MAX_KERNELS, names, us)
        if n < 0:
            _check(self.lib, self.handle, n)
        return [(names[i].decode(), float(us[i])) for i in range(n)]


def cvs_spec(z_iext=5, z_rtpr=5, z_eps=5, gauss=False, solver="midpoint", quantile_diff=0.475) -> ModelSpec:
    """data/cvs/config_cvs.py:6-52; u = [iext, rtpr] columns (models/mechanistic_cvs.py:131-135)."""
    return ModelSpec("cvs", gauss, 3, z_iext + z_rtpr + z_eps, z_eps, 2,
                     [PriorGroup("p_z_iext_given_iext", 0, z_iext, 0, 1), PriorGroup("p_z_rtprs_given_rtprs", z_iext, z_rtpr, 1, 1)],
                     solver=solver, quantile_diff=quantile_diff,
                     aux_heads=[AuxHead("q_iext_given_z_iext", "sigmoid", 0, z_iext, 0, 1),
                                AuxHead("q_rtpr_given_z_rtpr", "sigmoid", z_iext, z_rtpr, 1, 1)])


def challenge_spec(z_shed=5, z_symp=5, z_eps=5, gauss=False, solver="midpoint", quantile_diff=0.475) -> ModelSpec:
    """data/challenge/config_challenge.py; u = cat(symptoms, shedding) (models/mechanistic_challenge.py:167)."""
    return ModelSpec("challenge", gauss, 4, z_shed + z_symp + z_eps, z_eps, 2, [PriorGroup("p_z_u_given_u", 0, z_shed + z_symp, 0, 2)],
                     solver=solver, quantile_diff=quantile_diff,
                     aux_heads=[AuxHead("q_shedding_given_z_shedding", "sigmoid", 0, z_shed, 1, 1),
                                AuxHead("q_symptom_given_z_symptom", "sigmoid", z_shed, z_symp, 0, 1)])


def proc_spec(z_g=10, z_eps=10, gauss=False, solver="midpoint", quantile_diff=0.475) -> ModelSpec:
    """data/proc/config_proc.py; u = cat(aR[3], aS[4], C12, C6) (models/mechanistic_proc.py:196-198)."""
    aux = [AuxHead("q_aR_given_z_aR", "softmax", 0, z_g, 0, 3), AuxHead("q_aS_given_z_aS", "softmax", z_g, z_g, 3, 4),
           AuxHead("q_C12_given_z_C12", "expexp", 2 * z_g, z_g, 7, 1, "constant_std_C_12"),
           AuxHead("q_C6_given_z_C6", "expexp", 3 * z_g, z_g, 8, 1, "constant_std_C_6")]
    return ModelSpec("proc", gauss, 4, 4 * z_g + z_eps, z_eps, 9, [PriorGroup("p_z_u_given_u", 0, 4 * z_g, 0, 9)],
                     ode_state_dim=8, solver=solver, quantile_diff=quantile_diff, aux_heads=aux, labels_in_main=True)
