"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (eager PyTorch, fp32 by default, fp64 on request) of the reference's latent-ODE solve +
ELBO path, mirroring the reference op for op.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker / the reported CPU
baseline -- never as something the product dispatches to.  The product path (``structured_latent_odes_amd``)
calls ``libslode.so`` (HIP, gfx950) and raises if that library is missing.

Pinning status
--------------
* Pieces of the reference that import in the build container (``EncoderCONV``, ``EncoderMLP``,
  ``Dynamics``/``OdeFunc``, ``OdeModel.initialize_state``, decoder heads + softplus std) are pinned by golden
  vectors generated from the reference modules themselves: ``tests/golden/make_golden.py`` ->
  ``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).
* ``torchdiffeq`` (integrator arithmetic; not vendored, version not pinned by the reference) and ``pyro-ppl``
  1.9.0 (ELBO assembly) are absent from the container and from ``/root/reference``.  Their published
  algorithms are restated here (fixed-grid euler / midpoint / 3/8-rule rk4, adaptive dopri5; Trace_ELBO =
  sum of scaled log-probs) and are pinned only by analytic known-answer tests
  (``tests/test_oracle_kat.py``): **parity unpinned** at those two boundaries (the reference has no tests).

Every function cites the reference file:line it follows (paths relative to the reference repo root).
All parameter dictionaries use the reference ``state_dict`` key names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch.distributions import Bernoulli, Laplace, Normal, OneHotCategorical

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------------------
# L2 modules
# ----------------------------------------------------------------------------------------------------------
def encoder_conv(p: Params, x: Tensor, pool_size: int, prefix: str = "encoder.") -> Tuple[Tensor, Tensor]:
    """models/encoder_conv.py:43-51 -- conv -> avgpool(stride 1) -> flatten -> lin -> tanh -> (z_loc, exp(z_scale)).

    x is the logical [B, C, T] tensor (may be a permuted view, training_cvs.py:25)."""
    h = F.conv1d(x, p[prefix + "conv.weight"], p[prefix + "conv.bias"])          # :44
    h = F.avg_pool1d(h, pool_size, stride=1)                                      # :45
    h = h.reshape(h.size(0), -1)                                                  # :46 (filter-major flatten)
    h = F.linear(h, p[prefix + "lin.weight"], p[prefix + "lin.bias"])             # :47
    h = torch.tanh(h)                                                             # :48
    z_loc = F.linear(h, p[prefix + "z_loc.weight"], p[prefix + "z_loc.bias"])     # :49
    z_scale = torch.exp(F.linear(h, p[prefix + "z_scale.0.weight"], p[prefix + "z_scale.0.bias"]))  # :50
    return z_loc, z_scale


def mlp_hidden_softplus(p: Params, prefix: str, x: Tensor) -> Tensor:
    """models/encoder_mlp.py:88-110 -- the one hidden Linear (DataParallel-wrapped => '.module.') + Softplus."""
    return F.softplus(F.linear(x, p[prefix + "sequential_mlp.1.module.weight"],
                               p[prefix + "sequential_mlp.1.module.bias"]))


def prior_net(p: Params, prefix: str, u: Tensor) -> Tuple[Tensor, Tensor]:
    """EncoderMLP(mlp_sizes=[n_u, [d, d]], output_activation=[None, Exp]) -- models/encoder_mlp.py:134-160;
    used as p(z_u | u): models/mechanistic_cvs.py:88-100, mechanistic_proc.py:107-114,
    mechanistic_challenge.py:88-95.  No hidden layer; two Linear heads -> (loc, exp(.))."""
    loc = F.linear(u, p[prefix + "sequential_mlp.1.0.0.weight"], p[prefix + "sequential_mlp.1.0.0.bias"])
    scale = torch.exp(F.linear(u, p[prefix + "sequential_mlp.1.1.0.weight"],
                               p[prefix + "sequential_mlp.1.1.0.bias"]))
    return loc, scale


def classifier_sigmoid(p: Params, prefix: str, z: Tensor) -> Tensor:
    """EncoderMLP([d, U, k], Softplus, Sigmoid) -- mechanistic_cvs.py:66-80, mechanistic_challenge.py:67-80."""
    h = mlp_hidden_softplus(p, prefix, z)
    return torch.sigmoid(F.linear(h, p[prefix + "sequential_mlp.3.weight"], p[prefix + "sequential_mlp.3.bias"]))


def classifier_softmax(p: Params, prefix: str, z: Tensor) -> Tensor:
    """EncoderMLP([d, U, k], Softplus, Softmax(dim=1)) -- mechanistic_proc.py:67-80, encoder_mlp.py:15-16."""
    h = mlp_hidden_softplus(p, prefix, z)
    return torch.softmax(F.linear(h, p[prefix + "sequential_mlp.3.weight"], p[prefix + "sequential_mlp.3.bias"]), dim=1)


def regressor_exp_exp(p: Params, prefix: str, z: Tensor) -> Tuple[Tensor, Tensor]:
    """EncoderMLP([d, U, [1, 1]], Softplus, [Exp, Exp]) -- mechanistic_proc.py:82-99."""
    h = mlp_hidden_softplus(p, prefix, z)
    a = torch.exp(F.linear(h, p[prefix + "sequential_mlp.3.0.0.weight"], p[prefix + "sequential_mlp.3.0.0.bias"]))
    b = torch.exp(F.linear(h, p[prefix + "sequential_mlp.3.1.0.weight"], p[prefix + "sequential_mlp.3.1.0.bias"]))
    return a, b


# ----------------------------------------------------------------------------------------------------------
# L1 ODE core
# ----------------------------------------------------------------------------------------------------------
_ODE = "decoder.ode_model."


def initialize_state(p: Params, z: Tensor, prefix: str = _ODE) -> Tensor:
    """models/blackbox_ode.py:19-22,32-34 -- x0 = sigmoid(W2 relu(W1 z + b1) + b2)."""
    h = torch.relu(F.linear(z, p[prefix + "latent_to_ode_net.0.weight"], p[prefix + "latent_to_ode_net.0.bias"]))
    return torch.sigmoid(F.linear(h, p[prefix + "latent_to_ode_net.2.weight"], p[prefix + "latent_to_ode_net.2.bias"]))


def dynamics(p: Params, t: Tensor, state: Tensor, z: Tensor, prefix: str = _ODE) -> Tensor:
    """models/blackbox_ode.py:97-109 (Dynamics.forward) via OdeFunc.forward :57-61.

    Mirrors the reference op for op: t.repeat -> cat([t, z]) -> hidden Linear + ReLU evaluated twice
    (prod and degr share the hidden layer object, :84-95) -> two sigmoid heads -> xa - xd * state."""
    n_batch = z.shape[0]
    t_expanded = t.repeat([n_batch, 1])                                            # :99
    x = torch.cat([t_expanded, z], dim=1)                                          # :101 (t is column 0)
    wh, bh = p[prefix + "dynamics.dynamics_hidden.weight"], p[prefix + "dynamics.dynamics_hidden.bias"]
    xa = torch.sigmoid(F.linear(torch.relu(F.linear(x, wh, bh)),
                                p[prefix + "dynamics.dyanamics_growth.weight"],
                                p[prefix + "dynamics.dyanamics_growth.bias"]))     # :106 prod
    xd = torch.sigmoid(F.linear(torch.relu(F.linear(x, wh, bh)),
                                p[prefix + "dynamics.dyanmics_degradation.weight"],
                                p[prefix + "dynamics.dyanmics_degradation.bias"]))  # :107 degr
    return xa - xd * state                                                         # :108


# --- integrator: restatement of torchdiffeq (third-party, absent; see module docstring) -------------------
_ONE_THIRD = 1.0 / 3.0
_TWO_THIRDS = 2.0 / 3.0


def _step_euler(f, t0, dt, t1, y0):
    return dt * f(t0, y0)


def _step_midpoint(f, t0, dt, t1, y0):
    half_dt = 0.5 * dt
    y_mid = y0 + f(t0, y0) * half_dt
    return dt * f(t0 + half_dt, y_mid)


def _step_rk4_38(f, t0, dt, t1, y0):
    """torchdiffeq's 'rk4' is the 3/8-rule variant (rk4_alt_step_func), not classic RK4."""
    k1 = f(t0, y0)
    k2 = f(t0 + dt * _ONE_THIRD, y0 + dt * k1 * _ONE_THIRD)
    k3 = f(t0 + dt * _TWO_THIRDS, y0 + dt * (k2 - k1 * _ONE_THIRD))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3))
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


_FIXED_STEPS = {"euler": _step_euler, "midpoint": _step_midpoint, "rk4": _step_rk4_38}


def odeint_fixed(f: Callable[[Tensor, Tensor], Tensor], y0: Tensor, times: Tensor, method: str) -> Tensor:
    """torchdiffeq FixedGridODESolver.integrate with step_size=None: the grid IS ``times``; one step per
    output interval; solution[0] = y0.  Call site: models/blackbox_ode.py:41-45.  Returns [T, B, S]."""
    step = _FIXED_STEPS[method]
    sol = [y0]
    y = y0
    for n in range(times.shape[0] - 1):
        t0, t1 = times[n], times[n + 1]
        dt = t1 - t0
        y = y + step(f, t0, dt, t1, y)
        sol.append(y)
    return torch.stack(sol, dim=0)


_DYN_KEYS = ("dynamics.dynamics_hidden.weight", "dynamics.dynamics_hidden.bias", "dynamics.dyanamics_growth.weight",
             "dynamics.dyanamics_growth.bias", "dynamics.dyanmics_degradation.weight", "dynamics.dyanmics_degradation.bias")


class _OdeintAdjoint(torch.autograd.Function):
    """Restatement of ``torchdiffeq.odeint_adjoint`` (third-party, absent; version unpinned => parity unpinned) for the fixed-grid
    solvers, as the reference calls it by default (models/blackbox_ode.py:40-42; ``adjoint_solver = True`` in all three
    configs).  Forward: plain ``odeint`` under no_grad.  Backward: for i = T-1 .. 1 the augmented state (y, a_y, a_params) is
    integrated from t[i] to t[i-1] with the SAME fixed-grid method (one step: the grid is ``times``), where
        d a_y / dt = -a_y^T df/dy,   d a_params / dt = -a_y^T df/dparams,
    then y is reset to the stored forward value y[i-1] and a_y += grad_y[i-1].  ``adjoint_params = tuple(func.parameters())``:
    the dynamics net only -- ``OdeFunc.constants`` (= z) is a plain tensor (blackbox_ode.py:55), so NO gradient reaches z through
    the dynamics in this mode (SURVEY hard part 2); z still gets gradient through y0."""

    @staticmethod
    def forward(ctx, fb, y0, times, method, *params):
        ctx.fb, ctx.method = fb, method
        with torch.no_grad():
            y = odeint_fixed(fb(params), y0, times, method)
        ctx.save_for_backward(times, y, *params)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        times, y, *params = ctx.saved_tensors
        fb, step = ctx.fb, _FIXED_STEPS[ctx.method]

        class Aug:  # tuple state with the arithmetic the step functions use
            def __init__(self, parts): self.parts = tuple(parts)
            def __add__(self, o): return Aug(a + b for a, b in zip(self.parts, o.parts))
            def __sub__(self, o): return Aug(a - b for a, b in zip(self.parts, o.parts))
            def __mul__(self, c): return Aug(a * c for a in self.parts)
            __rmul__ = __mul__

        def aug_dyn(t, st):
            yy, a = st.parts[0], st.parts[1]
            with torch.enable_grad():
                yy_ = yy.detach().requires_grad_(True)
                pr = [q.detach().requires_grad_(True) for q in params]
                fe = fb(pr)(t, yy_)
                vj = torch.autograd.grad(fe, [yy_] + pr, -a, allow_unused=True)
            vj = [v if v is not None else torch.zeros_like(x) for v, x in zip(vj, [yy_] + pr)]
            return Aug([fe.detach(), vj[0]] + vj[1:])

        with torch.no_grad():
            a_y = grad_y[-1].clone()
            a_p = [torch.zeros_like(q) for q in params]
            for i in range(times.shape[0] - 1, 0, -1):
                t0, t1 = times[i], times[i - 1]
                st = Aug([y[i], a_y] + a_p)
                st = st + step(aug_dyn, t0, t1 - t0, t1, st)
                a_y = st.parts[1] + grad_y[i - 1]
                a_p = list(st.parts[2:])
        return (None, a_y, None, None, *a_p)


def stage_times(times: Tensor, method: str) -> Tensor:
    """The distinct times at which a fixed-grid method evaluates f, in evaluation order, computed with the
    same fp32 arithmetic as the step functions above (t0 + dt*c).  Layout: R entries per step
    (euler R=1: t0; midpoint R=2: t0, t0+dt/2; rk4 R=3: t0, t0+dt/3, t0+2dt/3) followed by times[-1]
    (rk4's k4 time of the last step; present for every method so the table length is R*(T-1)+1)."""
    t0 = times[:-1]
    dt = times[1:] - t0
    if method == "euler":
        cols = [t0]
    elif method == "midpoint":
        cols = [t0, t0 + 0.5 * dt]
    elif method == "rk4":
        cols = [t0, t0 + dt * _ONE_THIRD, t0 + dt * _TWO_THIRDS]
    else:
        raise ValueError(method)
    tab = torch.stack(cols, dim=1).reshape(-1)
    return torch.cat([tab, times[-1:]])


# Dormand-Prince 5(4) tableau as used by torchdiffeq's Dopri5Solver.
_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_CSOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP_CERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP_CMID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def odeint_dopri5(f, y0: Tensor, times: Tensor, rtol: float = 1e-7, atol: float = 1e-9,
                  per_trajectory: bool = False, max_steps: int = 100000) -> Tensor:
    """Restatement of torchdiffeq's adaptive Dopri5 (RKAdaptiveStepsizeODESolver): FSAL, mixed
    error norm rms(err / (atol + rtol*max(|y0|,|y1|))), step factor clamp [0.2, 10], safety 0.9,
    Hairer initial step, 4th-order dense interpolation through (y0, y_mid, y1, f0, f1) to the output times.

    ``per_trajectory=False`` follows upstream (one step size for the whole [B,S] tensor; batch-coupled).
    ``per_trajectory=True`` runs one controller per batch row -- the shardable contract used for the
    proc config (SURVEY hard part 3); both agree to solver tolerance."""
    dtype = y0.dtype
    B = y0.shape[0]

    def norm(x):  # rms over the state (per row or over the whole tensor)
        if per_trajectory:
            return x.pow(2).mean(dim=1).sqrt()               # [B]
        return x.pow(2).mean().sqrt().reshape(1)             # [1]

    def bc(v):  # broadcast a per-row (or global) scalar over the state dim
        return v.reshape(-1, 1)

    t0 = times[0].to(dtype)
    f0 = f(t0, y0)
    # Hairer initial step (order 4 estimate, q = 4 => exponent 1/5)
    scale = atol + y0.abs() * rtol
    d0, d1 = norm(y0 / scale), norm(f0 / scale)
    h0 = torch.where((d0 < 1e-5) | (d1 < 1e-5), torch.full_like(d0, 1e-6), 0.01 * d0 / d1)
    y1 = y0 + bc(h0) * f0
    f1 = f_rows(f, t0 + h0, y1, per_trajectory)
    d2 = norm((f1 - f0) / scale) / h0
    h1 = torch.where((d1 <= 1e-15) & (d2 <= 1e-15), torch.maximum(torch.full_like(h0, 1e-6), h0 * 1e-3),
                     (0.01 / torch.maximum(d1, d2)) ** (1.0 / 5.0))
    # the controller is not differentiated: step sizes are data of the discrete solve (differentiating ratio ** -0.2 through
    # rejected / floor-limited steps is meaningless and overflows); gradients are those of the accepted steps + dense output
    dt = torch.minimum(100 * h0, h1).detach()

    n_ctl = dt.shape[0]
    t = t0.expand(n_ctl).clone()
    y, fy = y0.clone(), f0.clone()
    interp = None  # (t0, t1, y0, ymid, y1, f0, f1) of the last accepted step
    out = [y0]
    # state kept per controller
    t_prev = t.clone()
    y_prev, y_mid, f_prev = y.clone(), y.clone(), fy.clone()
    for j in range(1, times.shape[0]):
        tj = times[j].to(dtype)
        steps = 0
        while bool((t < tj).any()):
            steps += 1
            if steps > max_steps:
                raise RuntimeError("dopri5: max_steps exceeded")
            active = t < tj                                   # controllers that still need to advance
            ks = [fy]
            for i, (a, brow) in enumerate(zip(_DP_ALPHA, _DP_BETA)):
                yi = y + bc(dt) * sum(b * k for b, k in zip(brow, ks))
                ti = t + a * dt if a != 1.0 else t + dt
                ks.append(f_rows(f, ti, yi, per_trajectory))
            y_new = y + bc(dt) * sum(c * k for c, k in zip(_DP_CSOL, ks))
            err = bc(dt) * sum(c * k for c, k in zip(_DP_CERR, ks))
            tol = atol + rtol * torch.maximum(y.abs(), y_new.abs())
            ratio = norm(err / tol)
            accept = (ratio <= 1) & active
            ymid_new = y + bc(dt) * sum(c * k for c, k in zip(_DP_CMID, ks))
            acc = bc(accept) if per_trajectory else accept
            # record interpolation data for accepted controllers
            t_prev = torch.where(accept, t, t_prev)
            y_prev = torch.where(acc, y, y_prev)
            y_mid = torch.where(acc, ymid_new, y_mid)
            f_prev = torch.where(acc, fy, f_prev)
            t = torch.where(accept, t + dt, t)
            y = torch.where(acc, y_new, y)
            fy = torch.where(acc, ks[-1], fy)
            # step-size update (only for controllers that took part)
            safe = torch.where(ratio == 0, torch.full_like(ratio, 10.0),
                               0.9 * ratio.clamp_min(1e-300) ** (-1.0 / 5.0))
            factor = torch.where(ratio < 1, safe.clamp(1.0, 10.0), safe.clamp(0.2, 10.0))   # dfactor = 1 iff ratio < 1 (torchdiffeq)
            factor = torch.where(ratio == 0, torch.full_like(ratio, 10.0), factor)
            dt = torch.where(active, dt * factor, dt).detach()
        # dense output at tj through the last accepted step of each controller
        out.append(_dopri5_interp(bc(t_prev), bc(t), y_prev, y_mid, y, f_prev, fy, tj))
    return torch.stack(out, dim=0)


def f_rows(f, t, y, per_trajectory):
    """Evaluate f at a per-row time vector (per_trajectory) or at a single time."""
    if not per_trajectory:
        return f(t.reshape(()), y)
    return f(t.reshape(-1, 1), y)


def _dopri5_interp(t0, t1, y0, ymid, y1, f0, f1, t):
    """torchdiffeq _interp_fit + _interp_evaluate (quartic through y0, ymid, y1 with end slopes)."""
    dt = t1 - t0
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * ymid
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * ymid
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * ymid
    d = dt * f0
    e = y0
    x = (t - t0) / dt
    return e + x * (d + x * (c + x * (b + x * a)))


def solve_ode(p: Params, z: Tensor, times: Tensor, method: str, prefix: str = _ODE, grad_mode: str = "exact", **kw) -> Tensor:
    """models/blackbox_ode.py:36-47 (OdeModel.solve_ODE) -> [B, T, S].  grad_mode "exact": autograd through the unrolled solver
    (== adjoint_solver=False); "reference_adjoint": torchdiffeq.odeint_adjoint's backward (the reference default, :40-42)."""
    x0 = initialize_state(p, z, prefix)                                            # :37
    if grad_mode == "reference_adjoint" and method != "dopri5":
        zc = z.detach()

        def fb(params):
            q = dict(p)
            for k, v in zip(_DYN_KEYS, params):
                q[prefix + k] = v
            return lambda t, x: dynamics(q, t, x, zc, prefix)
        sol = _OdeintAdjoint.apply(fb, x0, times, method, *[p[prefix + k] for k in _DYN_KEYS])
        return sol.permute(1, 0, 2)
    if method == "dopri5":
        # reference_adjoint with the adaptive solver: odeint_adjoint's adjoint_params are the dynamics weights only (z is a plain
        # tensor, hard part 2), and its continuous adjoint -- itself solved by dopri5 -- equals the gradient of the discrete solve to
        # solver tolerance.  Restated at solution level: autograd through the solve with the latent detached inside the dynamics.
        zd = z.detach() if grad_mode == "reference_adjoint" else z

        def fr(t, x):
            if t.dim() == 0:
                return dynamics(p, t, x, zd, prefix)
            return _dynamics_rowtime(p, t, x, zd, prefix)
        sol = odeint_dopri5(fr, x0, times, **kw)
    else:
        sol = odeint_fixed(lambda t, x: dynamics(p, t, x, z, prefix), x0, times, method)  # :41-45
    return sol.permute(1, 0, 2)                                                    # :47


def _dynamics_rowtime(p, t_col, state, z, prefix):
    """Same arithmetic as :func:`dynamics` with a per-row time column (per-trajectory dopri5 only)."""
    x = torch.cat([t_col, z], dim=1)
    wh, bh = p[prefix + "dynamics.dynamics_hidden.weight"], p[prefix + "dynamics.dynamics_hidden.bias"]
    h = torch.relu(F.linear(x, wh, bh))
    xa = torch.sigmoid(F.linear(h, p[prefix + "dynamics.dyanamics_growth.weight"], p[prefix + "dynamics.dyanamics_growth.bias"]))
    xd = torch.sigmoid(F.linear(h, p[prefix + "dynamics.dyanmics_degradation.weight"], p[prefix + "dynamics.dyanmics_degradation.bias"]))
    return xa - xd * state


# ----------------------------------------------------------------------------------------------------------
# L2 decoders
# ----------------------------------------------------------------------------------------------------------
def decoder_ald(p: Params, z: Tensor, times: Tensor, method: str, **kw):
    """models/decoders.py:42-54 (Decoder.forward) -> (solution[B,T,S], mu_75, mu_50, mu_25, std) each [B,C,T]."""
    sol = solve_ode(p, z, times, method, **kw)                                     # :43
    mu50 = F.linear(sol, p["decoder.output_q50.0.weight"]).permute(0, 2, 1)        # :45
    mu75 = F.linear(sol, p["decoder.output_q75.0.weight"]).permute(0, 2, 1)        # :46
    mu25 = F.linear(sol, p["decoder.output_q25.0.weight"]).permute(0, 2, 1)        # :47
    std = torch.ones_like(mu50) * F.softplus(p["decoder.constant_std"])            # :52-53
    return sol, mu75, mu50, mu25, std


def decoder_gauss(p: Params, z: Tensor, times: Tensor, method: str, **kw):
    """models/decoders.py:84-91 (GaussianDecoder.forward) -> (solution, mean, std)."""
    sol = solve_ode(p, z, times, method, **kw)
    mean = F.linear(sol, p["decoder.output_mean.0.weight"]).permute(0, 2, 1)
    std = torch.ones_like(mean) * F.softplus(p["decoder.constant_std"])
    return sol, mean, std


# ----------------------------------------------------------------------------------------------------------
# L3 likelihoods and log-probs (Pyro distributions are thin wrappers over torch.distributions)
# ----------------------------------------------------------------------------------------------------------
def ald_loglik(obs: Tensor, mu: Tensor, std: Tensor, tau: float) -> Tensor:
    """models/mechanistic_cvs.py:142-158,180-211 (get_series + compute_likelihood), all channels.

    Two Laplace sites per channel: elements with target < pred scaled by (1 - tau) ('_g'), elements with
    target >= pred scaled by tau ('_l').  Restated without masked_select (identical arithmetic per element)."""
    lp = Laplace(mu, std).log_prob(obs)
    ge = obs.ge(mu)
    return (1 - tau) * lp[~ge].sum() + tau * lp[ge].sum()


def ald_loglik_3q(obs, mu75, mu50, mu25, std, quantile_diff: float) -> Tensor:
    """models/mechanistic_cvs.py:160-172 -- tau in (0.5 w/ mu_50, 0.5+diff w/ mu_75, 0.5-diff w/ mu_25)."""
    median = 0.5
    lower, upper = median - quantile_diff, median + quantile_diff
    return ald_loglik(obs, mu50, std, median) + ald_loglik(obs, mu75, std, upper) + ald_loglik(obs, mu25, std, lower)


def gauss_loglik(obs, mean, std) -> Tensor:
    """models/mechanistic_cvs_Gauss.py:163-169 -- sum over channels of Normal(mean_k, std_k).log_prob(x_k)."""
    return Normal(mean, std).log_prob(obs).sum()


def normal_lp(x, loc, scale) -> Tensor:
    return Normal(loc, scale).log_prob(x).sum()


# ----------------------------------------------------------------------------------------------------------
# Model specifications (one per dataset) and -ELBO assembly (restating Pyro Trace_ELBO; module docstring)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class PriorGroup:
    """One p(z_g | u_g) net: z dims [z_off, z_off+z_dim), label columns [u_off, u_off+u_dim) of u."""
    prefix: str
    z_off: int
    z_dim: int
    u_off: int
    u_dim: int


@dataclass
class Spec:
    name: str                      # 'cvs' | 'proc' | 'challenge'
    gauss: bool
    n_channels: int
    latent_dim: int
    z_eps_dim: int
    pool_size: int
    solver: str
    quantile_diff: float
    prior_groups: List[PriorGroup]
    aux_mult: float = 46.0
    # aux heads: (kind, prefix, z_off, z_dim, u_off, u_dim); kind in 'bernoulli'|'onehot'|'laplace'
    aux_heads: List[Tuple[str, str, int, int, int, int]] = field(default_factory=list)
    labels_in_main: bool = False   # proc: main model also scores the labels (mechanistic_proc.py:145-146)
    solver_kw: dict = field(default_factory=dict)
    grad_mode: str = "exact"       # "exact" (adjoint_solver=False) | "reference_adjoint" (torchdiffeq.odeint_adjoint, the default)


def cvs_spec(z_iext=5, z_rtpr=5, z_eps=5, gauss=False, solver="midpoint", quantile_diff=0.475, pool_size=5) -> Spec:
    """data/cvs/config_cvs.py:6-52 + models/mechanistic_cvs.py:18-103.  u = [iext, rtpr] columns."""
    return Spec("cvs", gauss, 3, z_iext + z_rtpr + z_eps, z_eps, pool_size, solver, quantile_diff,
                [PriorGroup("p_z_iext_given_iext.", 0, z_iext, 0, 1),
                 PriorGroup("p_z_rtprs_given_rtprs.", z_iext, z_rtpr, 1, 1)],
                aux_heads=[("bernoulli", "q_iext_given_z_iext.", 0, z_iext, 0, 1),
                           ("bernoulli", "q_rtpr_given_z_rtpr.", z_iext, z_rtpr, 1, 1)])


def challenge_spec(z_shed=5, z_symp=5, z_eps=5, gauss=False, solver="midpoint", quantile_diff=0.475, pool_size=5) -> Spec:
    """data/challenge/config_challenge.py + models/mechanistic_challenge.py.  u = cat(symptoms, shedding)
    (:167).  z layout: [z_shedding, z_symptoms, z_epsilon] (:244-262)."""
    return Spec("challenge", gauss, 4, z_shed + z_symp + z_eps, z_eps, pool_size, solver, quantile_diff,
                [PriorGroup("p_z_u_given_u.", 0, z_shed + z_symp, 0, 2)],
                aux_heads=[("bernoulli", "q_shedding_given_z_shedding.", 0, z_shed, 1, 1),
                           ("bernoulli", "q_symptom_given_z_symptom.", z_shed, z_symp, 0, 1)])


def proc_spec(z_g=10, z_eps=10, gauss=False, solver="midpoint", quantile_diff=0.475, pool_size=5, **solver_kw) -> Spec:
    """data/proc/config_proc.py + models/mechanistic_proc.py.  u = cat(aR[3], aS[4], C12[1], C6[1]) (:196-198).
    z layout: [z_aR, z_aS, z_C12, z_C6, z_epsilon] (:282-311)."""
    return Spec("proc", gauss, 4, 4 * z_g + z_eps, z_eps, pool_size, solver, quantile_diff,
                [PriorGroup("p_z_u_given_u.", 0, 4 * z_g, 0, 9)],
                aux_heads=[("onehot", "q_aR_given_z_aR.", 0, z_g, 0, 3),
                           ("onehot", "q_aS_given_z_aS.", z_g, z_g, 3, 4),
                           ("laplace", "q_C12_given_z_C12.", 2 * z_g, z_g, 7, 1),
                           ("laplace", "q_C6_given_z_C6.", 3 * z_g, z_g, 8, 1)],
                labels_in_main=True, solver_kw=solver_kw)


def _label_terms(p: Params, spec: Spec, z: Tensor, u: Tensor) -> Tensor:
    """q_label / q_continous: mechanistic_cvs.py:261-270, mechanistic_proc.py:334-353,
    mechanistic_challenge.py:282-291.  Returns sum of UNSCALED label log-probs (caller applies 46x)."""
    total = z.new_zeros(())
    for kind, prefix, zo, zd, uo, ud in spec.aux_heads:
        zg, lab = z[:, zo:zo + zd], u[:, uo:uo + ud]
        if kind == "bernoulli":
            total = total + Bernoulli(probs=classifier_sigmoid(p, prefix, zg)).log_prob(lab).sum()
        elif kind == "onehot":
            total = total + OneHotCategorical(probs=classifier_softmax(p, prefix, zg)).log_prob(lab).sum()
        elif kind == "laplace":
            loc, _ = regressor_exp_exp(p, prefix, zg)
            std = F.softplus(p["constant_std_C_12" if "C12" in prefix else "constant_std_C_6"])
            total = total + Laplace(loc, std).log_prob(lab).sum()
        else:
            raise ValueError(kind)
    return total


def prior_loc_scale(p: Params, spec: Spec, u: Tensor) -> Tuple[Tensor, Tensor]:
    """Concatenated prior (loc, scale) over all latent dims; z_epsilon dims are N(0, 1)."""
    B = u.shape[0]
    locs, scales = [], []
    for g in spec.prior_groups:
        l, s = prior_net(p, g.prefix, u[:, g.u_off:g.u_off + g.u_dim])
        locs.append(l)
        scales.append(s)
    locs.append(u.new_zeros(B, spec.z_eps_dim))
    scales.append(u.new_ones(B, spec.z_eps_dim))
    return torch.cat(locs, 1), torch.cat(scales, 1)


def decode_loglik(p: Params, spec: Spec, obs: Tensor, z: Tensor, times: Tensor):
    if spec.gauss:
        sol, mean, std = decoder_gauss(p, z, times, spec.solver, grad_mode=spec.grad_mode, **spec.solver_kw)
        return gauss_loglik(obs, mean, std), (sol, mean, std)
    sol, mu75, mu50, mu25, std = decoder_ald(p, z, times, spec.solver, grad_mode=spec.grad_mode, **spec.solver_kw)
    return ald_loglik_3q(obs, mu75, mu50, mu25, std, spec.quantile_diff), (sol, mu75, mu50, mu25, std)


def main_loss(p: Params, spec: Spec, obs: Tensor, u: Tensor, eps: Tensor, times: Tensor,
              return_parts: bool = False):
    """-ELBO of SVI(model, guide) summed over the batch (training_cvs.py:236,152; Trace_ELBO restated):
    guide (mechanistic_cvs.py:213-238): z = loc + scale*eps, log q = sum logN(z; loc, scale);
    model (:105-178): log p(z) under the conditional priors + likelihood sites (+ 46x label sites for proc)."""
    loc, scale = encoder_conv(p, obs, spec.pool_size)
    z = loc + scale * eps                                    # Normal.rsample with explicit eps
    log_q = normal_lp(z, loc, scale)
    ploc, pscale = prior_loc_scale(p, spec, u)
    log_p = normal_lp(z, ploc, pscale)
    ll, dec = decode_loglik(p, spec, obs, z, times)
    elbo = ll + log_p - log_q
    if spec.labels_in_main:
        elbo = elbo + spec.aux_mult * _label_terms(p, spec, z, u)
    if return_parts:
        return -elbo, dict(loc=loc, scale=scale, z=z, log_q=log_q, log_p=log_p, ll=ll, dec=dec)
    return -elbo


def aux_loss(p: Params, spec: Spec, obs: Tensor, u: Tensor, eps: Tensor) -> Tensor:
    """-ELBO of SVI(model_meta, guide_meta) (mechanistic_cvs.py:240-276): encoder again; the group latents are
    sampled IN THE MODEL (empty guide) so their Normal log-probs enter with +; labels scored at 46x.
    Only the label-group dims of eps are used."""
    loc, scale = encoder_conv(p, obs, spec.pool_size)
    n_u = spec.latent_dim - spec.z_eps_dim
    zg = loc[:, :n_u] + scale[:, :n_u] * eps[:, :n_u]
    z_full = torch.cat([zg, loc[:, n_u:]], 1)                # eps dims unused by the aux heads
    lp = normal_lp(zg, loc[:, :n_u], scale[:, :n_u])
    return -(lp + spec.aux_mult * _label_terms(p, spec, z_full, u))


# ----------------------------------------------------------------------------------------------------------
# Parameter construction with the reference initialisers -- used to build synthetic models
# ----------------------------------------------------------------------------------------------------------
def init_params(spec: Spec, T: int, S: int = 5, H: int = 25, F_: int = 10, K: int = 10, Hc: int = 50,
                U: int = 25, constant_std: float = 1e-2, seed: int = 12, dtype=torch.float32) -> Params:
    """Random parameters with the reference's shapes, key names and initialisers:
    conv/lin orthogonal (encoder_conv.py:32,35); MLP hidden N(0, 1e-3) (encoder_mlp.py:91-92);
    dynamics xavier_uniform gains 1 / 0.5 / 1 (blackbox_ode.py:74-81); constant_std (decoders.py:39);
    everything else torch.nn.Linear / Conv1d defaults."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    st = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        C, L = spec.n_channels, spec.latent_dim
        n_pool = T - (K - 1) - (spec.pool_size - 1)
        p: Params = {}

        def lin(key, n_in, n_out, bias=True, init=None):
            m = nn.Linear(n_in, n_out, bias=bias)
            if init is not None:
                init(m)
            p[key + ".weight"] = m.weight.detach().clone()
            if bias:
                p[key + ".bias"] = m.bias.detach().clone()

        conv = nn.Conv1d(C, F_, K)
        nn.init.orthogonal_(conv.weight)
        p["encoder.conv.weight"], p["encoder.conv.bias"] = conv.weight.detach().clone(), conv.bias.detach().clone()
        lin("encoder.lin", F_ * n_pool, Hc, init=lambda m: nn.init.orthogonal_(m.weight))
        lin("encoder.z_loc", Hc, L)
        lin("encoder.z_scale.0", Hc, L)
        for gq in spec.prior_groups:
            lin(gq.prefix + "sequential_mlp.1.0.0", gq.u_dim, gq.z_dim)
            lin(gq.prefix + "sequential_mlp.1.1.0", gq.u_dim, gq.z_dim)

        def small(m):
            m.weight.data.normal_(0, 0.001)
            m.bias.data.normal_(0, 0.001)
        for kind, prefix, zo, zd, uo, ud in spec.aux_heads:
            lin(prefix + "sequential_mlp.1.module", zd, U, init=small)
            if kind == "laplace":
                lin(prefix + "sequential_mlp.3.0.0", U, ud)
                lin(prefix + "sequential_mlp.3.1.0", U, ud)
            else:
                lin(prefix + "sequential_mlp.3", U, ud)
        if spec.name == "proc":
            p["constant_std_C_12"] = torch.ones(1) * constant_std
            p["constant_std_C_6"] = torch.ones(1) * constant_std
        lin(_ODE + "latent_to_ode_net.0", L, H)
        lin(_ODE + "latent_to_ode_net.2", H, S)
        lin(_ODE + "dynamics.dynamics_hidden", L + 1, H, init=lambda m: nn.init.xavier_uniform_(m.weight))
        lin(_ODE + "dynamics.dyanamics_growth", H, S, init=lambda m: nn.init.xavier_uniform_(m.weight, gain=0.5))
        lin(_ODE + "dynamics.dyanmics_degradation", H, S, init=lambda m: nn.init.xavier_uniform_(m.weight, gain=1))
        if spec.gauss:
            lin("decoder.output_mean.0", S, C, bias=False)
        else:
            for q in ("q50", "q75", "q25"):
                lin("decoder.output_%s.0" % q, S, C, bias=False)
        p["decoder.constant_std"] = torch.ones(C, T) * constant_std
        return {k: v.to(dtype) for k, v in p.items()}
    finally:
        torch.random.set_rng_state(st)


def synthetic_batch(spec: Spec, B: int, T: int, seed: int = 1234, dtype=torch.float32):
    """SURVEY 8(d) synthetic inputs (deterministic from a seed): obs as a contiguous [B,T,C] tensor returned as
    its [B,C,T] permuted view (cvs/challenge; training_cvs.py:25) or contiguous [B,C,T] (proc;
    utils/proc_dataset.py:150), labels u[B,n_u], eps[B,L], times[T]."""
    g = torch.Generator().manual_seed(seed)
    C, L = spec.n_channels, spec.latent_dim
    t = torch.arange(T, dtype=torch.float64)
    if spec.name == "cvs":
        u = torch.bernoulli(torch.full((B, 2), 0.5), generator=g)
        times = torch.arange(0.0, T * 1.0, 1.0)
    elif spec.name == "challenge":
        u = torch.stack([torch.bernoulli(torch.full((B,), 0.54), generator=g),
                         torch.bernoulli(torch.full((B,), 0.31), generator=g)], 1)
        times = torch.arange(0.0, T * 1.0, 1.0)
    else:
        aR = F.one_hot(torch.randint(0, 3, (B,), generator=g), 3).float()
        aS = F.one_hot(torch.randint(0, 4, (B,), generator=g), 4).float()
        c = torch.log1p(torch.rand(B, 2, generator=g) * 25000.0)
        u = torch.cat([aR, aS, c], 1)
        times = (0.1944 * torch.arange(T) + (torch.rand(T, generator=g) - 0.5) * 0.002).float()
        times[0] = 0.0
    base = torch.rand(B, C, 1, generator=g).double() * 0.2
    sgn = 0.3 + 0.5 * torch.rand(B, C, 1, generator=g).double() + 0.2 * u[:, :1].double().unsqueeze(-1)
    tau = torch.tensor([10.0, 20.0, 40.0, 15.0][:C], dtype=torch.float64).reshape(1, C, 1) * (T / 86.0)
    curve = base + sgn * (1 - torch.exp(-t.reshape(1, 1, T) / tau))
    obs = (curve + 0.05 * torch.randn(B, C, T, generator=g).double()).clamp(0, 1).to(dtype)
    if spec.name != "proc":
        obs = obs.permute(0, 2, 1).contiguous().permute(0, 2, 1)     # [B,C,T] view of contiguous [B,T,C]
    eps = torch.randn(B, L, generator=g).to(dtype)
    return obs, u.to(dtype), eps, times.to(dtype)


def loss_and_grads(p: Params, spec: Spec, obs, u, eps, times, which: str = "main"):
    """Autograd through the unrolled solver == the reference with adjoint_solver=False (SURVEY hard part 2)."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss = main_loss(q, spec, obs, u, eps, times) if which == "main" else aux_loss(q, spec, obs, u, eps)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in q.items()}
    return loss.detach(), grads
