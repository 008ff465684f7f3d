"""Test infrastructure: a numpy (fp64) emulation of the ALGORITHM the HIP kernels implement -- each fixed-grid
step collapsed to an affine map x' = A x + b (because f(t,x) = a(t,z) - d(t,z) x, blackbox_ode.py:97-109), the
hand-derived exact discrete adjoint, and the closed-form latent/likelihood gradients.  It exists so the
derivations in csrc/*.hip can be checked against oracle autograd on the CPU (tests/test_kernel_math.py).
It is not imported by the product."""
import numpy as np

_O = "decoder.ode_model."
ONE_THIRD = 1.0 / 3.0


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softplus(x):
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def stage_layout(method):
    return {"euler": 1, "midpoint": 2, "rk4": 3}[method]


def stage_times(times, method):
    t0, dt = times[:-1], np.diff(times)
    cols = {"euler": [t0], "midpoint": [t0, t0 + 0.5 * dt], "rk4": [t0, t0 + dt * ONE_THIRD, t0 + dt * 2 * ONE_THIRD]}[method]
    return np.concatenate([np.stack(cols, 1).reshape(-1), times[-1:]])


def step_coeffs(method, h, a, d):
    """a, d: [..., R+1, S] stage values of one step (last = next step's first).  Returns A, b and a cache."""
    if method == "euler":
        return 1 - h * d[..., 0, :], h * a[..., 0, :], None
    if method == "midpoint":
        m, c = 1 - 0.5 * h * d[..., 0, :], 0.5 * h * a[..., 0, :]
        return 1 - h * d[..., 1, :] * m, h * (a[..., 1, :] - d[..., 1, :] * c), (m, c)
    h3 = h * ONE_THIRD
    p1, q1 = a[..., 0, :], -d[..., 0, :]
    c2, m2 = h3 * p1, 1 + h3 * q1
    p2, q2 = a[..., 1, :] - d[..., 1, :] * c2, -d[..., 1, :] * m2
    c3, m3 = h * (p2 - p1 * ONE_THIRD), 1 + h * (q2 - q1 * ONE_THIRD)
    p3, q3 = a[..., 2, :] - d[..., 2, :] * c3, -d[..., 2, :] * m3
    c4, m4 = h * (p1 - p2 + p3), 1 + h * (q1 - q2 + q3)
    p4, q4 = a[..., 3, :] - d[..., 3, :] * c4, -d[..., 3, :] * m4
    G = h * 0.125
    return 1 + G * (q1 + 3 * (q2 + q3) + q4), G * (p1 + 3 * (p2 + p3) + p4), (c2, m2, c3, m3, c4, m4)


def step_coeffs_bwd(method, h, a, d, cache, gA, gb):
    """Returns ga, gd with the shape of a, d (per-step contributions; slot R is the shared next-step slot)."""
    ga, gd = np.zeros_like(a), np.zeros_like(d)
    if method == "euler":
        gd[..., 0, :], ga[..., 0, :] = -h * gA, h * gb
        return ga, gd
    if method == "midpoint":
        m, c = cache
        ga[..., 1, :] = h * gb
        gd[..., 1, :] = -h * (m * gA + c * gb)
        gm, gc = -h * d[..., 1, :] * gA, -h * d[..., 1, :] * gb
        gd[..., 0, :], ga[..., 0, :] = -0.5 * h * gm, 0.5 * h * gc
        return ga, gd
    c2, m2, c3, m3, c4, m4 = cache
    h3, G = h * ONE_THIRD, h * 0.125
    gp1, gq1, gp4, gq4 = G * gb, G * gA, G * gb, G * gA
    gp2, gp3, gq2, gq3 = 3 * G * gb, 3 * G * gb, 3 * G * gA, 3 * G * gA
    ga[..., 3, :] = gp4
    gd[..., 3, :] = -c4 * gp4 - m4 * gq4
    gc4, gm4 = -d[..., 3, :] * gp4, -d[..., 3, :] * gq4
    gp1, gp2, gp3 = gp1 + h * gc4, gp2 - h * gc4, gp3 + h * gc4
    gq1, gq2, gq3 = gq1 + h * gm4, gq2 - h * gm4, gq3 + h * gm4
    ga[..., 2, :] = gp3
    gd[..., 2, :] = -c3 * gp3 - m3 * gq3
    gc3, gm3 = -d[..., 2, :] * gp3, -d[..., 2, :] * gq3
    gp2, gp1 = gp2 + h * gc3, gp1 - h3 * gc3
    gq2, gq1 = gq2 + h * gm3, gq1 - h3 * gm3
    ga[..., 1, :] = gp2
    gd[..., 1, :] = -c2 * gp2 - m2 * gq2
    gc2, gm2 = -d[..., 1, :] * gp2, -d[..., 1, :] * gq2
    gp1, gq1 = gp1 + h3 * gc2, gq1 + h3 * gm2
    ga[..., 0, :], gd[..., 0, :] = gp1, -gq1
    return ga, gd


def ode_forward(p, z, times, method):
    """Returns x[B,T,S] and everything the backward needs."""
    Wh, bh = p[_O + "dynamics.dynamics_hidden.weight"], p[_O + "dynamics.dynamics_hidden.bias"]
    Wg, bg = p[_O + "dynamics.dyanamics_growth.weight"], p[_O + "dynamics.dyanamics_growth.bias"]
    Wd, bd = p[_O + "dynamics.dyanmics_degradation.weight"], p[_O + "dynamics.dyanmics_degradation.bias"]
    W1, b1 = p[_O + "latent_to_ode_net.0.weight"], p[_O + "latent_to_ode_net.0.bias"]
    W2, b2 = p[_O + "latent_to_ode_net.2.weight"], p[_O + "latent_to_ode_net.2.bias"]
    B, T, R = z.shape[0], times.shape[0], stage_layout(method)
    ts = stage_times(times, method)                                   # [R(T-1)+1]
    u = z @ Wh[:, 1:].T + bh                                          # [B,H]   time-invariant part
    pre = Wh[:, 0][None, None, :] * ts[None, :, None] + u[:, None, :]  # [B,nt,H]
    hid = np.maximum(pre, 0)
    a = sigmoid(hid @ Wg.T + bg)                                      # [B,nt,S]
    d = sigmoid(hid @ Wd.T + bd)
    pre0 = z @ W1.T + b1
    hid0 = np.maximum(pre0, 0)
    x0 = sigmoid(hid0 @ W2.T + b2)
    h = np.diff(times)[None, :, None]                                 # [1,T-1,1]
    idx = (np.arange(T - 1)[:, None] * R + np.arange(R + 1)[None, :])  # [T-1,R+1] stage slots of each step
    A, bb, cache = step_coeffs(method, h, a[:, idx, :], d[:, idx, :])
    x = np.empty((B, T, x0.shape[1]))
    x[:, 0] = x0
    for n in range(T - 1):
        x[:, n + 1] = A[:, n] * x[:, n] + bb[:, n]
    return x, dict(ts=ts, u=u, pre=pre, hid=hid, a=a, d=d, pre0=pre0, hid0=hid0, x0=x0, A=A, idx=idx, cache=cache, h=h, R=R)


def ode_backward(p, z, times, method, x, sv, gx):
    """Exact discrete adjoint.  gx[B,T,S] = dLoss/dx.  Returns gz and a dict of parameter grads."""
    Wh = p[_O + "dynamics.dynamics_hidden.weight"]
    Wg, Wd = p[_O + "dynamics.dyanamics_growth.weight"], p[_O + "dynamics.dyanmics_degradation.weight"]
    W1, W2 = p[_O + "latent_to_ode_net.0.weight"], p[_O + "latent_to_ode_net.2.weight"]
    B, T, S = x.shape
    A, idx, R = sv["A"], sv["idx"], sv["R"]
    lam = np.empty_like(x)
    lam[:, T - 1] = gx[:, T - 1]
    for n in range(T - 2, -1, -1):
        lam[:, n] = gx[:, n] + A[:, n] * lam[:, n + 1]
    gb = lam[:, 1:]
    gA = lam[:, 1:] * x[:, :-1]
    ga_s, gd_s = step_coeffs_bwd(method, sv["h"], sv["a"][:, idx, :], sv["d"][:, idx, :], sv["cache"], gA, gb)
    ga, gd = np.zeros_like(sv["a"]), np.zeros_like(sv["d"])
    np.add.at(ga, (slice(None), idx), ga_s)
    np.add.at(gd, (slice(None), idx), gd_s)
    gapre, gdpre = ga * sv["a"] * (1 - sv["a"]), gd * sv["d"] * (1 - sv["d"])
    g = {}
    g[_O + "dynamics.dyanamics_growth.weight"] = np.einsum("bis,bij->sj", gapre, sv["hid"])
    g[_O + "dynamics.dyanamics_growth.bias"] = gapre.sum((0, 1))
    g[_O + "dynamics.dyanmics_degradation.weight"] = np.einsum("bis,bij->sj", gdpre, sv["hid"])
    g[_O + "dynamics.dyanmics_degradation.bias"] = gdpre.sum((0, 1))
    ghid = gapre @ Wg + gdpre @ Wd
    gpre = ghid * (sv["pre"] > 0)
    gwt = np.einsum("bij,i->j", gpre, sv["ts"])
    gu = gpre.sum(1)                                                  # [B,H]
    gWh = np.concatenate([gwt[:, None], gu.T @ z], 1)
    g[_O + "dynamics.dynamics_hidden.weight"], g[_O + "dynamics.dynamics_hidden.bias"] = gWh, gu.sum(0)
    gz = gu @ Wh[:, 1:]
    go = lam[:, 0] * sv["x0"] * (1 - sv["x0"])
    g[_O + "latent_to_ode_net.2.weight"], g[_O + "latent_to_ode_net.2.bias"] = go.T @ sv["hid0"], go.sum(0)
    gpre0 = (go @ W2) * (sv["pre0"] > 0)
    g[_O + "latent_to_ode_net.0.weight"], g[_O + "latent_to_ode_net.0.bias"] = gpre0.T @ z, gpre0.sum(0)
    gz = gz + gpre0 @ W1
    return gz, g


def decode_ll(p, spec, obs, x):
    """Loss contribution -LL and its gradients wrt x, heads and constant_std.  obs [B,C,T]."""
    cstd = p["decoder.constant_std"]
    sig = softplus(cstd)[None]                                       # [1,C,T]
    if spec.gauss:
        heads, taus = ["decoder.output_mean.0.weight"], [None]
    else:
        dq = spec.quantile_diff
        heads, taus = ["decoder.output_q50.0.weight", "decoder.output_q75.0.weight", "decoder.output_q25.0.weight"], [0.5, 0.5 + dq, 0.5 - dq]
    nll, gx, g, gsig = 0.0, np.zeros_like(x), {}, np.zeros_like(obs)
    for key, tau in zip(heads, taus):
        W = p[key]                                                   # [C,S]
        mu = np.einsum("bts,cs->bct", x, W)
        r = obs - mu
        if tau is None:
            nll -= (-np.log(sig) - 0.5 * np.log(2 * np.pi) - r * r / (2 * sig * sig)).sum()
            gmu = -(r / (sig * sig))
            gsig += -(-1 / sig + r * r / sig ** 3)
        else:
            w = np.where(obs >= mu, tau, 1 - tau)
            nll -= (w * (-np.log(2 * sig) - np.abs(r) / sig)).sum()
            gmu = -(w * np.sign(r) / sig)
            gsig += -(w * (-1 / sig + np.abs(r) / (sig * sig)))
        gx += np.einsum("bct,cs->bts", gmu, W)
        g[key] = np.einsum("bct,bts->cs", gmu, x)
    dsp = np.where(cstd > 20.0, 1.0, sigmoid(cstd))
    g["decoder.constant_std"] = gsig.sum(0) * dsp
    return nll, gx, g


def latent_terms(p, spec, loc, scale, eps, u):
    """z, loss contribution (-log p + log q), dLoss/dz (direct), g_scale extra, prior-net grads."""
    z = loc + scale * eps
    B, L = z.shape
    log_q = (-np.log(scale) - 0.5 * np.log(2 * np.pi) - (z - loc) ** 2 / (2 * scale * scale)).sum()
    ploc, pls = np.zeros((B, L)), np.zeros((B, L))
    for gq in spec.prior_groups:
        ug = u[:, gq.u_off:gq.u_off + gq.u_dim]
        ploc[:, gq.z_off:gq.z_off + gq.z_dim] = ug @ p[gq.prefix + "sequential_mlp.1.0.0.weight"].T + p[gq.prefix + "sequential_mlp.1.0.0.bias"]
        pls[:, gq.z_off:gq.z_off + gq.z_dim] = ug @ p[gq.prefix + "sequential_mlp.1.1.0.weight"].T + p[gq.prefix + "sequential_mlp.1.1.0.bias"]
    ps = np.exp(pls)
    dz = (z - ploc) / ps
    log_p = (-pls - 0.5 * np.log(2 * np.pi) - 0.5 * dz * dz).sum()
    gz = dz / ps                                                     # d(-log p)/dz
    g = {}
    g_pl, g_ls = -dz / ps, 1 - dz * dz
    for gq in spec.prior_groups:
        ug = u[:, gq.u_off:gq.u_off + gq.u_dim]
        sl = slice(gq.z_off, gq.z_off + gq.z_dim)
        g[gq.prefix + "sequential_mlp.1.0.0.weight"], g[gq.prefix + "sequential_mlp.1.0.0.bias"] = g_pl[:, sl].T @ ug, g_pl[:, sl].sum(0)
        g[gq.prefix + "sequential_mlp.1.1.0.weight"], g[gq.prefix + "sequential_mlp.1.1.0.bias"] = g_ls[:, sl].T @ ug, g_ls[:, sl].sum(0)
    return z, log_q - log_p, gz, -1.0 / scale, g


def encoder_forward(p, obs, pool):
    W, bconv = p["encoder.conv.weight"], p["encoder.conv.bias"]     # [F,C,K]
    Fn, C, K = W.shape
    B, _, T = obs.shape
    n_conv, n_pool = T - K + 1, T - K + 1 - pool + 1
    win = np.stack([obs[:, :, k:k + n_conv] for k in range(K)], -1)  # [B,C,n_conv,K]
    conv = np.einsum("bcpk,fck->bfp", win, W) + bconv[None, :, None]
    pooled = np.stack([conv[:, :, q:q + n_pool] for q in range(pool)], 0).sum(0) / pool
    flat = pooled.reshape(B, -1)
    hid = np.tanh(flat @ p["encoder.lin.weight"].T + p["encoder.lin.bias"])
    loc = hid @ p["encoder.z_loc.weight"].T + p["encoder.z_loc.bias"]
    scale = np.exp(hid @ p["encoder.z_scale.0.weight"].T + p["encoder.z_scale.0.bias"])
    return loc, scale, dict(win=win, flat=flat, hid=hid, n_conv=n_conv, n_pool=n_pool, pool=pool)


def encoder_backward(p, sv, scale, g_loc, g_scale):
    g = {}
    hid, flat = sv["hid"], sv["flat"]
    g_ls = g_scale * scale
    g["encoder.z_loc.weight"], g["encoder.z_loc.bias"] = g_loc.T @ hid, g_loc.sum(0)
    g["encoder.z_scale.0.weight"], g["encoder.z_scale.0.bias"] = g_ls.T @ hid, g_ls.sum(0)
    g_hid = g_loc @ p["encoder.z_loc.weight"] + g_ls @ p["encoder.z_scale.0.weight"]
    g_pre = g_hid * (1 - hid * hid)
    g["encoder.lin.weight"], g["encoder.lin.bias"] = g_pre.T @ flat, g_pre.sum(0)
    Fn = p["encoder.conv.weight"].shape[0]
    B = hid.shape[0]
    g_pool = (g_pre @ p["encoder.lin.weight"]).reshape(B, Fn, sv["n_pool"])
    g_conv = np.zeros((B, Fn, sv["n_conv"]))
    for q in range(sv["pool"]):
        g_conv[:, :, q:q + sv["n_pool"]] += g_pool / sv["pool"]
    g["encoder.conv.weight"] = np.einsum("bfp,bcpk->fck", g_conv, sv["win"])
    g["encoder.conv.bias"] = g_conv.sum((0, 2))
    return g


def main_step(p, spec, obs, u, eps, times):
    """Whole main-loss step (loss = -ELBO summed over the batch) and all gradients, kernel-style."""
    loc, scale, esv = encoder_forward(p, obs, spec.pool_size)
    z, lat_loss, gz_lat, gscale_q, g = latent_terms(p, spec, loc, scale, eps, u)
    x, sv = ode_forward(p, z, times, spec.solver)
    nll, gx, gdec = decode_ll(p, spec, obs, x)
    gz_ode, gode = ode_backward(p, z, times, spec.solver, x, sv, gx)
    gz = gz_lat + gz_ode
    g_loc, g_scale = gz, gz * eps + gscale_q
    g.update(gdec)
    g.update(gode)
    g.update(encoder_backward(p, esv, scale, g_loc, g_scale))
    return nll + lat_loss, g, dict(x=x, loc=loc, scale=scale, z=z, g_loc=g_loc, g_scale=g_scale)


# ---- adaptive Dormand-Prince: per-trajectory controller with step records, and the reverse sweep over the records -----------
# (csrc/dopri5_kernel.hip: dopri5_kernel / dopri5_bwd_kernel; one trajectory at a time here)
_DP_A = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
_DP_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0]                      # six distinct stage times (stages 6 and 7 share t + dt)
_DP_B = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
_DP_E = [35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
         11 / 84 - 649 / 6300, -1 / 60]
_DP_M = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _dp_unpack(p):
    Wh, bh = p[_O + "dynamics.dynamics_hidden.weight"], p[_O + "dynamics.dynamics_hidden.bias"]
    Wg, bg = p[_O + "dynamics.dyanamics_growth.weight"], p[_O + "dynamics.dyanamics_growth.bias"]
    Wd, bd = p[_O + "dynamics.dyanmics_degradation.weight"], p[_O + "dynamics.dyanmics_degradation.bias"]
    return Wh, bh, Wg, bg, Wd, bd


def _dp_coef(t, wt, u, Wg, bg, Wd, bd):
    pre = wt * t + u
    h = np.maximum(pre, 0)
    return sigmoid(Wg @ h + bg), sigmoid(Wd @ h + bd), pre, h


def _dp_stages(t, dt, y, wt, u, Wg, bg, Wd, bd):
    """The seven stage slopes, the stage states and the coefficients at the six stage times of one step."""
    A, D, PRE, HID = zip(*[_dp_coef(t + c * dt if c != 1.0 else t + dt, wt, u, Wg, bg, Wd, bd) for c in _DP_C])
    ks, ys = [A[0] - D[0] * y], [y]
    for i, row in enumerate(_DP_A):
        yi = y + dt * sum(c * k for c, k in zip(row, ks))
        ys.append(yi)
        ks.append(A[i + 1] - D[i + 1] * yi)
    y1 = y + dt * sum(c * k for c, k in zip(_DP_B, ks))
    ks.append(A[5] - D[5] * y1)                                          # k7 = f(t + dt, y1)
    return A, D, PRE, HID, ks, ys, y1


def dopri5_forward(p, z, times, rtol, atol, max_steps=100000):
    """x[B,T,S] and, per trajectory, the accepted steps [(t, dt, y)] -- torchdiffeq's algorithm with one controller per trajectory."""
    Wh, bh, Wg, bg, Wd, bd = _dp_unpack(p)
    W1, b1 = p[_O + "latent_to_ode_net.0.weight"], p[_O + "latent_to_ode_net.0.bias"]
    W2, b2 = p[_O + "latent_to_ode_net.2.weight"], p[_O + "latent_to_ode_net.2.bias"]
    B, T, S = z.shape[0], times.shape[0], Wg.shape[0]
    x, recs = np.empty((B, T, S)), []
    rms = lambda v: np.sqrt(np.mean(v * v))
    for b in range(B):
        wt, u = Wh[:, 0], Wh[:, 1:] @ z[b] + bh
        y = sigmoid(W2 @ np.maximum(W1 @ z[b] + b1, 0) + b2)
        x[b, 0] = y
        t = times[0]
        f = (lambda tt, yy: (lambda a, d, *_: a - d * yy)(*_dp_coef(tt, wt, u, Wg, bg, Wd, bd)))
        f0 = f(t, y)
        sc = atol + np.abs(y) * rtol
        d0, d1 = rms(y / sc), rms(f0 / sc)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        d2 = rms((f(t + h0, y + h0 * f0) - f0) / sc) / h0
        h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
        dt = min(100 * h0, h1)
        j, rec = 1, []
        for _ in range(max_steps):
            if j >= T:
                break
            A, D, PRE, HID, ks, ys, y1 = _dp_stages(t, dt, y, wt, u, Wg, bg, Wd, bd)
            err = dt * sum(c * k for c, k in zip(_DP_E, ks))
            ratio = rms(err / (atol + rtol * np.maximum(np.abs(y), np.abs(y1))))
            if ratio <= 1:
                rec.append((t, dt, y.copy()))
                t1 = t + dt
                ymid = y + dt * sum(c * k for c, k in zip(_DP_M, ks))
                f0_, f1_ = ks[0], ks[6]
                ca = 2 * dt * (f1_ - f0_) - 8 * (y1 + y) + 16 * ymid
                cb = dt * (5 * f0_ - 3 * f1_) + 18 * y + 14 * y1 - 32 * ymid
                cc = dt * (f1_ - 4 * f0_) - 11 * y - 5 * y1 + 16 * ymid
                cd = dt * f0_
                while j < T and times[j] <= t1:
                    q = (times[j] - t) / dt
                    x[b, j] = y + q * (cd + q * (cc + q * (cb + q * ca)))
                    j += 1
                t, y = t1, y1
            factor = 10.0 if ratio == 0 else min(10.0, max(0.9 * ratio ** -0.2, 1.0 if ratio < 1 else 0.2))
            dt *= factor
        assert j >= T, "max_steps exceeded"
        recs.append(rec)
    return x, recs


def dopri5_backward(p, z, times, recs, gx, drop_z=False):
    """Reverse mode of the accepted steps and of the dense output (step sizes fixed): the formulas of dopri5_bwd_kernel.
    gx[B,T,S] = dLoss/dx.  Returns gz[B,L] and the gradients of the ode parameters."""
    Wh, bh, Wg, bg, Wd, bd = _dp_unpack(p)
    W1, b1 = p[_O + "latent_to_ode_net.0.weight"], p[_O + "latent_to_ode_net.0.bias"]
    W2, b2 = p[_O + "latent_to_ode_net.2.weight"], p[_O + "latent_to_ode_net.2.bias"]
    B, T, S = gx.shape
    H = Wh.shape[0]
    GWg, GWd, Gbg, Gbd = np.zeros_like(Wg), np.zeros_like(Wd), np.zeros(S), np.zeros(S)
    GWh, Gbh = np.zeros_like(Wh), np.zeros(H)
    GW1, Gb1, GW2, Gb2 = np.zeros_like(W1), np.zeros(H), np.zeros_like(W2), np.zeros(S)
    gz = np.zeros_like(z)
    for b in range(B):
        wt, u = Wh[:, 0], Wh[:, 1:] @ z[b] + bh
        lam, j, gu = np.zeros(S), T - 1, np.zeros(H)
        for (t, dt, y) in reversed(recs[b]):
            A, D, PRE, HID, ks, ys, y1 = _dp_stages(t, dt, y, wt, u, Wg, bg, Wd, bd)
            Ga = Gb = Gc = Gd = np.zeros(S)
            gy = np.zeros(S)
            while j >= 1 and times[j] > t:                                     # dense outputs inside (t, t + dt]
                q = (times[j] - t) / dt
                g = gx[b, j]
                gy = gy + g
                Gd, Gc, Gb, Ga = Gd + q * g, Gc + q * q * g, Gb + q ** 3 * g, Ga + q ** 4 * g
                j -= 1
            gf0 = dt * (-2 * Ga + 5 * Gb - 4 * Gc + Gd)
            gf1 = dt * (2 * Ga - 3 * Gb + Gc)
            gm = 16 * Ga - 32 * Gb + 16 * Gc
            gy = gy - 8 * Ga + 18 * Gb - 11 * Gc + gm
            gy1 = lam - 8 * Ga + 14 * Gb - 5 * Gc
            gk = [dt * gm * m for m in _DP_M]                                   # dL/dk_1..7 from y_mid
            gk[0] = gk[0] + gf0
            gk[6] = gk[6] + gf1
            sig = [None] * 6                                                    # (dL/d pre-sigmoid growth, degradation) per stage time
            # stage 7: k7 = a5 - d5 * y1
            sa, sd = gk[6] * A[5] * (1 - A[5]), -gk[6] * y1 * D[5] * (1 - D[5])
            gy1 = gy1 - D[5] * gk[6]
            gy = gy + gy1
            for i in range(6):
                gk[i] = gk[i] + dt * _DP_B[i] * gy1
            # stages 6 .. 2: k_i = a - d * y_i, y_i = y + dt * sum_j a_ij k_j
            for st in range(5, 0, -1):
                ga_, gd_ = gk[st] * A[st] * (1 - A[st]), -gk[st] * ys[st] * D[st] * (1 - D[st])
                if st == 5:
                    ga_, gd_ = ga_ + sa, gd_ + sd
                sig[st] = (ga_, gd_)
                ev = -D[st] * gk[st]
                gy = gy + ev
                for jj, c in enumerate(_DP_A[st - 1]):
                    gk[jj] = gk[jj] + dt * c * ev
            sig[0] = (gk[0] * A[0] * (1 - A[0]), -gk[0] * y * D[0] * (1 - D[0]))
            gy = gy - D[0] * gk[0]
            lam = gy
            for e in range(6):
                te = t + _DP_C[e] * dt if _DP_C[e] != 1.0 else t + dt
                ga_, gd_ = sig[e]
                GWg += np.outer(ga_, HID[e]); GWd += np.outer(gd_, HID[e]); Gbg += ga_; Gbd += gd_
                gh = (ga_ @ Wg + gd_ @ Wd) * (PRE[e] > 0)
                gu += gh
                GWh[:, 0] += gh * te
        pre0 = W1 @ z[b] + b1
        hp = np.maximum(pre0, 0)
        x0 = sigmoid(W2 @ hp + b2)
        go = (lam + gx[b, 0]) * x0 * (1 - x0)
        GW2 += np.outer(go, hp); Gb2 += go
        gp = (W2.T @ go) * (pre0 > 0)
        GW1 += np.outer(gp, z[b]); Gb1 += gp
        GWh[:, 1:] += np.outer(gu, z[b]); Gbh += gu
        gz[b] = W1.T @ gp + (0 if drop_z else Wh[:, 1:].T @ gu)
    g = {_O + "dynamics.dynamics_hidden.weight": GWh, _O + "dynamics.dynamics_hidden.bias": Gbh,
         _O + "dynamics.dyanamics_growth.weight": GWg, _O + "dynamics.dyanamics_growth.bias": Gbg,
         _O + "dynamics.dyanmics_degradation.weight": GWd, _O + "dynamics.dyanmics_degradation.bias": Gbd,
         _O + "latent_to_ode_net.0.weight": GW1, _O + "latent_to_ode_net.0.bias": Gb1,
         _O + "latent_to_ode_net.2.weight": GW2, _O + "latent_to_ode_net.2.bias": Gb2}
    return gz, g


# ---- round-2 algorithms of csrc/ode_kernel.hip (ALG 0) in numpy fp64 ---------------------------------------------------------------
def switching_indices(wt, u, ts):
    """Per hidden unit: (ms, sf).  relu(wt t + u) is on exactly where wt*t + u > 0; along a monotone table that predicate flips at
    most once:  sf = 1: on for m >= ms (ms = 0 always, ms = nt never);  sf = 0: on for m < ms  (P0b of the kernel)."""
    nt = ts.shape[0]
    ms, sf = np.zeros(wt.shape[0], int), np.ones(wt.shape[0], int)
    for j in range(wt.shape[0]):
        pred = wt[j] * ts + u[j] > 0
        assert (np.diff(pred.astype(int)) != 0).sum() <= 1, "predicate must flip at most once along a monotone table"
        if pred[0] == pred[-1]:
            ms[j] = 0 if pred[0] else nt
        else:
            ms[j], sf[j] = int(np.argmax(pred != pred[0])), int(pred[-1])
    return ms, sf


def pwl_heads(wt, u, W, bias, ts):
    """Head pre-activations o[m, c] = bias_c + sum_j W[c, j] relu(wt_j ts[m] + u_j) from the piecewise-linear table of P0c / P1:
    row k = [value at the k-th segment's first stage time | slope], events in (switching index, unit) order."""
    H, nt = wt.shape[0], ts.shape[0]
    ms, sf = switching_indices(wt, u, ts)
    order = sorted(range(H), key=lambda j: (ms[j], j))
    tau = [ts[0]] + [ts[min(ms[j], nt - 1)] for j in order]
    on0 = sf == 0
    al = (W[:, on0] * wt[on0]).sum(1)
    V = bias + (W[:, on0] * (wt[on0] * ts[0] + u[on0])).sum(1)
    rows = [(V.copy(), al.copy())]
    for k, j in enumerate(order):
        sw = (1.0 if sf[j] else -1.0) * W[:, j]
        V = V + al * (tau[k + 1] - tau[k]) + sw * (wt[j] * tau[k + 1] + u[j])
        al = al + sw * wt[j]
        rows.append((V.copy(), al.copy()))
    ps = np.array(sorted(ms))
    out = np.empty((nt, W.shape[0]))
    for m in range(nt):
        k = int(np.searchsorted(ps, m, side="right"))
        out[m] = rows[k][0] + rows[k][1] * (ts[m] - tau[k])
    return out


def contraction_by_switching_sums(wt, u, W, g, ts, n_chunks=25):
    """P6: from the per-sample head gradients g[m, c] to (dW[c, j], dbias[c], dLoss/du_j, dLoss/dwt_j) through chunk sums and each
    unit's switching index -- no per-sample x per-unit work."""
    H, nt, NC = wt.shape[0], ts.shape[0], g.shape[1]
    ms, sf = switching_indices(wt, u, ts)
    CL = (nt + n_chunks - 1) // n_chunks
    cg = np.zeros((n_chunks, NC)); cgt = np.zeros((n_chunks, NC))
    for q in range(n_chunks):
        sl = slice(q * CL, min(nt, (q + 1) * CL))
        cg[q], cgt[q] = g[sl].sum(0), (g[sl] * ts[sl, None]).sum(0)
    GM, GT = np.zeros((NC, H)), np.zeros((NC, H))
    for j in range(H):
        qs = ms[j] // CL
        whole = range(qs + 1, n_chunks) if sf[j] else range(0, min(qs, n_chunks))
        cut = range(ms[j], min(nt, (qs + 1) * CL)) if sf[j] else range(qs * CL, ms[j])
        GM[:, j] = sum(cg[q] for q in whole) + sum(g[m] for m in cut)
        GT[:, j] = sum(cgt[q] for q in whole) + sum(g[m] * ts[m] for m in cut)
    dW = wt[None, :] * GT + u[None, :] * GM
    return dW, cg.sum(0), (W * GM).sum(0), (W * GT).sum(0)


# ---- round-2 algorithms of csrc/dopri5_kernel.hip in numpy fp64 --------------------------------------------------------------------
def segment_table(wt, u, W, bias, tlo, thi, big=3.0e38):
    """load_units: switching times th_j = -u_j / wt_j (wt_j == 0: the sign of u_j decides, always / never on), dir bit (on from th_j
    upwards), their order by counting, and the H + 1 rows [value at the segment centre | slope] built event by event in the centred
    form.  Returns (th, dirs, centres, V[H+1, NC], AL[H+1, NC])."""
    H = wt.shape[0]
    th = np.where(wt != 0, -u / np.where(wt != 0, wt, 1.0), np.where(u > 0, -big, big))
    th = np.clip(th, -big, big)
    dirs = wt >= 0
    rank = np.array([sum((th[k] < th[j]) or (th[k] == th[j] and k < j) for k in range(H)) for j in range(H)])
    order = np.argsort(rank)
    c = tlo
    on0 = ~dirs                                     # below every switching time the units with wt < 0 are on
    V = bias + (W[:, on0] * (wt[on0] * c + u[on0])).sum(1)
    AL = (W[:, on0] * wt[on0]).sum(1)
    rows_V, rows_AL, centres = [V.copy()], [AL.copy()], [c]
    for j in order:
        c1 = min(max(th[j], tlo), thi)
        sg = 1.0 if dirs[j] else -1.0
        V = V + AL * (c1 - c) + sg * W[:, j] * (wt[j] * c1 + u[j])
        AL = AL + sg * W[:, j] * wt[j]
        c = c1
        rows_V.append(V.copy()); rows_AL.append(AL.copy()); centres.append(c)
    return th, dirs, np.array(centres), np.array(rows_V), np.array(rows_AL)


def eval_segment(t, th, centres, V, AL):
    """eval_ad: the segment of t is the NUMBER of switching times <= t (a popcount of the t >= th_j bits)."""
    r = int((t >= th).sum())
    return V[r] + AL[r] * (t - centres[r])


def sweep_by_switching_times(wt, u, W, th, dirs, ts_desc, g):
    """grp::sweep_sample + the end phase of dopri5_bwd_kernel: samples visited in DEcreasing time, running sums RS = sum g, RT = sum g t,
    parked when a unit's `t >= th_j` bit flips (at most once per unit); then GM_j = snapshot | total - snapshot | total | 0.
    ts_desc[m], g[m, c]: the samples in the order of the sweep.  Returns (dW[c, j], dbias[c], dLoss/du_j, dLoss/dwt_j)."""
    H, NC = wt.shape[0], g.shape[1]
    RS, RT = np.zeros(NC), np.zeros(NC)
    snapS, snapT, flipped = np.zeros((H, NC)), np.zeros((H, NC)), np.zeros(H, bool)
    ge_prev = None
    for m in range(ts_desc.shape[0]):
        ge = ts_desc[m] >= th
        if ge_prev is not None:
            for j in np.nonzero(ge != ge_prev)[0]:
                assert not flipped[j], "a unit's bit flips at most once along a monotone sweep"
                snapS[j], snapT[j], flipped[j] = RS, RT, True
        ge_prev = ge
        RS = RS + g[m]
        RT = RT + g[m] * ts_desc[m]
    on_early = ~(ge_prev ^ dirs)                    # on at the earliest (last visited) sample
    GM, GT = np.zeros((NC, H)), np.zeros((NC, H))
    for j in range(H):
        sm, st = (snapS[j], snapT[j]) if flipped[j] else (0.0, 0.0)
        GM[:, j] = RS - sm if on_early[j] else sm
        GT[:, j] = RT - st if on_early[j] else st
    dW = wt[None, :] * GT + u[None, :] * GM
    return dW, RS, (W * GM).sum(0), (W * GT).sum(0)
