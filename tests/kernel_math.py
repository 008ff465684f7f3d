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
