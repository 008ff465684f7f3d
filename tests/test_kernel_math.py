"""The algorithm the HIP kernels implement (tests/kernel_math.py: affine-collapsed steps + hand-derived discrete
adjoint) against oracle autograd, fp64, CPU only.  SURVEY 8(c) K5."""
import numpy as np
import pytest
import torch

from oracle import slode_oracle as O
from tests import kernel_math as KM


def _case(name, method):
    spec = {"cvs": O.cvs_spec(3, 3, 2, solver=method), "cvs_gauss": O.cvs_spec(1, 1, 2, gauss=True, solver=method),
            "challenge": O.challenge_spec(solver=method)}[name]
    T, B = 24, 5
    p = {k: v.double() for k, v in O.init_params(spec, T=T).items()}
    # move away from the near-zero initialisation so every gradient path is exercised
    g = torch.Generator().manual_seed(5)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g, dtype=torch.float64) for k, v in p.items()}
    obs, u, eps, times = [t.double() for t in O.synthetic_batch(spec, B, T)]
    if method == "midpoint":
        times = times * 0.37 + 0.01 * torch.rand(T, generator=g).double().cumsum(0)   # non-uniform grid
    return spec, p, obs, u, eps, times


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("name", ["cvs", "cvs_gauss", "challenge"])
def test_kernel_algorithm_matches_oracle_autograd(name, method):
    spec, p, obs, u, eps, times = _case(name, method)
    loss, grads = O.loss_and_grads(p, spec, obs, u, eps, times, "main")
    pn = {k: v.numpy() for k, v in p.items()}
    kl, kg, aux = KM.main_step(pn, spec, obs.numpy(), u.numpy(), eps.numpy(), times.numpy())
    assert kl == pytest.approx(loss.item(), rel=1e-11)
    with torch.no_grad():
        _, parts = O.main_loss(p, spec, obs, u, eps, times, return_parts=True)
    np.testing.assert_allclose(aux["x"], parts["dec"][0].numpy(), rtol=0, atol=1e-12)
    main_keys = [k for k, v in grads.items() if v.abs().max() > 0]
    assert set(main_keys) <= set(kg.keys())
    for k in main_keys:
        np.testing.assert_allclose(kg[k], grads[k].numpy(), rtol=1e-8, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("mode", ["exact", "reference_adjoint"])
def test_dopri5_reverse_sweep_matches_oracle_autograd(mode):
    """The adaptive path of csrc/dopri5_kernel.hip -- per-trajectory controller with recorded steps, then the hand-derived reverse
    mode of the Dormand-Prince stages and of the quartic dense output (step sizes fixed) -- against autograd through the oracle's
    per-trajectory dopri5, fp64, same tolerances => same accepted steps.  `reference_adjoint`: no z -> dynamics path."""
    spec = O.cvs_spec(3, 3, 2, solver="dopri5")
    T, B, S = 16, 3, 5
    p = {k: v.double() for k, v in O.init_params(spec, T=T).items()}
    g = torch.Generator().manual_seed(7)
    p = {k: v + 0.1 * torch.randn(v.shape, generator=g, dtype=torch.float64) for k, v in p.items()}
    times = torch.arange(T, dtype=torch.float64) * 0.6
    z = torch.randn(B, spec.latent_dim, generator=g, dtype=torch.float64)
    gx = torch.randn(B, T, S, generator=g, dtype=torch.float64)
    kw = dict(rtol=1e-6, atol=1e-8, per_trajectory=True)
    q = {k: v.clone().requires_grad_("ode_model" in k) for k, v in p.items()}
    zz = z.clone().requires_grad_(True)
    x = O.solve_ode(q, zz, times, "dopri5", grad_mode=mode, **kw)
    (x * gx).sum().backward()
    pn = {k: v.numpy() for k, v in p.items()}
    xk, recs = KM.dopri5_forward(pn, z.numpy(), times.numpy(), 1e-6, 1e-8)
    # same algorithm => the same accepted steps, up to the conditioning of the error estimate: it is a difference of O(1) slopes
    # worth ~1e-6 of them, so last-bit differences in f (torch cat + linear vs the split hidden layer here) move dt by ~1e-10
    np.testing.assert_allclose(xk, x.detach().numpy(), rtol=0, atol=1e-7)
    assert all(len(r) >= T - 1 for r in recs)
    gz, kg = KM.dopri5_backward(pn, z.numpy(), times.numpy(), recs, gx.numpy(), drop_z=(mode == "reference_adjoint"))
    # gradients inherit that 1e-10 jitter of the step sizes times the sensitivity of the kink quadrature (observed <= 2e-5 relative)
    np.testing.assert_allclose(gz, zz.grad.numpy(), rtol=2e-4, atol=1e-5)
    for k, v in kg.items():
        np.testing.assert_allclose(v, q[k].grad.numpy(), rtol=2e-4, atol=1e-5, err_msg=k)


def test_piecewise_linear_heads_and_switching_sum_contraction():
    """The round-2 kernel algorithms (csrc/ode_kernel.hip, ALG 0) against the direct formulas, fp64: the piecewise-linear table of
    the dynamics heads reproduces bias + W relu(wt t + u) at every stage time, and the chunk-sum / switching-index contraction
    reproduces the per-sample x per-unit weight-gradient sums, for increasing, decreasing and irregular (jittered) time grids and
    for units that are always on, never on, or flip exactly at a grid point."""
    rng = np.random.default_rng(5)
    H, S = 25, 5
    for case in range(6):
        T = [40, 200, 86, 33, 100, 64][case]
        times = np.cumsum(0.05 + rng.random(T)) if case % 2 else np.arange(T, dtype=np.float64)
        if case == 3:
            times = times[::-1].copy()                                   # decreasing grid (torchdiffeq accepts it)
        ts = KM.stage_times(times, "rk4")
        wt = rng.normal(size=H) * 0.4
        u = -wt * rng.choice(ts, size=H) + rng.normal(size=H) * 0.3          # thresholds inside the grid
        wt[0], u[0] = 0.0, 1.0                                                # always on
        wt[1], u[1] = 0.0, -1.0                                               # never on
        wt[2], u[2] = 0.5, -0.5 * ts[7]                                       # pre == 0 exactly at a table entry
        W, bias = rng.normal(size=(2 * S, H)) * 0.3, rng.normal(size=2 * S)
        pre = wt[None, :] * ts[:, None] + u[None, :]
        direct = np.maximum(pre, 0) @ W.T + bias
        assert np.abs(KM.pwl_heads(wt, u, W, bias, ts) - direct).max() < 1e-11 * max(1.0, np.abs(direct).max())
        g = rng.normal(size=(ts.shape[0], 2 * S))
        mask = (pre > 0).astype(np.float64)
        dW_want = g.T @ (np.maximum(pre, 0))                                  # [2S, H]
        gh = (g @ W) * mask                                                   # [nt, H]
        # (case 1: 598 samples in 26 chunks of 23 -- a chunk count that divides the sample count exactly, the shape of the round-2 fuzz find)
        dW, dbias, gu, gwt = KM.contraction_by_switching_sums(wt, u, W, g, ts, n_chunks=[7, 26, 12, 5, 32, 9][case])
        scale = np.abs(dW_want).max()
        assert np.abs(dW - dW_want).max() < 1e-10 * scale
        assert np.abs(dbias - g.sum(0)).max() < 1e-10 * scale
        assert np.abs(gu - gh.sum(0)).max() < 1e-10 * scale and np.abs(gwt - (gh * ts[:, None]).sum(0)).max() < 1e-9 * scale


def test_dopri5_segment_table_and_switching_time_sweep():
    """The round-2 algorithms of csrc/dopri5_kernel.hip against the direct formulas, fp64: the per-trajectory segment table
    (switching times, rank by counting, centred rows, segment = popcount of `t >= th_j`) reproduces bias + W relu(wt t + u) at arbitrary
    evaluation times -- also outside the integration range and for always-on / never-on units -- and the sweep over the samples in
    decreasing time (running sums parked at each unit's switching time) reproduces the per-sample x per-unit weight-gradient sums.
    Away from a switching time the table is exact; AT one the kernel's predicate is `t >= th_j` with the rounded th_j where the
    reference's relu switches at `wt t + u > 0`: the heads are continuous there, so the two differ by O(ulp) of the pre-activation."""
    rng = np.random.default_rng(11)
    H, S = 25, 8
    for case in range(4):
        tlo, thi = (0.0, 99.0) if case < 3 else (2.0, 40.0)
        wt = rng.normal(size=H) * 0.4
        u = -wt * rng.uniform(tlo - 10, thi + 10, size=H)                    # switching times inside and outside the range
        wt[0], u[0] = 0.0, 1.0                                                # always on
        wt[1], u[1] = 0.0, -1.0                                               # never on
        wt[2], u[2] = wt[3], u[3]                                             # two units switching at the same time
        W, bias = rng.normal(size=(2 * S, H)) * 0.3, rng.normal(size=2 * S)
        th, dirs, centres, V, AL = KM.segment_table(wt, u, W, bias, tlo, thi)
        assert np.all(np.diff(centres) >= 0) and centres[0] == tlo and centres[-1] <= thi
        # evaluation times of an adaptive solve: arbitrary, not on any grid; a few outside the range (Hairer's probe step, dense output)
        ts = np.concatenate([rng.uniform(tlo, thi, size=400), [tlo - 3.0, thi + 5.0, tlo, thi]])
        for t in ts:
            direct = np.maximum(wt * t + u, 0) @ W.T + bias
            got = KM.eval_segment(t, th, centres, V, AL)
            assert np.abs(got - direct).max() < 1e-11 * max(1.0, np.abs(direct).max()), (case, t)
        # the reverse sweep visits its samples in decreasing time: 6 stage times per accepted step, steps from last to first
        edges = np.sort(rng.uniform(tlo, thi, size=45))
        edges[0], edges[-1] = tlo, thi
        samples = []
        for k in range(len(edges) - 2, -1, -1):
            t0, dt = edges[k], edges[k + 1] - edges[k]
            samples += [t0 + dt, t0 + dt * 8 / 9, t0 + dt * 4 / 5, t0 + dt * 3 / 10, t0 + dt / 5, t0]
        tsd = np.array(samples)
        g = rng.normal(size=(tsd.shape[0], 2 * S))
        pre = wt[None, :] * tsd[:, None] + u[None, :]
        mask = (tsd[:, None] >= th[None, :]) == dirs[None, :]                 # the kernel's on/off bits
        assert np.all(mask == (pre > 0)) or np.abs(pre[mask != (pre > 0)]).max() < 1e-12   # they are relu's, up to rounding at a switch
        dW_want = g.T @ (pre * mask)
        gh = (g @ W) * mask
        dW, dbias, gu, gwt = KM.sweep_by_switching_times(wt, u, W, th, dirs, tsd, g)
        scale = np.abs(dW_want).max()
        assert np.abs(dW - dW_want).max() < 1e-10 * scale
        assert np.abs(dbias - g.sum(0)).max() < 1e-10 * scale
        assert np.abs(gu - gh.sum(0)).max() < 1e-10 * scale and np.abs(gwt - (gh * tsd[:, None]).sum(0)).max() < 1e-9 * scale
