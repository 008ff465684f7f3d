"""Known-answer tests pinning the parts of the oracle that restate absent third-party code
(torchdiffeq integrators, Pyro ELBO pieces).  SURVEY 8(c) K1-K4.  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import slode_oracle as O


def _lin_f(a, d):
    return lambda t, x: a - d * x


@pytest.mark.parametrize("method,order", [("euler", 1), ("midpoint", 2), ("rk4", 4)])
def test_fixed_grid_order_of_convergence(method, order):
    a, d, x0 = 0.7, 1.3, torch.tensor([[0.2]], dtype=torch.float64)
    exact = a / d + (0.2 - a / d) * math.exp(-d * 1.0)
    errs = []
    for n in (8, 16, 32):
        times = torch.linspace(0, 1, n + 1, dtype=torch.float64)
        sol = O.odeint_fixed(_lin_f(a, d), x0, times, method)
        errs.append(abs(sol[-1].item() - exact))
    slopes = [math.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert all(abs(s - order) < 0.25 for s in slopes), slopes


def test_one_step_algebra():
    a, d, h, x = 0.3, 0.8, 0.5, 0.9
    x0 = torch.tensor([[x]], dtype=torch.float64)
    t = torch.tensor([0.0, h], dtype=torch.float64)
    f = lambda y: a - d * y
    assert O.odeint_fixed(_lin_f(a, d), x0, t, "euler")[1].item() == pytest.approx(x + h * f(x), abs=1e-15)
    assert O.odeint_fixed(_lin_f(a, d), x0, t, "midpoint")[1].item() == pytest.approx(x + h * f(x + h / 2 * f(x)), abs=1e-15)
    k1 = f(x); k2 = f(x + h * k1 / 3); k3 = f(x + h * (k2 - k1 / 3)); k4 = f(x + h * (k1 - k2 + k3))
    assert O.odeint_fixed(_lin_f(a, d), x0, t, "rk4")[1].item() == pytest.approx(x + h * (k1 + 3 * (k2 + k3) + k4) / 8, abs=1e-15)
    # 3/8 rule differs from classic RK4 on a time-dependent problem
    g = lambda t, y: torch.cos(3 * t) - y
    c1 = g(t[0], x0); c2 = g(t[0] + h / 2, x0 + h / 2 * c1); c3 = g(t[0] + h / 2, x0 + h / 2 * c2); c4 = g(t[1], x0 + h * c3)
    classic = x0 + h * (c1 + 2 * c2 + 2 * c3 + c4) / 6
    assert abs(O.odeint_fixed(g, x0, t, "rk4")[1].item() - classic.item()) > 1e-6


@pytest.mark.parametrize("method,R", [("euler", 1), ("midpoint", 2), ("rk4", 3)])
def test_stage_time_table(method, R):
    times = torch.tensor([0.0, 0.19, 0.41, 0.6, 1.0])
    tab = O.stage_times(times, method)
    assert tab.shape[0] == R * 4 + 1 and tab[-1] == times[-1]
    seen = []
    O.odeint_fixed(lambda t, x: (seen.append(float(t)), -x)[1], torch.ones(1, 1), times, method)
    distinct = []
    for s in seen:                      # rk4's k4 time equals the next step's k1 time
        if not distinct or distinct[-1] != s:
            distinct.append(s)
    assert np.array_equal(np.array(distinct, dtype=np.float32), tab.numpy()[:len(distinct)])


@pytest.mark.parametrize("per_traj", [False, True])
def test_dopri5_vs_scipy(per_traj):
    from scipy.integrate import solve_ivp
    torch.manual_seed(3)
    spec = O.cvs_spec(3, 3, 2, solver="dopri5")
    p = {k: v.double() for k, v in O.init_params(spec, T=20).items()}
    z = torch.randn(3, 8, dtype=torch.float64)
    times = torch.linspace(0, 6, 13, dtype=torch.float64)
    sol = O.solve_ode(p, z, times, "dopri5", rtol=1e-9, atol=1e-11, per_trajectory=per_traj)
    x0 = O.initialize_state(p, z)

    def rhs(t, y):
        yy = torch.from_numpy(y).reshape(3, 5)
        return O.dynamics(p, torch.tensor(t, dtype=torch.float64), yy, z).reshape(-1).numpy()
    ref = solve_ivp(rhs, (0, 6), x0.reshape(-1).numpy(), method="RK45", t_eval=times.numpy(), rtol=1e-11, atol=1e-13)
    want = torch.from_numpy(ref.y.T.reshape(13, 3, 5)).permute(1, 0, 2)
    assert (sol - want).abs().max() < 1e-6


def test_ald_vs_bruteforce_and_logprob_formulas():
    torch.manual_seed(0)
    obs, mu = torch.rand(2, 3, 7, dtype=torch.float64), torch.rand(2, 3, 7, dtype=torch.float64)
    std = torch.rand(2, 3, 7, dtype=torch.float64) + 0.1
    for tau in (0.5, 0.975, 0.025):
        want = 0.0
        for b in range(2):
            for k in range(3):
                for t in range(7):
                    x, m, s = obs[b, k, t].item(), mu[b, k, t].item(), std[b, k, t].item()
                    w = tau if x >= m else 1 - tau
                    want += w * (-math.log(2 * s) - abs(x - m) / s)
        assert O.ald_loglik(obs, mu, std, tau).item() == pytest.approx(want, rel=1e-12)
    x, m, s = 0.3, -0.2, 0.7
    assert O.normal_lp(torch.tensor(x), torch.tensor(m), torch.tensor(s)).item() == pytest.approx(
        -math.log(s) - 0.5 * math.log(2 * math.pi) - (x - m) ** 2 / (2 * s * s), rel=1e-6)
    assert O.gauss_loglik(obs, mu, std).item() == pytest.approx(
        (-torch.log(std) - 0.5 * math.log(2 * math.pi) - (obs - mu) ** 2 / (2 * std ** 2)).sum().item(), rel=1e-12)


@pytest.mark.parametrize("name", ["cvs", "cvs_gauss", "challenge", "proc"])
def test_losses_finite_and_differentiable(name):
    spec = {"cvs": O.cvs_spec(1, 1, 2, solver="rk4"), "cvs_gauss": O.cvs_spec(3, 3, 2, gauss=True, solver="midpoint"),
            "challenge": O.challenge_spec(solver="rk4"), "proc": O.proc_spec(z_g=3, z_eps=2, solver="euler")}[name]
    S = 8 if name == "proc" else 5
    p = O.init_params(spec, T=30, S=S)
    obs, u, eps, times = O.synthetic_batch(spec, 4, 30)
    loss, g = O.loss_and_grads(p, spec, obs, u, eps, times, "main")
    assert torch.isfinite(loss) and all(torch.isfinite(v).all() for v in g.values())
    used = [k for k, v in g.items() if v.abs().max() > 0]
    assert "encoder.lin.weight" in used and "decoder.ode_model.dynamics.dynamics_hidden.weight" in used
    loss2, g2 = O.loss_and_grads(p, spec, obs, u, eps, times, "aux")
    assert torch.isfinite(loss2)
    # evaluate_loss == step loss value (Trace_ELBO: same number under no_grad)
    with torch.no_grad():
        assert torch.equal(O.main_loss(p, spec, obs, u, eps, times), loss)
