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
