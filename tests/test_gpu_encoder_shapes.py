"""Encoder shapes other than the reference's (n_filters = 10, filter_size = 10, pool_size = 5) through the folded ELBO step.

The chain-rule launch runs one workgroup per (hidden unit, PAIR of conv filters) and the fold / chain kernels are instantiated for a
14-tap box-filtered kernel (every reference config) and for the general tap bound; the reference's configs reach neither an odd filter
count (a last pair with ONE filter), nor a hidden width that is not 50, nor the long-tap instantiation.  Each case here is a full
ELBO step (loss + every gradient tensor) against the fp64 oracle, the workspace poisoned with NaN beforehand, run twice (bitwise equal).
Tolerances as everywhere: -ELBO 1e-5 relative, gradients 5e-4 norm-wise per tensor."""
import dataclasses

import pytest
import torch

from oracle import slode_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    # name: (family, T, n_filters, filter_size, pool_size, cnn_hidden_dim)
    "cvs_F7_K6_P3_Hc40": ("cvs", 64, 7, 6, 3, 40),        # odd filter count: the last filter pair holds one filter; J = 8
    "cvs_F4_K12_P6_Hc50": ("cvs", 90, 4, 12, 6, 50),      # J = 17 > 14: the general-tap instantiations of fold and chain kernels
    "challenge_F5_K10_P5_Hc33": ("challenge", 72, 5, 10, 5, 33),   # four channels, odd filter count, odd hidden width
    "cvs_F1_K10_P5_Hc50": ("cvs", 50, 1, 10, 5, 50),      # a single filter
}
B = 10


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", list(CASES))
def test_folded_step_with_other_encoder_shapes(name):
    from structured_latent_odes_amd import engine as E
    fam, T, F_, K, P, Hc = CASES[name]
    ospec = {"cvs": O.cvs_spec, "challenge": O.challenge_spec}[fam](solver="rk4", pool_size=P)
    p = O.init_params(ospec, T=T, S=5, F_=F_, K=K, Hc=Hc)
    g = torch.Generator().manual_seed(23)
    p = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    obs, u, eps, times = O.synthetic_batch(ospec, B, T)
    want_loss, want = O.loss_and_grads({k: v.double() for k, v in p.items()}, ospec, obs.double(), u.double(), eps.double(), times.double())
    dev = torch.device("cuda:0")
    espec = {"cvs": E.cvs_spec, "challenge": E.challenge_spec}[fam](solver="rk4")
    espec = dataclasses.replace(espec, n_filters=F_, filter_size=K, pool_size=P, cnn_hidden_dim=Hc)
    eng = E.Engine(espec, T, dev)
    eng.set_times(times)
    flat = eng.pack(p)
    obs_d = obs.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)      # native [B,T,C] layout: the folded encoder path
    u_d, eps_d = u.to(dev).contiguous(), eps.to(dev).contiguous()
    outs = []
    for rep in range(2):
        eng.workspace(B).fill_(float("nan"))
        loss = torch.full((1,), float("nan"), device=dev)
        grads = torch.full((eng.n_params,), float("nan"), device=dev)
        eng.elbo_step(flat, obs_d, u_d, eps_d, loss, grads=grads)
        outs.append((loss.clone(), grads.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    loss, grads = outs[0]
    assert torch.isfinite(loss).all() and torch.isfinite(grads).all()
    assert abs(loss.item() - want_loss.item()) / abs(want_loss.item()) < 1e-5
    bad = {k: _rel(v, want[k]) for k, v in eng.unpack(grads).items() if _rel(v, want[k]) > 5e-4}
    assert not bad, bad
    # the same step with Adam fused into the last launch: the weights move as torch.optim.Adam moves them on these gradients
    flat2, m_, v_ = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    loss2, grads2 = torch.zeros(1, device=dev), torch.zeros(eng.n_params, device=dev)
    eng.elbo_adam_step(flat2, obs_d, u_d, eps_d, loss2, grads2, m_, v_, 1e-3, 1)
    ref = torch.nn.Parameter(flat.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    ref.grad = grads2[:flat.numel()].clone()
    opt.step()
    assert torch.equal(grads2, grads)
    assert _rel(flat2, ref.detach()) < 1e-6
