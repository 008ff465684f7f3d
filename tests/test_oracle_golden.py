"""Pin the oracle (oracle/slode_oracle.py) against vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import slode_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _params(npz, prefix):
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


@pytest.mark.parametrize("case", range(5))
def test_encoder_conv_matches_reference(golden_dir, case):
    g = _load(golden_dir, "g1_encoder_conv.npz")
    p = {"encoder." + k: v for k, v in _params(g, "c%d.p." % case).items()}
    x = torch.from_numpy(g["c%d.x" % case])
    loc, scale = O.encoder_conv(p, x, pool_size=5)
    # the same torch ops => bit-exact
    assert torch.equal(loc, torch.from_numpy(g["c%d.loc" % case]))
    assert torch.equal(scale, torch.from_numpy(g["c%d.scale" % case]))
    # a permuted view of the [B,T,C] layout gives the same values (training_cvs.py:25)
    xv = x.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    loc2, _ = O.encoder_conv(p, xv, pool_size=5)
    torch.testing.assert_close(loc2, loc, rtol=1e-6, atol=1e-6)


def test_encoder_mlp_patterns_match_reference(golden_dir):
    g = _load(golden_dir, "g2_encoder_mlp.npz")
    for name, fn in [("sig", O.classifier_sigmoid), ("softmax", O.classifier_softmax)]:
        p = {"m." + k: v for k, v in _params(g, name + ".p.").items()}
        y = fn(p, "m.", torch.from_numpy(g[name + ".x"]))
        assert torch.equal(y, torch.from_numpy(g[name + ".y0"]))
    p = {"m." + k: v for k, v in _params(g, "expexp.p.").items()}
    a, b = O.regressor_exp_exp(p, "m.", torch.from_numpy(g["expexp.x"]))
    assert torch.equal(a, torch.from_numpy(g["expexp.y0"])) and torch.equal(b, torch.from_numpy(g["expexp.y1"]))
    for name in ("prior1", "prior9"):
        p = {"m." + k: v for k, v in _params(g, name + ".p.").items()}
        loc, scale = O.prior_net(p, "m.", torch.from_numpy(g[name + ".x"]))
        assert torch.equal(loc, torch.from_numpy(g[name + ".y0"]))
        assert torch.equal(scale, torch.from_numpy(g[name + ".y1"]))


@pytest.mark.parametrize("case", range(4))
def test_dynamics_and_init_state_match_reference(golden_dir, case):
    g = _load(golden_dir, "g3_dynamics.npz")
    p = _params(g, "d%d.p." % case)
    z = torch.from_numpy(g["d%d.z" % case])
    st = torch.from_numpy(g["d%d.state" % case])
    assert torch.equal(O.initialize_state(p, z), torch.from_numpy(g["d%d.x0" % case]))
    for j in range(4):
        t = torch.tensor(float(g["d%d.t%d" % (case, j)]))
        f = O.dynamics(p, t, st, z)
        assert torch.equal(f, torch.from_numpy(g["d%d.f%d" % (case, j)]))
    # aliasing of the shared hidden layer in the reference state_dict (checkpoint row)
    assert np.array_equal(g["d%d.p.decoder.ode_model.dynamics.prod.0.weight" % case],
                          g["d%d.p.decoder.ode_model.dynamics.dynamics_hidden.weight" % case])


def test_decoder_heads_and_std_match_reference(golden_dir):
    g = _load(golden_dir, "g4_decoders.npz")
    p = _params(g, "ald.p.")
    sol = torch.from_numpy(g["ald.sol"])
    import torch.nn.functional as F
    for q in ("50", "75", "25"):
        mu = F.linear(sol, p["decoder.output_q%s.0.weight" % q]).permute(0, 2, 1)
        assert torch.equal(mu, torch.from_numpy(g["ald.mu" + q]))
    std = torch.ones(4, 3, 40) * F.softplus(p["decoder.constant_std"])
    assert torch.equal(std, torch.from_numpy(g["ald.std"]))
    p = _params(g, "gauss.p.")
    mean = F.linear(sol, p["decoder.output_mean.0.weight"]).permute(0, 2, 1)
    assert torch.equal(mean, torch.from_numpy(g["gauss.mean"]))
