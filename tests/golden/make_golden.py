"""Generate golden input/output vectors from the REFERENCE's own importable modules.

Run in the build container only (``/root/reference`` never travels to the GPU box):

    python tests/golden/make_golden.py

Writes ``tests/golden/*.npz`` (data only: weights, inputs, outputs).  No reference source is copied.
``torchdiffeq`` and ``pyro`` are absent here; the reference modules below import them at module top but never
dereference them on the code paths exercised, so inert empty placeholder modules are registered for the import
(SURVEY 8c).  The integrator (``solve_ODE``) and the Pyro ELBO assembly can NOT be run => they are not in
these fixtures (parity unpinned there; see oracle/slode_oracle.py).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.path.insert(0, REF)
    sys.modules.setdefault("torchdiffeq", types.ModuleType("torchdiffeq"))
    pyro = types.ModuleType("pyro")
    pd = types.ModuleType("pyro.distributions")
    pu = types.ModuleType("pyro.distributions.util")
    pu.broadcast_shape = None        # only dereferenced when allow_broadcast=True (never set by the reference)
    sys.modules.setdefault("pyro", pyro)
    sys.modules.setdefault("pyro.distributions", pd)
    sys.modules.setdefault("pyro.distributions.util", pu)
    from models.blackbox_ode import OdeModel                     # noqa
    from models.encoder_conv import EncoderCONV                  # noqa
    from models.encoder_mlp import EncoderMLP                    # noqa
    from models.decoders import Decoder, GaussianDecoder         # noqa
    from utils.exp import Exp                                    # noqa
    return OdeModel, EncoderCONV, EncoderMLP, Decoder, GaussianDecoder, Exp


def sd(m, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def main():
    OdeModel, EncoderCONV, EncoderMLP, Decoder, GaussianDecoder, Exp = _import_reference()
    torch.manual_seed(0)

    # G1: EncoderCONV, shapes (B,C,T) incl. a permuted-view input -------------------------------------------
    g1 = {}
    for i, (B, C, T, L) in enumerate([(4, 3, 86, 15), (4, 3, 200, 8), (3, 4, 100, 50), (2, 4, 300, 15), (5, 3, 100, 4)]):
        torch.manual_seed(i)
        enc = EncoderCONV(n_channels=C, n_filters=10, filter_size=10, pool_size=5, n_time=T, latent_dim=L, hidden_dim=50)
        x_btc = torch.rand(B, T, C)
        x = x_btc.permute(0, 2, 1) if i % 2 == 0 else x_btc.permute(0, 2, 1).contiguous()
        with torch.no_grad():
            loc, scale = enc(x)
        g1.update({"c%d.x" % i: x.contiguous().numpy(), "c%d.loc" % i: loc.numpy(), "c%d.scale" % i: scale.numpy(),
                   "c%d.meta" % i: np.array([B, C, T, L])})
        g1.update(sd(enc, "c%d.p." % i))
    np.savez_compressed(os.path.join(OUT, "g1_encoder_conv.npz"), **g1)

    # G2: EncoderMLP constructor patterns used by the three models ------------------------------------------
    g2 = {}
    pats = [("sig", [5, 25, 1], nn.Softplus, nn.Sigmoid),
            ("softmax", [10, 25, 3], nn.Softplus, nn.Softmax),
            ("expexp", [10, 25, [1, 1]], nn.Softplus, [Exp, Exp]),
            ("prior1", [1, [3, 3]], nn.Softplus, [None, Exp]),
            ("prior9", [9, [40, 40]], nn.Softplus, [None, Exp])]
    for name, sizes, act, oact in pats:
        torch.manual_seed(7)
        m = EncoderMLP(mlp_sizes=sizes, activation=act, output_activation=oact, allow_broadcast=False, use_cuda=True)
        # the reference initialises hidden layers to N(0, 1e-3); perturb so the test is sensitive
        with torch.no_grad():
            for prm in m.parameters():
                prm.add_(torch.randn_like(prm) * 0.3)
        x = torch.randn(6, sizes[0])
        with torch.no_grad():
            y = m(x)
        g2[name + ".x"] = x.numpy()
        if isinstance(y, (list, tuple)):
            g2[name + ".y0"], g2[name + ".y1"] = y[0].numpy(), y[1].numpy()
        else:
            g2[name + ".y0"] = y.numpy()
        g2.update(sd(m, name + ".p."))
    np.savez_compressed(os.path.join(OUT, "g2_encoder_mlp.npz"), **g2)

    # G3: Dynamics / OdeFunc / initialize_state -----------------------------------------------------------
    g3 = {}
    for i, (L, S) in enumerate([(4, 5), (8, 5), (15, 5), (50, 8)]):
        torch.manual_seed(10 + i)
        om = OdeModel()
        om.init_with_params(times=torch.arange(5.0), ode_state_dim=S, latent_dim=L, ode_hidden_dim=25,
                            adjoint_solver=False, solver="rk4", device="cpu")
        z = torch.randn(6, L)
        with torch.no_grad():
            x0 = om.initialize_state(z)
            f = om.gen_dynamics(z)
            st = torch.rand(6, S)
            for j, t in enumerate([0.0, 0.5, 1.0 / 3.0, 85.0]):
                g3["d%d.f%d" % (i, j)] = f(torch.tensor(t), st).numpy()
                g3["d%d.t%d" % (i, j)] = np.float32(t)
        g3.update({"d%d.z" % i: z.numpy(), "d%d.x0" % i: x0.numpy(), "d%d.state" % i: st.numpy()})
        g3.update(sd(om, "d%d.p.decoder.ode_model." % i))
    np.savez_compressed(os.path.join(OUT, "g3_dynamics.npz"), **g3)

    # G4: decoder heads + softplus std on a given trajectory tensor (forward dies at solve_ODE) -------------
    g4 = {}
    cfg = types.SimpleNamespace(ode_state_dim=5, obs_dim=3, ode_hidden_dim=25, system_input_dim=2,
                                adjoint_solver=False, solver="rk4", constant_std=1e-2)
    torch.manual_seed(21)
    times = torch.arange(0.0, 40.0)
    dec = Decoder(config=cfg, times=times, latent_dim=8, device="cpu")
    with torch.no_grad():
        dec.constant_std.add_(torch.randn_like(dec.constant_std))
        sol = torch.rand(4, 40, 5)
        g4["ald.sol"] = sol.numpy()
        g4["ald.mu50"] = torch.squeeze(dec.output_q50(sol)).permute(0, 2, 1).numpy()
        g4["ald.mu75"] = torch.squeeze(dec.output_q75(sol)).permute(0, 2, 1).numpy()
        g4["ald.mu25"] = torch.squeeze(dec.output_q25(sol)).permute(0, 2, 1).numpy()
        g4["ald.std"] = (torch.ones(4, 3, 40) * torch.nn.Softplus()(dec.constant_std)).numpy()
    g4.update(sd(dec, "ald.p.decoder."))
    gd = GaussianDecoder(config=cfg, times=times, latent_dim=8, device="cpu")
    with torch.no_grad():
        g4["gauss.sol"] = sol.numpy()
        g4["gauss.mean"] = torch.squeeze(gd.output_mean(sol)).permute(0, 2, 1).numpy()
        g4["gauss.std"] = (torch.ones(4, 3, 40) * torch.nn.Softplus()(gd.constant_std)).numpy()
    g4.update(sd(gd, "gauss.p.decoder."))
    np.savez_compressed(os.path.join(OUT, "g4_decoders.npz"), **g4)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
