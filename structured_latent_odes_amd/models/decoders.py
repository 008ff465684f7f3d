"""models/decoders.py of the reference (Decoder, GaussianDecoder) on the HIP engine.

``forward(z)`` materialises ``solution`` and the head outputs (the ``recon`` API) with two C-ABI calls
(``slode_ode_solve_fwd``, ``slode_decode_heads``).  Training does not go through here: the fused ``slode_elbo_step`` never
materialises the heads."""
from __future__ import annotations

import torch
import torch.nn as nn

from .blackbox_ode import OdeModel


class _HeadsFn(torch.autograd.Function):
    """Heads + softplus std.  Forward = ``slode_decode_heads``, backward = ``slode_decode_heads_bwd`` (the materialising API; the fused
    training path never comes through here).  The backward differentiates at the weights the FORWARD saw: the head weights and
    ``constant_std`` are views of the flat parameter vector, which the fused Adam kernels update in place through raw pointers
    (no autograd version bump), so the forward keeps its own copy of that small segment for the backward."""

    @staticmethod
    def forward(ctx, dec, x, cstd, *heads):
        b = dec.ode_model._binding_or_raise()
        x = x.contiguous()
        mu, std = b.engine.decode_heads(b.flat, x)
        ctx.binding = b
        snap, ctx.snap_lo = b.engine.heads_snapshot(b.flat)
        ctx.save_for_backward(x, snap)
        ctx.mu_shape = tuple(mu.shape)
        return (std, *[mu[i] for i in range(mu.shape[0])])

    @staticmethod
    def backward(ctx, g_std, *g_mu):
        x, snap = ctx.saved_tensors
        b = ctx.binding
        stacked = torch.zeros(ctx.mu_shape, dtype=torch.float32, device=x.device)
        for i, g in enumerate(g_mu):
            if g is not None:
                stacked[i].copy_(g)
        g_x, g_heads, g_c = b.engine.decode_heads_bwd(b.flat, x, stacked, g_std.contiguous() if g_std is not None else None,
                                                      snapshot=(snap, ctx.snap_lo))
        return (None, g_x, g_c if g_std is not None else None, *[g_heads[i] for i in range(g_heads.shape[0])])


class _DecoderBase(nn.Module):
    _head_names = ()

    def __init__(self, config, times, latent_dim, device):
        super().__init__()
        self.ode_model = OdeModel()
        self.times = times
        self.ode_state_dim, self.obs_dim, self.latent_dim = config.ode_state_dim, config.obs_dim, latent_dim
        self.ode_hidden_dim = config.ode_hidden_dim
        self.ode_model.init_with_params(times=times, ode_state_dim=self.ode_state_dim, latent_dim=latent_dim,
                                        ode_hidden_dim=self.ode_hidden_dim, adjoint_solver=config.adjoint_solver,
                                        solver=config.solver, device=device)
        for name in self._head_names:
            setattr(self, name, nn.Sequential(nn.Linear(self.ode_state_dim, self.obs_dim, bias=False)))
        self.constant_std = nn.Parameter(torch.ones(self.obs_dim, len(times)) * config.constant_std, requires_grad=True)

    def _named_for_binding(self):
        named = self.ode_model._named_for_binding()
        for name in self._head_names:
            named["decoder.%s.0.weight" % name] = getattr(self, name)[0].weight
        named["decoder.constant_std"] = self.constant_std
        return named

    def _solve_and_heads(self, z):
        solution = self.ode_model.solve_ODE(z=z)
        outs = _HeadsFn.apply(self, solution, self.constant_std, *[getattr(self, n)[0].weight for n in self._head_names])
        B = z.shape[0]
        std = outs[0].unsqueeze(0).expand(B, -1, -1)      # the reference materialises ones_like(mu) * softplus(std)
        return solution, std, outs[1:]


class Decoder(_DecoderBase):
    """forward(z) -> (solution[B,T,S], mu_75, mu_50, mu_25, std) each [B,C,T]  (decoders.py:42-54)."""
    _head_names = ("output_q50", "output_q75", "output_q25")

    def forward(self, z):
        solution, std, (mu_50, mu_75, mu_25) = self._solve_and_heads(z)
        return solution, mu_75, mu_50, mu_25, std


class GaussianDecoder(_DecoderBase):
    """forward(z) -> (solution, mean, std)  (decoders.py:84-91)."""
    _head_names = ("output_mean",)

    def forward(self, z):
        solution, std, (mean,) = self._solve_and_heads(z)
        return solution, mean, std


class VarianceGaussianDecoder(nn.Module):
    """Present in the reference (decoders.py:94-141) but used by none of its models (dead code) -- not built."""

    def __init__(self, *a, **k):
        raise NotImplementedError("VarianceGaussianDecoder is unused by every reference model; out of scope (DESIGN.md section 7)")
