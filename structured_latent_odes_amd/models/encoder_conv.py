"""models/encoder_conv.py of the reference (EncoderCONV, Exp) on the HIP engine.

forward = ``slode_encoder_conv_fwd``; autograd backward = ``slode_encoder_conv_bwd`` (encoder_conv.py:43-51)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import ModelSpec
from ..utils.exp import Exp  # noqa: F401  (re-exported like the reference module does)

_KEYS = ["conv.weight", "conv.bias", "lin.weight", "lin.bias", "z_loc.weight", "z_loc.bias", "z_scale.0.weight", "z_scale.0.bias"]


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, obs, *params):
        b = module._binding_or_raise()
        loc, scale, pooled, hid = b.engine.encoder_fwd(b.flat, obs)
        ctx.module, ctx.obs = module, obs
        ctx.save_for_backward(scale, pooled, hid)
        return loc, scale

    @staticmethod
    def backward(ctx, g_loc, g_scale):
        b = ctx.module._binding_or_raise()
        scale, pooled, hid = ctx.saved_tensors
        grads = torch.zeros(b.engine.n_params, dtype=torch.float32, device=b.engine.device)
        b.engine.encoder_bwd(b.flat, ctx.obs, scale, pooled, hid, g_loc.contiguous(), g_scale.contiguous(), grads)
        out = [grads[b.slices[ctx.module._prefix + k]].view(p.shape) for k, p in zip(_KEYS, ctx.module._param_list())]
        return (None, None, *out)


class EncoderCONV(nn.Module):
    """Same constructor and ``forward(x[B,C,T]) -> (z_loc, z_scale)`` as the reference (encoder_conv.py:17-51); same
    initialisers (orthogonal conv and lin, :32,35) and ``state_dict`` keys (conv, lin, z_loc, z_scale.0)."""

    def __init__(self, n_channels, n_filters, filter_size, pool_size, n_time, latent_dim, hidden_dim):
        super().__init__()
        self.hidden_dim, self.latent_dim = hidden_dim, latent_dim
        self.n_channels, self.n_filters, self.filter_size, self.pool_size, self.n_time = n_channels, n_filters, filter_size, pool_size, n_time
        n_conv = n_time - (filter_size - 1)
        n_pool = n_conv - (pool_size - 1)
        self.n_hidden_layer = n_pool * n_filters
        self.conv = nn.Conv1d(n_channels, n_filters, filter_size)
        nn.init.orthogonal_(self.conv.weight)
        self.pool = nn.AvgPool1d(pool_size, stride=1)   # parameter-free; kept for repr/state parity, computed in the kernel
        self.lin = nn.Linear(self.n_hidden_layer, hidden_dim)
        nn.init.orthogonal_(self.lin.weight)
        self.act = nn.Tanh()
        self.z_loc = nn.Linear(hidden_dim, latent_dim)
        self.z_scale = nn.Sequential(nn.Linear(hidden_dim, latent_dim), Exp())
        self._binding, self._prefix = None, "encoder."

    def _param_list(self):
        return [self.conv.weight, self.conv.bias, self.lin.weight, self.lin.bias, self.z_loc.weight, self.z_loc.bias,
                self.z_scale[0].weight, self.z_scale[0].bias]

    def _named_for_binding(self, prefix="encoder."):
        return {prefix + k: p for k, p in zip(_KEYS, self._param_list())}

    def _binding_or_raise(self):
        if self._binding is None:   # standalone use: private engine whose non-encoder segments stay zero
            from ._binding import Binding
            from .blackbox_ode import OdeModel
            dev = self.conv.weight.device
            spec = ModelSpec("encoder_only", False, self.n_channels, self.latent_dim, self.latent_dim, 0, [],
                             n_filters=self.n_filters, filter_size=self.filter_size, pool_size=self.pool_size,
                             cnn_hidden_dim=self.hidden_dim, solver="euler")
            times = torch.arange(self.n_time, dtype=torch.float32)
            named = self._named_for_binding()
            named.update(_zero_fill_named(spec, self.n_time, dev, skip=named))
            self._binding = Binding(spec, times, dev, named)
        return self._binding

    def forward(self, x):
        return _EncoderFn.apply(self, x, *self._param_list())


def _zero_fill_named(spec, T, dev, skip):
    """Placeholder parameters for layout segments a standalone module does not own."""
    from ..engine import Engine
    eng = Engine(spec, T, dev)
    out = {}
    for key, off, shp in eng.param_table():
        if key not in skip:
            out[key] = nn.Parameter(torch.zeros(*shp, device=dev), requires_grad=False)
    return out
