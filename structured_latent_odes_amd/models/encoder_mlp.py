"""models/encoder_mlp.py of the reference: the generic MLP builder used for the conditional priors p(z_u | u) (no hidden
layer, heads [identity, Exp]) and for the auxiliary classifiers / regressors (one Softplus hidden layer).

Inside the fused ELBO step the prior nets are evaluated by ``ode_elbo_kernel`` (phase P0) straight from the flat parameter
vector; this module owns their parameters (same ``state_dict`` keys, incl. the ``.module.`` level the reference gets from
wrapping hidden ``Linear`` layers in ``nn.DataParallel`` -- encoder_mlp.py:95-96, SURVEY note A) and provides the standalone
``forward`` used by the eval-side API (``recon(is_post=False)``, ``classifier`` / ``pred_inputs``)."""
from inspect import isclass

import torch
import torch.nn as nn


def call_nn_op(op):
    """Instantiate an activation class; Softmax-like ops act over dim 1 (encoder_mlp.py:7-18)."""
    return op(dim=1) if op in (nn.Softmax, nn.LogSoftmax) else op()


class ListOutModule(nn.ModuleList):
    """Applies every member to the same input and returns the list of results (encoder_mlp.py:21-31)."""

    def forward(self, *args, **kwargs):
        return [m(*args, **kwargs) for m in self]


class ConcatModule(nn.Module):
    """Concatenates a tuple/list of tensors along the last dim; a lone tensor passes through (encoder_mlp.py:34-57)."""

    def __init__(self, allow_broadcast=False):
        super().__init__()
        self.allow_broadcast = allow_broadcast

    def forward(self, *input_args):
        if len(input_args) == 1:
            input_args = input_args[0]
        if torch.is_tensor(input_args):
            return input_args
        if self.allow_broadcast:
            shape = torch.broadcast_shapes(*[s.shape[:-1] for s in input_args]) + (-1,)
            input_args = [s.expand(shape) for s in input_args]
        return torch.cat(input_args, dim=-1)


class _Wrapped(nn.Module):
    """Keeps the reference's ``<idx>.module.{weight,bias}`` key level for hidden layers (DataParallel is a pass-through
    on one device; only the name level matters)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return self.module(x)


class EncoderMLP(nn.Module):
    def __init__(self, mlp_sizes, activation=nn.ReLU, output_activation=None,
                 post_layer_fct=lambda layer_ix, total_layers, layer: None,
                 post_act_fct=lambda layer_ix, total_layers, layer: None, allow_broadcast=False, use_cuda=False):
        super().__init__()
        if len(mlp_sizes) < 2:
            raise AssertionError("Must have input and output layer sizes defined")
        in_size, hidden, out_size = mlp_sizes[0], mlp_sizes[1:-1], mlp_sizes[-1]
        width = in_size if isinstance(in_size, int) else sum(in_size)
        mods = [ConcatModule(allow_broadcast)]
        for ix, h in enumerate(hidden):
            if not isinstance(h, int):
                raise AssertionError("Hidden layer sizes must be ints")
            layer = nn.Linear(width, h)
            layer.weight.data.normal_(0, 0.001)      # encoder_mlp.py:91-92
            layer.bias.data.normal_(0, 0.001)
            mods.append(_Wrapped(layer) if use_cuda else layer)
            extra = post_layer_fct(ix + 1, len(hidden), mods[-1])
            if extra is not None:
                mods.append(extra)
            mods.append(activation())
            extra = post_act_fct(ix + 1, len(hidden), mods[-1])
            if extra is not None:
                mods.append(extra)
            width = h
        if isinstance(out_size, int):
            mods.append(nn.Linear(width, out_size))
            if output_activation is not None:
                mods.append(call_nn_op(output_activation) if isclass(output_activation) else output_activation)
        else:
            heads = []
            for oi, osz in enumerate(out_size):
                head = [nn.Linear(width, osz)]
                act = output_activation[oi] if isinstance(output_activation, (list, tuple)) else output_activation
                if act:
                    head.append(call_nn_op(act) if isclass(act) else act)
                heads.append(nn.Sequential(*head))
            mods.append(ListOutModule(heads))
        self.sequential_mlp = nn.Sequential(*mods)

    def forward(self, *args, **kwargs):
        return self.sequential_mlp.forward(*args, **kwargs)
