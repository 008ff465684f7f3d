"""models/blackbox_ode.py of the reference (OdeModel, OdeFunc, Dynamics) on the HIP engine.

``OdeModel.solve_ODE`` = ``slode_ode_solve_fwd`` (+ ``slode_ode_solve_bwd`` under autograd): no torchdiffeq.
``adjoint_solver=False``: exact gradients of the discrete scheme (autograd through ``torchdiffeq.odeint``);
``adjoint_solver=True`` (the reference default, :40-42): the gradients ``torchdiffeq.odeint_adjoint`` returns -- continuous
adjoint stepped backwards with the same fixed-grid method, no gradient to z through the dynamics (SURVEY hard part 2, row N2)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import ModelSpec

_O = "decoder.ode_model."
_KEYS = ["latent_to_ode_net.0.weight", "latent_to_ode_net.0.bias", "latent_to_ode_net.2.weight", "latent_to_ode_net.2.bias",
         "dynamics.dynamics_hidden.weight", "dynamics.dynamics_hidden.bias", "dynamics.dyanamics_growth.weight",
         "dynamics.dyanamics_growth.bias", "dynamics.dyanmics_degradation.weight", "dynamics.dyanmics_degradation.bias"]


class Dynamics(nn.Module):
    """Parameter container with the reference's names and initialisers (blackbox_ode.py:67-95): shared hidden layer
    ``dynamics_hidden`` ((1+L) -> H, time is column 0), heads ``dyanamics_growth`` / ``dyanmics_degradation`` (sic), and the
    ``prod`` / ``degr`` Sequentials aliasing them (so ``state_dict`` carries the same duplicate keys)."""

    def __init__(self, n_inputs, hidden_dim, n_outputs, hidden_activation=nn.Tanh):
        super().__init__()
        self.n_inputs, self.n_outputs = n_inputs, n_outputs
        self.dynamics_hidden = nn.Linear(n_inputs + 1, hidden_dim)
        nn.init.xavier_uniform_(self.dynamics_hidden.weight)
        act = hidden_activation()
        self.dyanamics_growth = nn.Linear(hidden_dim, n_outputs)
        nn.init.xavier_uniform_(self.dyanamics_growth.weight, gain=0.5)
        self.dyanmics_degradation = nn.Linear(hidden_dim, n_outputs)
        nn.init.xavier_uniform_(self.dyanmics_degradation.weight, gain=1)
        self.prod = nn.Sequential(self.dynamics_hidden, act, self.dyanamics_growth, nn.Sigmoid())
        self.degr = nn.Sequential(self.dynamics_hidden, act, self.dyanmics_degradation, nn.Sigmoid())
        self._owner = None

    def forward(self, t, state, constants, n_batch):
        """dx/dt = a(t,z) - d(t,z) * state (blackbox_ode.py:97-109) -- one ``slode_dynamics_eval`` call."""
        b = self._owner._binding_or_raise()
        return b.engine.dynamics_eval(b.flat, float(t), state.contiguous(), constants.contiguous())


class OdeFunc(nn.Module):
    def __init__(self, z, dynamics):
        super().__init__()
        self.dynamics = dynamics
        self.n_batch = z.shape[0]
        self.constants = torch.cat([z], dim=1)

    def forward(self, t, state):
        return self.dynamics.forward(t=t, state=state, constants=self.constants, n_batch=self.n_batch)


class _SolveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, z, *params):
        b = model._binding_or_raise()
        z = z.contiguous()
        x = b.engine.ode_solve(b.flat, z)
        ctx.model = model
        ctx.save_for_backward(z)
        return x

    @staticmethod
    def backward(ctx, g_x):
        b = ctx.model._binding_or_raise()
        (z,) = ctx.saved_tensors
        grads = torch.zeros(b.engine.n_params, dtype=torch.float32, device=b.engine.device)
        g_z = b.engine.ode_solve_bwd(b.flat, z, g_x.contiguous(), grads)
        out = [grads[b.slices[_O + k]].view(p.shape) for k, p in zip(_KEYS, ctx.model._param_list())]
        return (None, g_z, *out)


class OdeModel(nn.Module):
    """Constructor takes no arguments; ``init_with_params`` builds the nets (blackbox_ode.py:7-27)."""

    def __init__(self):
        super().__init__()

    def init_with_params(self, times, ode_state_dim, latent_dim, ode_hidden_dim, adjoint_solver, solver, device):
        self.times = times
        self.ode_state_dim, self.latent_dim, self.ode_hidden_dim = ode_state_dim, latent_dim, ode_hidden_dim
        self.device, self.adjoint_solver, self.solver = device, adjoint_solver, solver
        self.latent_to_ode_net = nn.Sequential(nn.Linear(latent_dim, ode_hidden_dim), nn.ReLU(),
                                               nn.Linear(ode_hidden_dim, ode_state_dim), nn.Sigmoid())
        self.dynamics = Dynamics(n_inputs=latent_dim, hidden_dim=ode_hidden_dim, n_outputs=ode_state_dim, hidden_activation=nn.ReLU)
        object.__setattr__(self.dynamics, "_owner", self)   # plain reference: registering it as a submodule would create a cycle
        self._binding = None

    def _param_list(self):
        n, d = self.latent_to_ode_net, self.dynamics
        return [n[0].weight, n[0].bias, n[2].weight, n[2].bias, d.dynamics_hidden.weight, d.dynamics_hidden.bias,
                d.dyanamics_growth.weight, d.dyanamics_growth.bias, d.dyanmics_degradation.weight, d.dyanmics_degradation.bias]

    def _named_for_binding(self):
        return {_O + k: p for k, p in zip(_KEYS, self._param_list())}

    def _binding_or_raise(self):
        if self._binding is None:   # standalone use: private engine, the other layout segments stay zero
            from ._binding import Binding
            from .encoder_conv import _zero_fill_named
            dev = self.latent_to_ode_net[0].weight.device
            T = int(self.times.numel())
            spec = ModelSpec("ode_only", True, 3, self.latent_dim, self.latent_dim, 0, [], ode_state_dim=self.ode_state_dim,
                             ode_hidden_dim=self.ode_hidden_dim, solver=self.solver,
                             grad_mode="reference_adjoint" if (self.adjoint_solver and self.solver != "dopri5") else "exact")
            if T < 14:
                spec.filter_size, spec.pool_size = 1, 1   # the (unused) encoder segment must still be a valid shape
            named = self._named_for_binding()
            named.update(_zero_fill_named(spec, T, dev, skip=named))
            self._binding = Binding(spec, self.times.to(torch.float32), dev, named)
        return self._binding

    def gen_dynamics(self, z):
        return OdeFunc(z=z, dynamics=self.dynamics)

    def initialize_state(self, z):
        """x0 = sigmoid(W2 relu(W1 z + b1) + b2) (blackbox_ode.py:32-34) = the first grid point of the solve."""
        b = self._binding_or_raise()
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self._param_list())):
            return self.solve_ODE(z)[:, 0, :]           # differentiable route: the first grid point of the (autograd-wrapped) solve
        return b.engine.initialize_state(b.flat, z.to(torch.float32).contiguous())

    def solve_ODE(self, z):
        """[B, L] -> [B, T, S] (blackbox_ode.py:36-47)."""
        return _SolveFn.apply(self, z, *self._param_list())
