"""Drop-in for the ``pyro.infer.SVI`` objects the reference training scripts build (training_cvs.py:236-249):
``ELBOStep.step(**batch) -> float`` and ``.evaluate_loss(**batch) -> float`` return -ELBO summed over the batch.

One step = one ``slode_elbo_step`` call (encoder -> latent sample -> ODE solve -> likelihood -> exact gradient, all HIP),
one RCCL SUM all-reduce of the flat gradient with the loss scalar appended (data parallel), one
``slode_adam_step`` call.  No Pyro, no torchdiffeq, no autograd on this path."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .engine import Engine


class FlatAdam:
    """torch.optim.Adam semantics (what pyro.optim.Adam applies per parameter, training_cvs.py:226-227) on the flat
    parameter vector; state lives on device; one HIP kernel per step."""

    def __init__(self, engine: Engine, params: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.engine, self.params, self.lr, self.betas, self.eps = engine, params, lr, betas, eps
        self.exp_avg = torch.zeros_like(params)
        self.exp_avg_sq = torch.zeros_like(params)
        self.t = 0

    def step(self, grads: torch.Tensor):
        self.t += 1
        self.engine.adam_step(self.params, grads, self.exp_avg, self.exp_avg_sq, self.lr, self.t, self.betas, self.eps)


class ELBOStep:
    """Main-loss SVI object.  ``eps_fn(B, L, device)`` supplies the reparameterisation noise (default: torch.randn on
    device, drawn in the guide's site order); pass a fixed tensor through ``eps=`` for reproducible parity runs."""

    def __init__(self, engine: Engine, params: torch.Tensor, optimizer: Optional[FlatAdam] = None,
                 label_fn: Optional[Callable[..., torch.Tensor]] = None, process_group=None):
        self.engine, self.params, self.optimizer = engine, params, optimizer
        self.label_fn = label_fn
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        # gradient buffer with the loss scalar appended => a single collective per step
        self.gbuf = torch.zeros(params.numel() + 1, dtype=torch.float32, device=params.device)
        self.grads = self.gbuf[:params.numel()]
        self.loss = self.gbuf[params.numel():]

    def _inputs(self, observations, eps, labels):
        eng = self.engine
        u = self.label_fn(**labels) if self.label_fn is not None else labels.get("u")
        if eps is None:
            eps = torch.randn(observations.shape[0], eng.spec.latent_dim, dtype=torch.float32, device=eng.device)
        return observations, u, eps

    def step_async(self, observations, eps=None, **labels):
        """Enqueue one optimisation step on the current stream; returns the device tensor holding -ELBO (global sum)."""
        obs, u, eps = self._inputs(observations, eps, labels)
        if self.params.numel() != self.engine.n_params:
            self.gbuf[self.engine.n_params:self.params.numel()].zero_()   # appended (aux) parameters get no main-loss gradient
        self.engine.elbo_step(self.params, obs, u, eps, self.loss, self.grads)
        if self.world > 1:
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        if self.optimizer is not None:
            self.optimizer.step(self.gbuf[:self.params.numel()])
        return self.loss

    def step(self, observations, eps=None, **labels) -> float:
        return float(self.step_async(observations, eps, **labels).item())

    def evaluate_loss(self, observations, eps=None, **labels) -> float:
        obs, u, eps = self._inputs(observations, eps, labels)
        self.engine.elbo_step(self.params, obs, u, eps, self.loss, None)
        if self.world > 1:
            torch.distributed.all_reduce(self.loss, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        return float(self.loss.item())
