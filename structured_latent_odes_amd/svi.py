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
        # parameters appended after engine.n_params (auxiliary heads) get no main-loss gradient: that region of gbuf is
        # zero-initialised and never written by slode_elbo_step, so it needs no per-step fill
        opt = self.optimizer
        if self.world == 1 and opt is not None and hasattr(self.engine, "elbo_adam_step"):
            # single process: Adam applied by the final gradient-reduction kernel (slode_elbo_adam_step)
            opt.t += 1
            self.engine.elbo_adam_step(self.params, obs, u, eps, self.loss, self.grads, opt.exp_avg, opt.exp_avg_sq, opt.lr, opt.t,
                                       opt.betas, opt.eps)
            return self.loss
        self.engine.elbo_step(self.params, obs, u, eps, self.loss, self.grads)
        if self.world > 1:
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        if opt is not None:
            opt.step(self.gbuf[:self.params.numel()])
        return self.loss

    def step(self, observations, eps=None, **labels) -> float:
        return float(self.step_async(observations, eps, **labels).item())

    def evaluate_loss(self, observations, eps=None, **labels) -> float:
        obs, u, eps = self._inputs(observations, eps, labels)
        self.engine.elbo_step(self.params, obs, u, eps, self.loss, None)
        if self.world > 1:
            torch.distributed.all_reduce(self.loss, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        return float(self.loss.item())


# ------------------------------------------------------------------------------------------------------------------------
# Name-compatible stand-ins for the three Pyro objects the reference training scripts construct (training_cvs.py:226-249):
#     optimizer = Adam({"lr": ..., "betas": (0.9, 0.999)});  elbo = Trace_ELBO(num_particles=1)
#     loss_basic = SVI(var_model.model, var_model.guide, optimizer, loss=elbo)
#     loss_aux   = SVI(var_model.model_meta, var_model.guide_meta, optimizer, loss=elbo)
# ------------------------------------------------------------------------------------------------------------------------
class Adam:
    """pyro.optim.Adam stand-in: one optimizer object shared by both SVI objects, state created on first use."""

    def __init__(self, optim_args):
        self.optim_args = dict(optim_args)
        self._flat: Optional[FlatAdam] = None

    def for_binding(self, binding) -> FlatAdam:
        if self._flat is None:
            a = self.optim_args
            self._flat = FlatAdam(binding.engine, binding.flat, lr=a.get("lr", 1e-3), betas=tuple(a.get("betas", (0.9, 0.999))),
                                  eps=a.get("eps", 1e-8))
            # pyro.optim.Adam keeps one torch.optim.Adam per parameter.  Both SVI objects register every parameter, so each is
            # stepped twice per minibatch (with a zero gradient by the loss that does not use it) -- except that the label heads of
            # the cvs / challenge families have no gradient at all in the very first main step and are skipped there: their step
            # count runs one behind (training_cvs.py:147-157, 236-249).
            lo, hi = binding.engine.aux_only_region()
            binding.engine.adam_region(lo, hi, -1)
        return self._flat


class Trace_ELBO:
    """pyro.infer.Trace_ELBO stand-in; only num_particles=1 (the reference's setting, config_cvs.py:44) is supported."""

    def __init__(self, num_particles=1, **kwargs):
        if num_particles != 1:
            raise NotImplementedError("num_particles != 1")
        self.num_particles = num_particles


class AuxStep:
    """-ELBO of SVI(model_meta, guide_meta) (mechanistic_cvs.py:240-276): one ``slode_aux_step`` call (encoder forward -> group
    latents sampled in the model + label heads at aux_loss_multiplier -> encoder backward -> reduction [+ Adam]), all HIP."""

    def __init__(self, owner, optimizer: Optional[FlatAdam], process_group=None):
        self.owner, self.optimizer, self.pg = owner, optimizer, process_group
        b = owner._bind()
        self.engine, self.params = b.engine, b.flat
        self.world = torch.distributed.get_world_size(process_group) if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.gbuf = torch.zeros(b.n_total + 1, dtype=torch.float32, device=b.flat.device)
        self.grads, self.loss = self.gbuf[:b.n_total], self.gbuf[b.n_total:]

    def _inputs(self, observations, eps, labels):
        o = self.owner
        if eps is None:
            eps = o.draw_eps(observations.shape[0], self.params.device)
        return observations, o.labels_to_u(**labels), eps

    def step_async(self, observations, eps=None, **labels):
        obs, u, eps = self._inputs(observations, eps, labels)
        opt = self.optimizer
        if self.world == 1 and opt is not None:
            opt.t += 1
            self.engine.aux_step(self.params, obs, u, eps, self.loss, self.grads, adam=(opt.exp_avg, opt.exp_avg_sq, opt.lr, opt.t, opt.betas, opt.eps))
            return self.loss
        self.engine.aux_step(self.params, obs, u, eps, self.loss, self.grads)
        if self.world > 1:
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        if opt is not None:
            opt.step(self.grads)
        return self.loss

    def step(self, observations, eps=None, **labels) -> float:
        return float(self.step_async(observations, eps, **labels).item())

    def evaluate_loss(self, observations, eps=None, **labels) -> float:
        obs, u, eps = self._inputs(observations, eps, labels)
        self.engine.aux_step(self.params, obs, u, eps, self.loss, None)
        if self.world > 1:
            torch.distributed.all_reduce(self.loss, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        return float(self.loss.item())


class SVI:
    """pyro.infer.SVI stand-in: ``SVI(var_model.model, var_model.guide, optimizer, loss=elbo)``.  ``step(**batch)`` /
    ``evaluate_loss(**batch)`` return -ELBO summed over the batch, like the reference call sites expect
    (training_cvs.py:81,152-155)."""

    def __init__(self, model, guide, optim, loss=None, **kwargs):
        owner = getattr(model, "__self__", None)
        if owner is None or not hasattr(owner, "_bind"):
            raise TypeError("SVI expects bound methods of a MechanisticModel (model/guide or model_meta/guide_meta)")
        self.owner, self.kind = owner, ("aux" if model.__name__ == "model_meta" else "main")
        binding = owner._bind()
        flat_opt = optim.for_binding(binding) if optim is not None else None
        if self.kind == "main":
            self._impl = ELBOStep(binding.engine, binding.flat, flat_opt)
        else:
            self._impl = AuxStep(owner, flat_opt)

    def _split(self, batch):
        batch = dict(batch)
        obs = batch.pop("observations")
        eps = batch.pop("eps", None)
        return obs, eps, batch

    def _main_args(self, obs, eps, labels):
        o = self.owner
        if eps is None:
            eps = o.draw_eps(obs.shape[0], obs.device)
        return obs, eps, o.labels_to_u(**labels)

    def step(self, **batch) -> float:
        obs, eps, labels = self._split(batch)
        if self.kind == "aux":
            return self._impl.step(obs, eps=eps, **labels)
        obs, eps, u = self._main_args(obs, eps, labels)
        return self._impl.step(obs, eps=eps, u=u)

    def evaluate_loss(self, **batch) -> float:
        obs, eps, labels = self._split(batch)
        if self.kind == "aux":
            return self._impl.evaluate_loss(obs, eps=eps, **labels)
        obs, eps, u = self._main_args(obs, eps, labels)
        return self._impl.evaluate_loss(obs, eps=eps, u=u)
