"""Drop-in for the ``pyro.infer.SVI`` objects the reference training scripts build (training_cvs.py:236-249):
``ELBOStep.step(**batch) -> float`` and ``.evaluate_loss(**batch) -> float`` return -ELBO summed over the batch.

One step = ONE ``slode_svi_step`` call (labels as the loader yields them, noise drawn in the kernels, encoder -> latent sample -> ODE
solve -> likelihood -> exact gradient -> Adam, all HIP) + the ``.item()`` the API demands; data parallel: gradient-only call, one RCCL SUM
all-reduce of the flat gradient with the loss scalar appended, one ``slode_adam_step`` call.  No Pyro, no torchdiffeq,
no autograd, no torch operator on this path."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import _lib as L
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


def _label_list(labels, u):
    """The step's label tensors in the model's concatenation order: a dense ``u`` matrix counts as one tensor of all columns."""
    if labels is not None:
        out = []
        for t in labels:
            if t.dtype != torch.float32:
                t = t.to(torch.float32)
            out.append(t if t.is_contiguous() else t.contiguous())
        return out
    return [u] if u is not None else []


class _StepBase:
    """Shared by the two SVI objects: ONE ``slode_svi_step`` call per step -- the label tensors go in as the loader yields them (no
    concatenation) and, unless ``eps`` is given, the reparameterisation noise is drawn inside the kernels (Philox stream of the engine:
    ``Engine.rng_seed``; the guide's ``rsample`` sites, mechanistic_cvs.py:225-237).  ``eps=`` / ``u=`` keep the explicit parity path.
    Data parallel (``world > 1``): gradient-only call -> one SUM all-reduce of [gradient | loss] -> ``slode_adam_step``."""
    KIND = L.SVI_MAIN

    def _setup(self, engine: Engine, params: torch.Tensor, n_grad: int, optimizer, process_group):
        self.engine, self.params, self.optimizer, self.pg = engine, params, optimizer, process_group
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size(process_group) if dist_on else 1
        self.rank = torch.distributed.get_rank(process_group) if dist_on else 0
        # gradient buffer with the loss scalar appended => a single collective per step
        self.gbuf = torch.zeros(n_grad + 1, dtype=torch.float32, device=params.device)
        self.grads, self.loss = self.gbuf[:n_grad], self.gbuf[n_grad:]
        self._rng_sharded = False
        self.dp_payload = "G"      # "G": reduce [G | head products | ODE-half row] and chain-rule after; "grad": reduce [flat gradient | loss]
        self._payload = None
        self.unfused = False       # True: take the data-parallel code path at world size 1 too (tests: RCCL at world size 1)

    def _batch(self, observations, eps, labels, u):
        eng = self.engine
        if self.world > 1 and not self._rng_sharded and eps is None:
            # in-kernel noise is keyed by the GLOBAL trajectory index: this rank's shard starts at rank * B (contiguous batch split)
            seed, b0, n = eng.rng_state()
            if b0 == 0 and self.rank > 0:
                eng.rng_seed(seed, self.rank * observations.shape[0])
                eng.rng_set_counter(n)
            self._rng_sharded = True
        return eng.make_batch(observations, _label_list(labels, u), eps)

    def _named(self, labels, named):
        """Label tensors given by name (``iext=..., rtpr=...``, as the reference's batches carry them): the owner model's order."""
        if named:
            owner = getattr(self, "owner", None)
            if owner is None or labels is not None:
                raise TypeError("labels by name need an SVI object built from a model, and exclude labels=")
            labels = [named[l].reshape(named[l].shape[0], -1) for l in owner.LABELS]
        return labels

    def step_async(self, observations, eps=None, u=None, labels=None, **named):
        """Enqueue one optimisation step on the current stream; returns the device tensor holding -ELBO (global sum)."""
        B = observations.shape[0]
        bt = self._batch(observations, eps, self._named(labels, named), u)
        opt = self.optimizer
        # parameters appended after engine.n_params get no main-loss gradient: that region of gbuf is zero-initialised and never
        # written by the main step, so it needs no per-step fill
        if self.world == 1 and opt is not None and not self.unfused:
            opt.t += 1   # single process: Adam applied by the final gradient-reduction kernel
            self.engine.svi_step(self.KIND, self.params, bt, B, self.loss, self.grads,
                                 adam=(opt.exp_avg, opt.exp_avg_sq, opt.lr, opt.t, opt.betas, opt.eps))
            return self.loss
        if self.dp_payload == "G" and (self.world > 1 or self.unfused) and self._payload is not False and hasattr(self.engine, "grad_partial"):
            # data parallel, small payload: every rank contributes G = g_pre^T [X | 1], its head-layer products and its ODE-half row with
            # the loss scalar (137 KB instead of the 386 KB flat gradient at the metric shape); the chain rule -- linear in G -- runs once,
            # on the reduced payload, with Adam applied by the same launch (include/slode.h: slode_grad_partial / slode_grad_apply)
            try:
                if self._payload is None:
                    self._payload = torch.zeros(self.engine.payload_floats(self.KIND), dtype=torch.float32, device=self.params.device)
                self.engine.grad_partial(self.KIND, self.params, bt, B, self._payload)
            except L.SlodeError:
                self._payload = False          # (observations the folded encoder path does not take: reduce the flat gradient instead)
            if self._payload is not False:
                if torch.distributed.is_available() and torch.distributed.is_initialized():
                    torch.distributed.all_reduce(self._payload, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                if opt is not None:
                    opt.t += 1
                self.engine.grad_apply(self.KIND, self.params, bt, B, self._payload, self.loss, self.grads,
                                       adam=(opt.exp_avg, opt.exp_avg_sq, opt.lr, opt.t, opt.betas, opt.eps) if opt is not None else None)
                return self.loss
        self.engine.svi_step(self.KIND, self.params, bt, B, self.loss, self.grads)
        if self.world > 1 or (self.unfused and torch.distributed.is_initialized()):
            torch.distributed.all_reduce(self.gbuf, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        if opt is not None:
            opt.step(self.gbuf[:self.params.numel()] if self.grads.numel() != self.params.numel() else self.grads)
        return self.loss

    @property
    def collective_bytes(self) -> int:
        """Bytes this object's step puts through its one all-reduce (after the first data-parallel step)."""
        return int(self._payload.numel() * 4) if isinstance(self._payload, torch.Tensor) else int(self.gbuf.numel() * 4)

    def step(self, observations, eps=None, u=None, labels=None, **named) -> float:
        return float(self.step_async(observations, eps, u, labels, **named).item())

    def evaluate_loss(self, observations, eps=None, u=None, labels=None, **named) -> float:
        B = observations.shape[0]
        self.engine.svi_step(self.KIND, self.params, self._batch(observations, eps, self._named(labels, named), u), B, self.loss, None)
        if self.world > 1:
            torch.distributed.all_reduce(self.loss, op=torch.distributed.ReduceOp.SUM, group=self.pg)
        return float(self.loss.item())


class ELBOStep(_StepBase):
    """Main-loss SVI object, SVI(model, guide) (training_cvs.py:236-243)."""
    KIND = L.SVI_MAIN

    def __init__(self, engine: Engine, params: torch.Tensor, optimizer: Optional[FlatAdam] = None, process_group=None, owner=None):
        self.owner = owner
        self._setup(engine, params, params.numel(), optimizer, process_group)


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


class AuxStep(_StepBase):
    """-ELBO of SVI(model_meta, guide_meta) (mechanistic_cvs.py:240-276): one ``slode_svi_step`` call of kind AUX (encoder forward ->
    group latents sampled in the model + label heads at aux_loss_multiplier -> encoder backward -> reduction [+ Adam]), all HIP."""
    KIND = L.SVI_AUX

    def __init__(self, owner, optimizer: Optional[FlatAdam], process_group=None):
        self.owner = owner
        b = owner._bind()
        self._setup(b.engine, b.flat, b.n_total, optimizer, process_group)


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
            self._impl = ELBOStep(binding.engine, binding.flat, flat_opt, owner=owner)
        else:
            self._impl = AuxStep(owner, flat_opt)

    def _split(self, batch):
        """observations, optional explicit eps, and the label tensors in the model's order -- as the loader yields them, not concatenated."""
        batch = dict(batch)
        obs = batch.pop("observations")
        eps = batch.pop("eps", None)
        o = self.owner
        labels = [batch[l].reshape(batch[l].shape[0], -1) for l in o.LABELS]
        return obs, eps, labels

    def step(self, **batch) -> float:
        obs, eps, labels = self._split(batch)
        return self._impl.step(obs, eps=eps, labels=labels)

    def step_async(self, **batch):
        obs, eps, labels = self._split(batch)
        return self._impl.step_async(obs, eps=eps, labels=labels)

    def evaluate_loss(self, **batch) -> float:
        obs, eps, labels = self._split(batch)
        return self._impl.evaluate_loss(obs, eps=eps, labels=labels)
