"""Shared implementation of the reference's six ``MechanisticModel[Gauss]`` classes (models/mechanistic_{cvs,proc,challenge}
[_Gauss].py).  The per-dataset modules only declare their label schema and attribute names."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from ..engine import AuxHead, ModelSpec, PriorGroup
from ..utils.exp import Exp
from .decoders import Decoder, GaussianDecoder
from .encoder_conv import EncoderCONV
from .encoder_mlp import EncoderMLP


class MechanisticBase(nn.Module):
    """Subclasses set:
      FAMILY        'cvs' | 'challenge' | 'proc'
      GAUSS         bool
      LABELS        ordered label names as the prior nets see them (columns of u), with their dims from config
      PRIORS        [(attr, [label names], [z group names])]            conditional prior nets p(z_g | u_g)
      AUX           [(attr, z group name, label name, kind)]            auxiliary heads q(label | z_g); kind in sigmoid|softmax|expexp
      Z_GROUPS      ordered latent group names (config attr = 'z_<name>_dim'); the last one is 'epsilon'
    """
    FAMILY, GAUSS = "", False
    LABELS: Tuple[str, ...] = ()
    PRIORS: List = []
    AUX: List = []
    Z_GROUPS: Tuple[str, ...] = ()
    LABELS_IN_MAIN = False   # proc: the main model also scores the labels on the replayed z (mechanistic_proc.py:145-146)

    def __init__(self, config, device, times):
        super().__init__()
        self.config, self.times, self.device = config, times, device
        self.obs_dim = config.obs_dim
        self.n_time = len(times)
        self.aux_loss_multiplier = float(config.aux_loss_multiplier)
        self.u_hidden_dim = config.u_hidden_dim
        self.use_cuda = True            # SURVEY note A: always truthy in the reference => hidden layers carry '.module.'
        self.allow_broadcast = False
        self.z_dims = {g: int(getattr(config, "z_%s_dim" % g)) for g in self.Z_GROUPS}
        self.z_off, off = {}, 0
        for g in self.Z_GROUPS:
            self.z_off[g] = off
            off += self.z_dims[g]
        self.latent_dim = off
        self.z_epsilon_dim = self.z_dims["epsilon"]
        self.label_dims = {l: int(getattr(config, "%s_dim" % l)) for l in self.LABELS}
        self.u_off, off = {}, 0
        for l in self.LABELS:
            self.u_off[l] = off
            off += self.label_dims[l]
        self.n_u = off
        self.setup_networks()
        self.l1_func = nn.L1Loss()
        self._binding = None
        self.to(device)

    # ---- construction -------------------------------------------------------------------------------------
    def setup_networks(self):
        cfg = self.config
        for attr, group, label, kind in self.AUX:
            zd, ld = self.z_dims[group], self.label_dims[label]
            if kind == "sigmoid":
                net = EncoderMLP([zd, self.u_hidden_dim, ld], activation=nn.Softplus, output_activation=nn.Sigmoid,
                                 allow_broadcast=False, use_cuda=self.use_cuda)
            elif kind == "softmax":
                net = EncoderMLP([zd, self.u_hidden_dim, ld], activation=nn.Softplus, output_activation=nn.Softmax,
                                 allow_broadcast=False, use_cuda=self.use_cuda)
            else:
                net = EncoderMLP([zd, self.u_hidden_dim, [ld, ld]], activation=nn.Softplus, output_activation=[Exp, Exp],
                                 allow_broadcast=False, use_cuda=self.use_cuda)
            setattr(self, attr, net)
        self.encoder = EncoderCONV(n_channels=self.obs_dim, n_time=self.n_time, n_filters=cfg.n_filters,
                                   filter_size=cfg.filter_size, pool_size=cfg.pool_size, latent_dim=self.latent_dim,
                                   hidden_dim=cfg.cnn_hidden_dim)
        for attr, labels, groups in self.PRIORS:
            n_in = sum(self.label_dims[l] for l in labels)
            n_out = sum(self.z_dims[g] for g in groups)
            setattr(self, attr, EncoderMLP([n_in, [n_out, n_out]], activation=nn.Softplus, output_activation=[None, Exp],
                                           allow_broadcast=False, use_cuda=self.use_cuda))
        dec = GaussianDecoder if self.GAUSS else Decoder
        self.decoder = dec(config=cfg, times=self.times, latent_dim=self.latent_dim, device=self.device)
        if self.FAMILY == "proc":
            self.constant_std_C_12 = nn.Parameter(torch.ones(1) * cfg.constant_std, requires_grad=True)
            self.constant_std_C_6 = nn.Parameter(torch.ones(1) * cfg.constant_std, requires_grad=True)
            self.softplus = nn.Softplus()

    def model_spec(self) -> ModelSpec:
        cfg = self.config
        groups = []
        for attr, labels, zgroups in self.PRIORS:
            groups.append(PriorGroup(attr, self.z_off[zgroups[0]], sum(self.z_dims[g] for g in zgroups),
                                     self.u_off[labels[0]], sum(self.label_dims[l] for l in labels)))
        aux = []
        for attr, group, label, kind in self.AUX:
            aux.append(AuxHead(attr, kind, self.z_off[group], self.z_dims[group], self.u_off[label], self.label_dims[label],
                               "constant_std_C_12" if label == "C12" else ("constant_std_C_6" if label == "C6" else "")))
        return ModelSpec(self.FAMILY, self.GAUSS, self.obs_dim, self.latent_dim, self.z_epsilon_dim, self.n_u, groups,
                         aux_heads=aux, labels_in_main=self.LABELS_IN_MAIN, u_hidden_dim=self.u_hidden_dim, aux_mult=self.aux_loss_multiplier,
                         ode_state_dim=cfg.ode_state_dim, ode_hidden_dim=cfg.ode_hidden_dim, n_filters=cfg.n_filters,
                         filter_size=cfg.filter_size, pool_size=cfg.pool_size, cnn_hidden_dim=cfg.cnn_hidden_dim,
                         solver=cfg.solver, quantile_diff=cfg.quantile_diff,
                         # config.adjoint_solver (True in all three reference configs) selects torchdiffeq.odeint_adjoint
                         # (models/blackbox_ode.py:40-42): its gradients are reproduced by grad_mode "reference_adjoint"
                         grad_mode="reference_adjoint" if getattr(cfg, "adjoint_solver", False) else "exact")

    def _bind(self):
        """Create the engine and move every parameter into the flat vector (first hot-path use; needs a HIP device)."""
        if self._binding is None:
            from ._binding import Binding
            named = self.encoder._named_for_binding()
            named.update(self.decoder._named_for_binding())
            for attr, _, _ in self.PRIORS:
                net = getattr(self, attr)
                for k, p in net.named_parameters():
                    named["%s.%s" % (attr, k)] = p
            for attr, _, _, _ in self.AUX:   # label heads: auxiliary-loss kernel (and the main loss for proc) => in the layout
                for k, p in getattr(self, attr).named_parameters():
                    named["%s.%s" % (attr, k)] = p
            if self.FAMILY == "proc":
                named["constant_std_C_12"], named["constant_std_C_6"] = self.constant_std_C_12, self.constant_std_C_6
            hot = set(id(p) for p in named.values())
            extra = [p for p in self.parameters() if id(p) not in hot]
            dev = next(self.parameters()).device
            self._binding = Binding(self.model_spec(), self.times.to(torch.float32), dev, named, extra)
            self.encoder._binding = self._binding
            self.decoder.ode_model._binding = self._binding
        return self._binding

    def load_state_dict(self, *args, **kwargs):
        """nn.Module.load_state_dict + a note to the engine: the weights were written behind its back, so the next step must fold the
        encoder again (the kept fold belongs to the old weights; include/slode.h, slode_fold_invalidate)."""
        res = super().load_state_dict(*args, **kwargs)
        if self._binding is not None:
            self._binding.engine.fold_invalidate()
        return res

    def parameters_changed(self):
        """Call after writing parameters by any other route (own optimizer, manual edits) between two SVI steps."""
        if self._binding is not None:
            self._binding.engine.fold_invalidate()

    # ---- helpers ------------------------------------------------------------------------------------------
    def labels_to_u(self, **labels) -> torch.Tensor:
        return torch.cat([labels[l].reshape(labels[l].shape[0], -1).to(torch.float32) for l in self.LABELS], dim=1).contiguous()

    def draw_eps(self, batch_size: int, device) -> torch.Tensor:
        """Reparameterisation noise in the guide's site order (one ``randn`` per ``pyro.sample`` site)."""
        parts = [torch.randn(batch_size, n, device=device) for n in self._site_dims()]
        return torch.cat(parts, dim=1)

    def _site_dims(self) -> List[int]:
        if self.FAMILY == "cvs":
            return [self.z_dims[g] for g in self.Z_GROUPS]                       # z_iext, z_rtpr, z_epsilon
        return [self.latent_dim - self.z_epsilon_dim, self.z_epsilon_dim]        # z_u, z_epsilon

    def _prior_loc_scale(self, labels: Dict[str, torch.Tensor]):
        """(loc, scale) [B, L] of p(z | labels): the conditional prior nets on the label columns, N(0, 1) for the z_epsilon dims -- one
        HIP kernel (``slode_prior_nets``), the same nets the fused ELBO kernel evaluates in its P0 phase."""
        b = self._bind()
        return b.engine.prior_nets(b.flat, self.labels_to_u(**labels))

    def _z_group(self, t: torch.Tensor, g: str) -> torch.Tensor:
        return t[:, self.z_off[g]:self.z_off[g] + self.z_dims[g]]

    # ---- the four callables the training scripts hand to SVI (training_cvs.py:236-249) -----------------------
    def model(self, observations, **labels):
        """Stands for the Pyro model (mechanistic_cvs.py:105-178); together with ``guide`` it defines the main loss, whose
        arithmetic is ``slode_elbo_step``.  Called directly it returns -ELBO (summed over the batch) for fresh noise."""
        from ..svi import SVI
        return SVI(self.model, self.guide, None).evaluate_loss(observations=observations, **labels)

    def guide(self, observations, **labels):
        """q(z | x): encoder + one reparameterised Normal per latent group (mechanistic_cvs.py:213-238)."""
        b = self._bind()
        loc, scale = self.encoder.forward(observations)
        z = b.engine.sample_normal(loc.contiguous(), scale.contiguous())
        return tuple(self._z_group(z, g) for g in self.Z_GROUPS)

    def model_meta(self, observations, **labels):
        """Auxiliary supervised loss (mechanistic_cvs.py:240-270); arithmetic in :class:`svi.AuxStep`."""
        from ..svi import SVI
        return SVI(self.model_meta, self.guide_meta, None).evaluate_loss(observations=observations, **labels)

    def guide_meta(self, observations, **labels):
        """Empty guide accompanying ``model_meta`` (mechanistic_cvs.py:272-276)."""
        return None

    # ---- eval-side API -------------------------------------------------------------------
    def _predict_labels(self, observations):
        """classifier / pred_inputs of the reference (mechanistic_cvs.py:278-296): encoder -> one posterior draw -> label heads
        (``slode_label_heads``: every head writes the label columns it scores) -> hard decisions."""
        b = self._bind()
        with torch.no_grad():
            loc, scale = self.encoder.forward(observations)
            z = b.engine.sample_normal(loc.contiguous(), scale.contiguous())
            probs = b.engine.label_heads(b.flat, z)
            res = {}
            heads = {h.prefix: h for h in b.engine.spec.aux_heads}
            for attr, group, label, kind in self.AUX:
                head = heads[attr]
                val = probs[:, head.u_off:head.u_off + head.u_dim]
                if kind == "sigmoid":
                    res[label] = (val > 0.5).float()
                elif kind == "softmax":
                    res[label] = torch.zeros_like(val).scatter_(1, val.argmax(1, keepdim=True), 1.0)
                else:
                    res[label] = val.clone()
            return res

    def recon(self, observations, is_post, **labels):
        """Posterior (is_post) or prior reconstruction (mechanistic_cvs.py:298-323): returns the reference's dict."""
        b = self._bind()
        with torch.no_grad():
            if is_post:
                loc, scale = self.encoder.forward(observations)
                z = b.engine.sample_normal(loc.contiguous(), scale.contiguous())
            else:
                ploc, pscale = self._prior_loc_scale(labels)      # [B, L]: conditional groups, then (0, 1) for z_epsilon
                z = b.engine.sample_normal(ploc, pscale)
            if self.GAUSS:
                solution_xt, mean, std = self.decoder.forward(z=z)
                return {"l1": self.l1_func(mean, observations), "solution_xt": solution_xt, "mean": mean, "std": std, "z": z}
            solution_xt, mu_75, mu_50, mu_25, std = self.decoder.forward(z=z)
            return {"l1": self.l1_func(mu_50, observations), "solution_xt": solution_xt, "mu_75": mu_75, "mu_50": mu_50,
                    "mu_25": mu_25, "std": std, "z": z}

    def recon_samples(self, observations, is_post, num_samples: int, eps=None, **labels):
        """``multiple_samples`` of the reference (training_proc.py:205-223, training_cvs.py / training_challenge.py alike): it calls
        ``recon`` ``num_samples`` times (config.num_samples = 200) and concatenates the quantile curves along a new last axis.
        Here the ``num_samples`` latent draws of the whole batch go through ONE ODE solve + ONE head launch
        (``num_samples * B`` trajectories).  Returns a dict of tensors shaped ``[B, C, T, num_samples]`` (``mu_25/mu_50/mu_75``, or
        ``mean`` for the Gaussian family) plus ``z`` ``[num_samples, B, L]``.  ``eps`` (``[num_samples, B, L]`` standard normal
        draws, optional) makes the result reproducible."""
        self._bind()
        with torch.no_grad():
            B, ns = observations.shape[0], int(num_samples)
            if is_post:
                loc, scale = self.encoder.forward(observations)
            else:
                loc, scale = self._prior_loc_scale(labels)
            if eps is None:
                eps = self._bind().engine.draw_normal(ns * B).view(ns, B, loc.shape[1])
            z = loc.unsqueeze(0) + scale.unsqueeze(0) * eps.to(loc.device)               # [ns, B, L]
            out = self.decoder.forward(z=z.reshape(ns * B, -1).contiguous())
            names = ("solution_xt", "mean", "std") if self.GAUSS else ("solution_xt", "mu_75", "mu_50", "mu_25", "std")
            res = {"z": z}
            for name, val in zip(names, out):
                if name in ("solution_xt", "std"):
                    continue
                res[name] = val.reshape(ns, B, val.shape[1], val.shape[2]).permute(1, 2, 3, 0).contiguous()   # [B, C, T, ns]
            return res

    def save_recon_samples(self, results_dir: str, observations, is_post, num_samples: int, **labels):
        """Writes the arrays ``multiple_samples`` saves, under the reference's file names (``mu_50_post_sample.npy`` ...)."""
        import os
        import numpy as np
        res = self.recon_samples(observations, is_post, num_samples, **labels)
        os.makedirs(results_dir, exist_ok=True)
        tag = "post_sample" if is_post else "prior_sample"
        written = []
        for name in ("mu_50", "mu_75", "mu_25", "mean"):
            if name in res:
                path = os.path.join(results_dir, "%s_%s.npy" % (name, tag))
                np.save(path, res[name].cpu().numpy())
                written.append(path)
        return written

