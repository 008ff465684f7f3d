#!/usr/bin/env python3
"""Entry point mirroring the reference's ``training_proc.py`` (``train(config)``, ``run_batch``, ``batch_to_device``,
``input_pred_stats``) on the slode engine: both ``SVI`` objects are :class:`structured_latent_odes_amd.svi.SVI` (HIP ELBO / aux
step + HIP Adam) instead of Pyro's.  Synthetic proc-shaped batches unless a loader is supplied.

    python training_proc.py [--epochs N] [--batches-per-epoch M]
"""
from structured_latent_odes_amd import training as _t
from structured_latent_odes_amd.configs import load_config_proc as load_config
from structured_latent_odes_amd.models.mechanistic_proc import MechanisticModel
from structured_latent_odes_amd.models.mechanistic_proc_Gauss import MechanisticModelGauss

FAMILY = "proc"


def batch_to_device(d, device):
    return _t.batch_to_device(d, device, FAMILY)


run_batch = _t.run_batch


def input_pred_stats(batches, model, losses, is_post, device):
    return _t.input_pred_stats(batches, model, losses, is_post, device, FAMILY)


def train(config, batches_per_epoch=7, **kw):
    return _t.train(config, FAMILY, MechanisticModel, MechanisticModelGauss, batches_per_epoch, **kw)


if __name__ == "__main__":
    _t.main(FAMILY, load_config, MechanisticModel, MechanisticModelGauss)
