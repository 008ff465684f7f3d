#!/usr/bin/env python3
"""Entry point mirroring the reference's ``training_cvs.py`` (``train(config)``, ``run_batch``, ``batch_to_device``,
``input_pred_stats``) on the slode engine: the two ``SVI`` objects are :class:`structured_latent_odes_amd.svi.SVI`
(HIP ELBO step + HIP Adam) instead of Pyro's.  The reference's CSV/pickle data loaders are out of scope:
batches are synthetic CVS-shaped tensors (``structured_latent_odes_amd.synthetic``) unless a loader is supplied.

    python training_cvs.py [--epochs N] [--batches-per-epoch M]
"""
import argparse
import logging
import os

import numpy as np
import torch

from structured_latent_odes_amd.configs import load_config_cvs as load_config
from structured_latent_odes_amd.models.mechanistic_cvs import MechanisticModel
from structured_latent_odes_amd.models.mechanistic_cvs_Gauss import MechanisticModelGauss
from structured_latent_odes_amd.svi import SVI, Adam, Trace_ELBO
from structured_latent_odes_amd.synthetic import synthetic_batch
from structured_latent_odes_amd.utils.utils import set_seed


def batch_to_device(d, device):
    """training_cvs.py:18-27: labels -> [B,1]; observations [B,T,C] -> the [B,C,T] permuted view (no copy)."""
    out = dict(d)
    out["iext"] = d["iext"].reshape(-1, 1).to(device)
    out["rtpr"] = d["rtpr"].reshape(-1, 1).to(device)
    obs = d["observations"]
    out["observations"] = (obs.permute(0, 2, 1) if obs.shape[1] != 3 else obs).to(device)
    return out


def run_batch(batch, losses):
    """training_cvs.py:147-157: one step of every SVI object; returns the per-trajectory losses."""
    epoch_losses = [0.0] * len(losses)
    for i, loss in enumerate(losses):
        new_loss = loss.step(observations=batch["observations"], iext=batch["iext"], rtpr=batch["rtpr"])
        epoch_losses[i] += new_loss / batch["observations"].shape[0]
    return epoch_losses


def input_pred_stats(batches, model, losses, is_post, device):
    """training_cvs.py:43-144 without plotting: -ELBO per trajectory, reconstruction L1, label accuracies."""
    total_elbo, total_l1, size, acc = [0.0] * len(losses), 0.0, 0, {"iext": 0.0, "rtpr": 0.0}
    for batch in batches:
        batch = batch_to_device(batch, device)
        B = batch["observations"].shape[0]
        for i, loss in enumerate(losses):
            total_elbo[i] += loss.evaluate_loss(observations=batch["observations"], iext=batch["iext"], rtpr=batch["rtpr"]) / B
        res = model.recon(observations=batch["observations"], iext=batch["iext"], rtpr=batch["rtpr"], is_post=is_post)
        total_l1 += float(res["l1"])
        pred = model.classifier(observations=batch["observations"])
        for k in acc:
            acc[k] += float((pred[k] == batch[k]).float().sum())
        size += B
    return {"iext": acc["iext"] / size, "rtpr": acc["rtpr"] / size, "l1": total_l1 / size, "elbo": torch.tensor(total_elbo)}


def make_batches(config, n_batches, seed):
    out = []
    for i in range(n_batches):
        obs, labels, _ = synthetic_batch("cvs", config.mini_batch_size, config.seq_len, config.obs_dim, seed=seed + i)
        out.append({"observations": obs, "iext": labels["iext"], "rtpr": labels["rtpr"]})
    return out


def train(config, batches_per_epoch=7):
    set_seed(config.seed)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    times = torch.arange(0.0, end=config.seq_len * config.delta_t, step=config.delta_t, device=device)
    if config.model == "Mechanistic":
        selected = MechanisticModel
    elif config.model == "MechanisticGauss":
        selected = MechanisticModelGauss
    else:
        raise ValueError("selected model is not implemented")
    var_model = selected(config=config, device=device, times=times).to(device)
    msg = "Model: %s -  with %d parameters." % (config.model, sum(p.numel() for p in var_model.parameters()))
    print(msg)
    logging.debug(msg)
    best_model = selected(config=config, device=device, times=times).to(device)
    optimizer = Adam({"lr": config.learning_rate, "betas": (0.9, 0.999)})
    elbo = Trace_ELBO(num_particles=config.num_particles)
    losses = [SVI(var_model.model, var_model.guide, optimizer, loss=elbo),
              SVI(var_model.model_meta, var_model.guide_meta, optimizer, loss=elbo)]
    train_b = make_batches(config, batches_per_epoch, seed=1000)
    val_b = make_batches(config, 1, seed=5000)
    best_val_loss, best_epoch = np.inf, 0
    for epoch in range(config.num_epochs + 1):
        epoch_loss = [run_batch(batch_to_device(b, device), losses) for b in train_b]
        val = input_pred_stats(val_b, var_model, losses, True, device)
        trn = input_pred_stats(train_b[:1], var_model, losses, True, device)
        val_elbo = torch.sum(val["elbo"]) * len(val["elbo"])
        improved = ""
        if best_val_loss >= val_elbo:
            best_val_loss, best_epoch, improved = val_elbo, epoch, "*"
            best_model.load_state_dict(var_model.state_dict())
        line = "[Epoch %d/%d] loss= %.4f  iext_acc=(%.4f,%.4f)  rtpr_acc=(%.4f,%.4f) l1=(%.6f,%.6f), %s" % (
            epoch, config.num_epochs, float(np.mean(epoch_loss)), trn["iext"], val["iext"], trn["rtpr"], val["rtpr"], trn["l1"], val["l1"], improved)
        print(line)
        logging.debug(line)
    return var_model, best_model, best_epoch


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batches-per-epoch", type=int, default=7)
    a = ap.parse_args()
    config = load_config()
    config.num_epochs = a.epochs
    os.makedirs("results_%s" % config.model, exist_ok=True)
    logging.basicConfig(filename="results_%s/model.log" % config.model, filemode="w", level=logging.DEBUG)
    train(config, a.batches_per_epoch)
