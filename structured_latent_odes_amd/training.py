"""Shared body of the three entry points (training_cvs.py / training_proc.py / training_challenge.py of the reference):
``train(config)`` with the reference's structure -- two SVI objects sharing one Adam (training_cvs.py:226-249), an epoch loop of
``run_batch`` (:147-157, :256-266), the four statistics passes per epoch (validation / training x posterior / prior: ``evaluate_loss`` +
``recon`` + label prediction, :43-144, :270-315), best-model copy (:325-331), the per-epoch summary line (:336-352) and the final
test passes on the best model (:355-397).  Batches come from ``synthetic.synthetic_batch`` unless ``--data-dir`` points at
the reference's data files (cvs: ``processed_data.pkl`` ...; challenge: ``data.pkl``; proc: the plate-reader CSVs), which are then read by ``data.py`` (SURVEY row
N3) and fed through pinned host buffers, or the caller passes its own list of batch dicts."""
from __future__ import annotations

import logging
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .svi import SVI, Adam, Trace_ELBO
from .synthetic import synthetic_batch
from .utils.utils import set_seed

FAMILY_LABELS = {"cvs": ("iext", "rtpr"), "challenge": ("symptoms", "shedding"), "proc": ("aR", "aS", "C12", "C6")}


def batch_to_device(d: Dict[str, torch.Tensor], device, family: str) -> Dict[str, torch.Tensor]:
    """training_cvs.py:18-27 / training_proc.py:25-33 / training_challenge.py:27-33: labels -> [B, dim]; cvs/challenge
    observations [B,T,C] -> the [B,C,T] permuted view (no copy); proc observations are already [B,C,T]."""
    out = {}
    for l in FAMILY_LABELS[family]:
        v = d[l]
        out[l] = (v.reshape(v.shape[0], -1) if v.dim() > 1 else v.reshape(-1, 1)).to(device)
    obs = d["observations"]
    out["observations"] = obs.to(device)
    return out


def run_batch(batch, losses) -> List[float]:
    """training_cvs.py:147-157."""
    B = batch["observations"].shape[0]
    return [loss.step(**batch) / B for loss in losses]


def input_pred_stats(batches, model, losses, is_post: bool, device, family: str):
    """training_cvs.py:43-144 without the plotting: -ELBO per trajectory for every loss, reconstruction L1, label predictions."""
    total_elbo, total_l1, size = [0.0] * len(losses), 0.0, 0
    hits = {l: 0.0 for l in FAMILY_LABELS[family]}
    predict = getattr(model, "classifier", None) or model.pred_inputs
    for batch in batches:
        batch = batch_to_device(batch, device, family)
        B = batch["observations"].shape[0]
        for i, loss in enumerate(losses):
            total_elbo[i] += loss.evaluate_loss(**batch) / B
        total_l1 += float(model.recon(is_post=is_post, **batch)["l1"])
        pred = predict(observations=batch["observations"])
        for l in hits:
            if pred[l].shape == batch[l].shape:
                hits[l] += float((pred[l] - batch[l]).abs().lt(0.5).all(dim=1).float().sum())
        size += B
    out = {l: hits[l] / max(size, 1) for l in hits}
    out.update(l1=total_l1 / max(size, 1), elbo=torch.tensor(total_elbo))
    return out


def make_batches(config, family: str, n_batches: int, seed: int):
    out = []
    for i in range(n_batches):
        obs, labels, _ = synthetic_batch(family, config.mini_batch_size, config.seq_len, config.obs_dim, seed=seed + i)
        d = {"observations": obs}
        d.update(labels)
        out.append(d)
    return out


def train(config, family: str, model_cls, model_cls_gauss, batches_per_epoch: int = 7,
          train_batches: Optional[Sequence[dict]] = None, val_batches: Optional[Sequence[dict]] = None, times: Optional[torch.Tensor] = None,
          test_batches: Optional[Sequence[dict]] = None):
    set_seed(config.seed)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if times is not None:
        times = times.to(device)
    elif family == "proc":
        _, _, times = synthetic_batch("proc", 1, config.seq_len, config.obs_dim, seed=0)     # non-uniform grid like the CSV times
        times = times.to(device)
    else:
        times = torch.arange(0.0, end=config.seq_len * config.delta_t, step=config.delta_t, device=device)
    if config.model == "Mechanistic":
        selected = model_cls
    elif config.model == "MechanisticGauss":
        selected = model_cls_gauss
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
    # batch sources may be lists or re-iterable feeders (data.BatchFeeder: a fresh pass, reshuffled, every epoch)
    train_b = train_batches if train_batches is not None else make_batches(config, family, batches_per_epoch, seed=1000)
    val_b = val_batches if val_batches is not None else make_batches(config, family, 1, seed=5000)
    test_b = test_batches if test_batches is not None else val_b          # cvs ships a test split; challenge / proc test on the validation fold
    best_val_loss, best_epoch = np.inf, 0
    names = FAMILY_LABELS[family][:2]
    for epoch in range(config.num_epochs + 1):
        t_ep, n_traj, epoch_loss = time.perf_counter(), 0, []
        for b in train_b:
            d = batch_to_device(b, device, family)
            epoch_loss.append(run_batch(d, losses))          # (every step ends in the .item() of its loss: the clock sees finished work)
            n_traj += d["observations"].shape[0]
        traj_per_s = n_traj / max(time.perf_counter() - t_ep, 1e-9)
        # the reference's four statistics passes per epoch (training_cvs.py:270-315): validation posterior / prior, training
        # posterior / prior -- every one a full pass over its loader: evaluate_loss of both SVI objects, recon, label prediction
        val = input_pred_stats(val_b, var_model, losses, True, device, family)
        _ = input_pred_stats(val_b, var_model, losses, False, device, family)
        trn = input_pred_stats(train_b, var_model, losses, True, device, family)
        trn_prior = input_pred_stats(train_b, var_model, losses, False, device, family)
        val_elbo = torch.sum(val["elbo"]) * len(val["elbo"])
        improved = ""
        if best_val_loss >= val_elbo:
            best_val_loss, best_epoch, improved = val_elbo, epoch, "*"
            best_model.load_state_dict(var_model.state_dict())
        # the reference's summary line (training_cvs.py:336-352) + the training throughput of the epoch: trajectories through
        # run_batch (main + auxiliary SVI step, two Adam passes) per second of wall time, host-to-device copies included
        line = "[Epoch %d/%d] loss= %.4f  %s_acc=(%.4f,%.4f)  %s_acc=(%.4f,%.4f) l1=(%.6f,%.6f), %s  trajectories/sec=%.0f" % (
            epoch, config.num_epochs, float(np.mean(epoch_loss)), names[0], trn[names[0]], val[names[0]], names[1], trn[names[1]],
            val[names[1]], trn["l1"], val["l1"], improved, traj_per_s)
        print(line)
        logging.debug(line)
    # final test passes on the best model, posterior and prior (training_cvs.py:355-397); the losses stay bound to var_model, as there
    test_post = input_pred_stats(test_b, best_model, losses, True, device, family)
    test_prior = input_pred_stats(test_b, best_model, losses, False, device, family)
    final = "FINAL TEST: %s_acc=(%.4f,%.4f)  %s_acc=(%.4f,%.4f) l1=(%.6f,%.6f)" % (
        names[0], test_post[names[0]], test_prior[names[0]], names[1], test_post[names[1]], test_prior[names[1]], test_post["l1"], test_prior["l1"])
    print(final)
    logging.debug(final)
    tail = "ELBO: best_epoch: {} post: {} prior: {}".format(best_epoch, test_post["elbo"], test_prior["elbo"])
    print(tail)
    logging.debug(tail)
    return var_model, best_model, best_epoch


def real_batches(config, family: str, data_dir: str):
    """(train_batches, val_batches, times or None) read from the reference's data files with the reference's transforms and splits
    (training_cvs.py:168-190, training_challenge.py:226-246); each an iterable of host batch dicts re-read every epoch."""
    from . import data as D
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if family == "cvs":
        tf = D.create_transforms(config.norm, D._torch_load(os.path.join(data_dir, "data_norm_params.pkl")))
        tr = D.CVSDataset(data_dir, "train", config.seq_len, False, tf)
        va = D.CVSDataset(data_dir, "val", config.seq_len, False, tf)
        te = D.CVSDataset(data_dir, "test", config.seq_len, False, tf)
    elif family == "challenge":
        pair = D.build_challenge_datasets(os.path.join(data_dir, "data.pkl"), config.seed, config.folds, config.split)
        tf = D.create_transforms(config.norm, pair.data_norm_params)
        tr, va = D.ChallengeDataset(pair.train, transforms=tf), D.ChallengeDataset(pair.test, transforms=tf)
    else:   # proc: observations are already [C, T]; the (non-uniform) time grid comes with the data
        tr, va, times = D.build_proc_datasets(data_dir, config.seed, config.folds, config.split, getattr(config, "heldout", None))
        config.seq_len = int(times.numel())
        return (D.BatchFeeder(tr, config.mini_batch_size, dev, shuffle=True, seed=config.seed),
                D.BatchFeeder(va, config.mini_batch_size, dev), times)
    class _AsBCT:
        """The datasets yield [B, T, C]; the models take the [B, C, T] permuted VIEW of it (training_cvs.py:25: no copy)."""

        def __init__(self, feeder):
            self.feeder = feeder

        def __len__(self):
            return len(self.feeder)

        def __iter__(self):
            for b in self.feeder:
                b["observations"] = b["observations"].permute(0, 2, 1)
                yield b

    out = (_AsBCT(D.BatchFeeder(tr, config.mini_batch_size, dev, shuffle=True, seed=config.seed)),
           _AsBCT(D.BatchFeeder(va, config.mini_batch_size, dev)), None)
    if family == "cvs":
        out = out + (_AsBCT(D.BatchFeeder(te, config.mini_batch_size, dev)),)
    return out


def main(family: str, load_config, model_cls, model_cls_gauss):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batches-per-epoch", type=int, default=7)
    ap.add_argument("--data-dir", default=None, help="directory with the reference's data files (default: synthetic batches)")
    a = ap.parse_args()
    config = load_config()
    config.num_epochs = a.epochs
    os.makedirs("results_%s" % config.model, exist_ok=True)
    logging.basicConfig(filename="results_%s/model.log" % config.model, filemode="w", level=logging.DEBUG)
    kw = {}
    if a.data_dir:
        got = real_batches(config, family, a.data_dir)
        kw["train_batches"], kw["val_batches"], kw["times"] = got[:3]
        if len(got) > 3:
            kw["test_batches"] = got[3]
    train(config, family, model_cls, model_cls_gauss, a.batches_per_epoch, **kw)
