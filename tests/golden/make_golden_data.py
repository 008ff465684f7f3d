"""Generates tests/golden/g5_data.npz: outputs of the REFERENCE's own data loaders on its shipped data files.

Build container only (needs /root/reference; never runs on the GPU box, never imported by the product).  It imports the reference's
modules as they are -- utils.ODE_dataset (ODEDataCSV, ODEDataChallenge, create_transforms), utils.utils.find_norm_params,
utils.proc_dataset (TimeSeriesDataset, build_datasets, merge_observations, scale_data), data.proc.load_proc_data.load,
data.challenge.challenge_data.build_datasets -- with two accommodations recorded here:
  * `munch` is not installed: an inert module object stands in for its import (`from munch import munchify` at the top of the two
    config files); nothing of it is ever called -- the configs are rebuilt below as plain attribute namespaces with the values of
    data/{cvs,challenge,proc}/config_*.py;
  * the .pkl files are numpy pickles (opcodes inspected, numpy reconstruct only); torch >= 2.6 refuses them under its
    default weights_only=True, so torch.load is called with weights_only=False for these three files.
The .npz holds DATA only: split ids, sizes, scales, time grids, the first samples of every split and float64 checksums.
Usage: python tests/golden/make_golden_data.py
"""
import functools
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g5_data.npz")


class NS(types.SimpleNamespace):
    pass


def proc_config():
    devices = ["Pcat_Y81C76", "RS100S32_Y81C76", "RS100S34_Y81C76", "R33S32_Y81C76", "R33S34_Y81C76", "R33S175_Y81C76"]
    groups = OrderedDict([("aR", [0, 1, 1, 2, 2, 2]), ("aS", [0, 1, 2, 1, 2, 3])])
    data = NS(devices=devices, normalize=None, subtract_background=True, conditions=["C6", "C12"],
              files=["proc140916.csv", "proc140930.csv", "proc141006.csv", "proc141021.csv", "proc141023.csv", "proc141028.csv"],
              signals=["OD", "mRFP1", "EYFP", "ECFP"], dtype="float32")
    data.component_maps = OrderedDict((k, OrderedDict(zip(devices, g))) for k, g in groups.items())
    data.device_depth = sum(len(set(g)) for g in groups.values())
    data.device_map = dict(zip(devices, (float(v) for v in range(len(devices)))))
    data.device_idx_to_device_name = dict(enumerate(devices))
    return NS(data_path="data/proc/", seed=12, heldout=None, folds=4, split=1, data=data)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.modules.setdefault("munch", types.ModuleType("munch"))
    sys.modules["munch"].munchify = lambda d: d                      # imported by the config modules, never called here
    import utils.ODE_dataset as OD
    from utils.utils import find_norm_params
    import utils.proc_dataset as PD
    from data.proc.load_proc_data import load
    import data.challenge.challenge_data as CD
    OD.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    OD.torch.load = functools.partial(torch.load, weights_only=False)
    out = {}

    # ---- cvs: training_cvs.py:168-192 -------------------------------------------------------------------------------------
    cfg = NS(data_path=REF + "/data/cvs/", norm="zero_to_one", seq_len=86)
    tr = OD.create_transforms(cfg)
    for split in ("train", "val", "test"):
        ds = OD.ODEDataCSV(data_dir=cfg.data_path, ds_type=split, seq_len=cfg.seq_len, random_start=False, transforms=tr)
        out["cvs.%s.len" % split] = np.array(len(ds))
        first = [ds[i] for i in range(3)] + [ds[len(ds) - 1]]
        out["cvs.%s.obs" % split] = np.stack([s["observations"].numpy() for s in first])
        out["cvs.%s.iext" % split] = np.array([float(s["iext"]) for s in first])
        out["cvs.%s.rtpr" % split] = np.array([float(s["rtpr"]) for s in first])
        out["cvs.%s.label_sums" % split] = np.array([float((ds.iext >= 0).float().sum()), float((ds.rtpr > 0).float().sum())])
    out["cvs.norm.min"] = np.asarray(tr["normalize"].min_val, dtype=np.float64)
    out["cvs.norm.max"] = np.asarray(tr["normalize"].max_val, dtype=np.float64)

    # ---- challenge: training_challenge.py:221-240, data/challenge/challenge_data.py:30-54 --------------------------------------
    ccfg = NS(seed=12, folds=5, split=5, norm="zero_to_one", seq_len=142)
    pair = CD.build_datasets(ccfg)
    out["challenge.n_train"], out["challenge.n_test"], out["challenge.max_time"] = np.array(pair.n_train), np.array(pair.n_test), np.array(pair.max_time)
    for k, v in pair.data_norm_params.items():
        out["challenge.norm.%s" % k] = np.asarray(v, dtype=np.float64)
    for name, d in (("train", pair.train), ("test", pair.test)):
        out["challenge.%s.obs_sum" % name] = np.array(np.asarray(d["observations"], dtype=np.float64).sum())
        out["challenge.%s.shedding" % name] = np.asarray(d["shedding"], dtype=np.float64)
        out["challenge.%s.symptoms" % name] = np.asarray(d["symptoms"], dtype=np.float64)
    ctr = OD.create_transforms(ccfg, data_norm_params=pair.data_norm_params)
    dsc = OD.ODEDataChallenge(pair.test, ds_type="val", seq_len=ccfg.seq_len, random_start=False, transforms=ctr)
    out["challenge.val.obs"] = np.stack([dsc[i]["observations"].numpy() for i in range(min(3, len(dsc)))])
    out["norm_params.synthetic_in"] = np.random.RandomState(0).randn(5, 7, 3)
    for k, v in find_norm_params(out["norm_params.synthetic_in"]).items():
        out["norm_params.synthetic.%s" % k] = np.asarray(v)

    # ---- proc: utils/proc_dataset.py:76-204, data/proc/load_proc_data.py:69-125 -----------------------------------------------
    pcfg = proc_config()
    for f in pcfg.data.files:
        dev, treat, times, obs = load(f, pcfg)
        out["proc.file.%s.devices" % f] = dev
        out["proc.file.%s.treatments" % f] = treat
        out["proc.file.%s.times" % f] = times
        out["proc.file.%s.obs_shape" % f] = np.array(obs.shape)
        out["proc.file.%s.obs_sum" % f] = np.array(obs.astype(np.float64).sum())
        out["proc.file.%s.obs_first" % f] = obs[:2]
    ds = PD.TimeSeriesDataset(pcfg, load)
    ds.init_multiple_merge()
    out["proc.devices"] = np.asarray(ds.devices)
    out["proc.inputs"] = ds.inputs.numpy()
    out["proc.times"] = ds.times.numpy()
    out["proc.scales"] = np.asarray(ds.scales, dtype=np.float64)
    out["proc.obs_shape"] = np.array(ds.observations.shape)
    out["proc.obs_sum"] = np.array(ds.observations.double().sum().item())
    out["proc.obs_first"] = ds.observations[:3].numpy()
    out["proc.obs_last"] = ds.observations[-2:].numpy()
    out["proc.dev_1hot"] = ds.dev_1hot.numpy()
    for split in (1, 3):
        pcfg.split, pcfg.heldout = split, None
        pr = PD.build_datasets(pcfg)
        out["proc.fold%d.train_ids" % split] = np.asarray(pr.train.indices)
        out["proc.fold%d.val_ids" % split] = np.asarray(pr.test.indices)
    pcfg.heldout = "R33S34_Y81C76"
    pr = PD.build_datasets(pcfg)
    out["proc.heldout.train_ids"], out["proc.heldout.val_ids"] = np.asarray(pr.train.indices), np.asarray(pr.test.indices)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "%d arrays, %d bytes" % (len(out), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
