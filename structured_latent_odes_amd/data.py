"""Real-data front end: the reference's dataset readers, normalisation transforms and splits, feeding ``[B, T, C]``
minibatches through pinned host buffers to the device.

What each piece restates (behaviour, not code):
  * ``CVSDataset``          -- ``utils/ODE_dataset.py:6-56``  (ODEDataCSV): ``processed_data.pkl`` + ``{train,test}_params_data.pkl``,
                               first 90 % of "train" = train, rest = val; labels ``iext = (i_ext >= 0)``, ``rtpr = (r_tpr_mod > 0)``;
                               optional random window start.
  * ``ChallengeDataset``    -- ``utils/ODE_dataset.py:59-84`` (ODEDataChallenge).
  * ``build_challenge_datasets`` -- ``data/challenge/challenge_data.py:30-54``: seeded permutation, ``np.array_split`` into ``folds``,
                               fold ``split`` (1-based) is the validation set; norm params from the training observations.
  * ``find_norm_params``    -- ``utils/utils.py:16-35`` (per-feature mean / population std / max / min over samples and time).
  * ``NormalizeToUnitSegment`` / ``NormalizeZScore`` / ``create_transforms`` -- ``utils/ODE_dataset.py:160-233``.
  * ``kfold_ids`` / ``holdout_ids`` -- the two split rules of ``utils/proc_dataset.py:150-204``.
  * ``ProcDataset`` + ``load_proc_csv`` -- ``data/proc/load_proc_data.py:69-125`` and ``utils/proc_dataset.py:76-140`` for the plate-reader
                               CSVs (device / condition columns, four signal blocks, ``timesall`` row).
``BatchFeeder`` is new: it collates a dataset's samples straight into two pinned host buffers and copies them to the device on a
side stream, so the H2D copy of batch k+1 overlaps the ELBO step of batch k (the path's inputs are otherwise resident in HBM).
Host-side code: no HIP kernels here, nothing on the measured path."""
from __future__ import annotations

import os
import pickle
import random
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ---- normalisation -------------------------------------------------------------------------------------------------------------
def find_norm_params(data: np.ndarray) -> Dict[str, np.ndarray]:
    """Per-feature statistics over (sample, time) of ``data[N, T, C]`` -- the dict the reference pickles as data_norm_params
    (utils/utils.py:16-35).  Each statistic is taken over the feature's own [N, T] slice, so numpy's summation order -- and with it
    the last bit of mean / std -- is the reference's."""
    data = np.asarray(data)
    per = [data[:, :, c] for c in range(data.shape[2])]
    stat = lambda fn: np.array([float(fn(x)) for x in per])
    return {"mean": stat(np.mean), "std": stat(np.std), "max": stat(np.max), "min": stat(np.min)}


class NormalizeToUnitSegment:
    """(x - min) / (max - min) per feature; ``denormalize`` inverts it on ``[B, T, C]`` batches.  The reference applies numpy float64
    scalars to a float32 tensor, feature by feature (utils/ODE_dataset.py:196-209): the range max - min is formed in float64 and only
    then rounded to float32 -- reproduced here so that the samples are bit-identical to the reference loader's."""

    def __init__(self, params):
        lo, hi = np.asarray(params["min"], dtype=np.float64), np.asarray(params["max"], dtype=np.float64)
        self.min_val = torch.as_tensor(lo, dtype=torch.float32)
        self.max_val = torch.as_tensor(hi, dtype=torch.float32)
        self.range = torch.as_tensor(hi - lo, dtype=torch.float32)

    def __call__(self, sample: torch.Tensor) -> torch.Tensor:
        out = sample.to(torch.float32).clone()
        c = self.min_val.shape[0]
        out[..., :c] = (out[..., :c] - self.min_val) / self.range
        return out

    def denormalize(self, batch: torch.Tensor) -> torch.Tensor:
        return batch * self.range.to(batch.device) + self.min_val.to(batch.device)


class NormalizeZScore:
    """(x - mean) / std per feature (features with std == 0 are only centred)."""

    def __init__(self, params):
        self.mean = torch.as_tensor(np.asarray(params["mean"]), dtype=torch.float32)
        self.std = torch.as_tensor(np.asarray(params["std"]), dtype=torch.float32)

    def __call__(self, sample: torch.Tensor) -> torch.Tensor:
        out = sample.to(torch.float32).clone()
        c = self.mean.shape[0]
        div = torch.where(self.std > 0, self.std, torch.ones_like(self.std))
        out[..., :c] = (out[..., :c] - self.mean) / div
        return out

    def denormalize(self, batch: torch.Tensor) -> torch.Tensor:
        return batch * self.std.to(batch.device) + self.mean.to(batch.device)


def create_transforms(norm: Optional[str], data_norm_params) -> "OrderedDict[str, object]":
    """``norm`` in {None, "zscore", "zero_to_one"} -> {"normalize": transform} like the reference's create_transforms."""
    out: "OrderedDict[str, object]" = OrderedDict()
    if norm is None:
        return out
    if norm == "zscore":
        out["normalize"] = NormalizeZScore(data_norm_params)
    elif norm == "zero_to_one":
        out["normalize"] = NormalizeToUnitSegment(data_norm_params)
    else:
        raise ValueError("Choose valid normalization function: zscore or zero_to_one")
    return out


# ---- splits --------------------------------------------------------------------------------------------------------------------
def kfold_ids(n: int, folds: int, split: int, seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """(train_ids, val_ids): ``np.random.seed(seed)``; permutation; ``array_split`` into ``folds``; fold ``split`` (1-based), sorted."""
    np.random.seed(seed)
    chunks = np.array_split(np.random.permutation(n), folds)
    val = np.sort(chunks[split - 1])
    return np.setdiff1d(np.arange(n, dtype=int), val), val


def holdout_ids(devices: np.ndarray, holdout_device_id: int) -> Tuple[np.ndarray, np.ndarray]:
    """(train_ids, val_ids) with every time series of one device held out."""
    devices = np.asarray(devices).astype(int)
    idx = np.arange(len(devices))
    return idx[devices != holdout_device_id], idx[devices == holdout_device_id]


def _torch_load(path):
    try:
        return torch.load(path, weights_only=False)
    except TypeError:   # older torch without the keyword
        return torch.load(path)


# ---- datasets ------------------------------------------------------------------------------------------------------------------
class CVSDataset(torch.utils.data.Dataset):
    def __init__(self, data_dir: str, ds_type: str, seq_len: int, random_start: bool, transforms=None):
        self.transforms = transforms if transforms is not None else {}
        self.random_start, self.ds_type, self.seq_len = random_start, ds_type, seq_len
        obs = _torch_load(os.path.join(data_dir, "processed_data.pkl"))
        train_params = _torch_load(os.path.join(data_dir, "train_params_data.pkl"))
        test_params = _torch_load(os.path.join(data_dir, "test_params_data.pkl"))
        buffer = int(round(obs["train"].shape[0] * (1 - 0.1)))
        if ds_type == "train":
            sl, src, par = slice(None, buffer), obs["train"], train_params
        elif ds_type == "val":
            sl, src, par = slice(buffer, None), obs["train"], train_params
        elif ds_type == "test":
            sl, src, par = slice(None), obs["test"], test_params
        else:
            raise ValueError("ds_type must be train | val | test")
        self.obs = torch.as_tensor(np.asarray(src), dtype=torch.float32)[sl]
        self.iext = torch.as_tensor(np.asarray(par["i_ext"]), dtype=torch.float32)[sl]
        self.rtpr = torch.as_tensor(np.asarray(par["r_tpr_mod"]), dtype=torch.float32)[sl]

    def __len__(self):
        return self.obs.size(0)

    def __getitem__(self, idx):
        start = random.randint(0, self.obs.size(1) - self.seq_len) if self.random_start else 0
        obs = self.obs[idx, start:start + self.seq_len]
        for t in self.transforms.values():
            obs = t(obs)
        return {"observations": obs, "iext": (self.iext[idx] >= 0).float().reshape(1), "rtpr": (self.rtpr[idx] > 0).float().reshape(1)}


class ChallengeDataset(torch.utils.data.Dataset):
    def __init__(self, data: Dict[str, np.ndarray], ds_type: str = "train", seq_len: Optional[int] = None, random_start: bool = False,
                 transforms=None):
        self.transforms = transforms if transforms is not None else {}
        self.ds_type, self.seq_len, self.random_start = ds_type, seq_len, random_start
        self.obs = torch.as_tensor(np.asarray(data["observations"]), dtype=torch.float32)
        self.shedding = torch.as_tensor(np.asarray(data["shedding"]), dtype=torch.float32)
        self.symptom = torch.as_tensor(np.asarray(data["symptoms"]), dtype=torch.float32)

    def __len__(self):
        return self.obs.size(0)

    def __getitem__(self, idx):
        obs = self.obs[idx]
        for t in self.transforms.values():
            obs = t(obs)
        return {"observations": obs, "shedding": self.shedding[idx].float(), "symptoms": self.symptom[idx].float()}


class DatasetPair:
    """train / test dicts + sizes + norm params, like the reference's TimeSeriesDatasetPair (challenge flavour)."""

    def __init__(self, dataset: Dict[str, np.ndarray], train_ids, test_ids, max_time, keys=("observations", "shedding", "symptoms")):
        self.train = {k: dataset[k][train_ids] for k in keys}
        self.test = {k: dataset[k][test_ids] for k in keys}
        self.n_train, self.n_test, self.max_time = len(train_ids), len(test_ids), max_time
        self.data_norm_params = find_norm_params(self.train["observations"])


def build_challenge_datasets(pkl_path: str, seed: int, folds: int, split: int) -> DatasetPair:
    with open(pkl_path, "rb") as fh:
        dataset = pickle.load(fh)
    train_ids, val_ids = kfold_ids(dataset["observations"].shape[0], folds, split, seed)
    return DatasetPair(dataset, train_ids, val_ids, dataset["n_time"])


# ---- proc (plate-reader CSV) -----------------------------------------------------------------------------------------------------
def _parse_conditions(cells: Sequence[str]) -> Tuple[List[str], np.ndarray]:
    """Condition cells ('C6=0.5', 'C6=0.5;C12=1', '' ...) -> (names in first-seen order, values [N, len(names)], absent = 0)."""
    names: List[str] = []
    parsed = []
    for cell in cells:
        pairs = []
        if "=" in cell:
            for item in cell.split(";"):
                key, val = item.split("=")
                if key not in names:
                    names.append(key)
                pairs.append((key, float(val)))
        parsed.append(pairs)
    vals = np.zeros((len(parsed), len(names)))
    for i, pairs in enumerate(parsed):
        for key, val in pairs:
            vals[i, names.index(key)] = val
    return names, vals


def load_proc_csv(path: str, devices: Sequence[str], device_map: Dict[str, float], conditions: Sequence[str], signals: Sequence[str],
                  dtype=np.float32, time_signal: str = "OD"):
    """One plate-reader CSV -> (device ids [N], treatments [N, len(conditions)], times [T], observations [N, len(signals), T]).
    File layout (data/proc/load_proc_data.py:69-83): columns 0..4 = device, colony, well column, well row, condition ('C6=<float>',
    several joined by ';'); then one column per (signal, reading) whose header carries the signal name in parentheses; the first data
    row holds the time of every reading column.  Kept rows: device in ``devices`` and zero for every condition outside ``conditions``
    (the reference derives that list of "other" conditions from the first kept row: load_proc_data.py:18-26; same here).  ``times``
    are those of the ``time_signal`` block.  Returns None when the file has no row of the requested devices.
    The numeric block is parsed once into one float array and split by boolean column masks."""
    import csv
    with open(path, newline="") as fh:
        table = list(csv.reader(fh))
    header, time_row, body = table[0], table[1], [r for r in table[2:] if r]
    body = [r for r in body if r[0] in set(devices)]
    if not body:
        return None
    col_signal = np.array([h[h.find("(") + 1:h.find(")")] if "(" in h and ")" in h[h.find("("):] else h for h in header[5:]])
    dev = np.array([device_map[r[0]] for r in body], dtype=int)
    names, vals = _parse_conditions([r[4] for r in body])
    first_row_names = [n for n, v in zip(names, vals[0]) if ("%s=" % n) in body[0][4]]
    others = [names.index(n) for n in first_row_names if n not in conditions]
    keep = np.all(vals[:, others] == 0.0, axis=1) if others else np.ones(len(body), dtype=bool)
    treatments = np.stack([vals[:, names.index(c)] if c in names else np.zeros(len(body)) for c in conditions], axis=1)[keep]
    block = np.array([r[5:] for r, k in zip(body, keep) if k], dtype=np.float64)
    obs = np.stack([block[:, col_signal == sig] for sig in signals], axis=1)
    times = np.array(time_row[5:], dtype=np.float64)[col_signal == time_signal]
    return dev[keep], treatments.astype(dtype), times.astype(dtype), obs.astype(dtype)


def merge_observations(times_list, observations_list):
    """Several files -> one array on a common time grid: the grid of the file with the fewest series; every other file contributes, for
    each grid time, its reading nearest in time (first one on ties)  (utils/proc_dataset.py:11-26)."""
    grid = np.asarray(times_list[int(np.argmin([len(o) for o in observations_list]))])
    picked = []
    for t, obs in zip(times_list, observations_list):
        nearest = np.abs(np.asarray(t)[None, :] - grid[:, None]).argmin(axis=1)
        picked.append(obs[:, :, nearest])
    return grid, np.concatenate(picked, axis=0)


def scale_data(X: np.ndarray, normalize=None, subtract_background: bool = False):
    """Every signal divided by its maximum over the whole data set (or by ``normalize[i]``), then -- optionally -- every series shifted
    so that its smallest value is 0  (utils/proc_dataset.py:37-50).  Returns (scaled copy, scales)."""
    X = np.array(X, copy=True)
    scales = [m for m in X.max(axis=(0, 2)).astype(np.float32)] if normalize is None else list(normalize)
    X /= np.asarray(scales, dtype=X.dtype)[None, :, None]
    if subtract_background:
        X -= X.min(axis=2, keepdims=True)
    return X, scales


class ProcDataset(torch.utils.data.Dataset):
    """Merged plate-reader files (utils/proc_dataset.py:76-140): ``inputs = log(1 + treatments)``, ``scale_data`` on the
    observations ``[N, signals, T]``, one-hot device cassettes supplied by the caller (``dev_1hot_fn(device_ids)``)."""

    def __init__(self, parsed: Sequence[tuple], normalize=None, subtract_background: bool = False, dev_1hot_fn=None):
        parsed = [p for p in parsed if p is not None]
        times, obs = merge_observations([p[2] for p in parsed], [p[3] for p in parsed])
        self.devices = np.concatenate([p[0] for p in parsed])
        self.inputs = torch.as_tensor(np.log(1.0 + np.concatenate([p[1] for p in parsed])))
        self.times = torch.as_tensor(times)
        obs, self.scales = scale_data(obs, normalize, subtract_background)
        self.observations = torch.as_tensor(obs)
        self.dev_1hot = torch.as_tensor(dev_1hot_fn(self.devices)) if dev_1hot_fn is not None else None

    def __len__(self):
        return len(self.devices)

    def __getitem__(self, idx):
        out = {"devices": self.devices[idx], "inputs": self.inputs[idx], "observations": self.observations[idx]}
        if self.dev_1hot is not None:
            out["dev_1hot"] = self.dev_1hot[idx]
        return out


PROC_DEVICES = ["Pcat_Y81C76", "RS100S32_Y81C76", "RS100S34_Y81C76", "R33S32_Y81C76", "R33S34_Y81C76", "R33S175_Y81C76"]
PROC_GROUPS = OrderedDict([("aR", [0, 1, 1, 2, 2, 2]), ("aS", [0, 1, 2, 1, 2, 3])])   # LuxR / LasR RBS group of every device
PROC_FILES = ["proc140916.csv", "proc140930.csv", "proc141006.csv", "proc141021.csv", "proc141023.csv", "proc141028.csv"]
PROC_SIGNALS = ["OD", "mRFP1", "EYFP", "ECFP"]
PROC_CONDITIONS = ["C6", "C12"]


def proc_cassettes(device_ids: np.ndarray, groups=PROC_GROUPS) -> np.ndarray:
    """Multi-hot device description (data/proc/config_proc.py:70-110, utils/proc_dataset.py:49-73): for every component group the
    one-hot of the device's group index, stacked -- aR (3) then aS (4) for the shipped device list."""
    rows = []
    for d in np.asarray(device_ids).astype(int):
        parts = []
        for g in groups.values():
            v = np.zeros(len(set(g)), dtype=np.float32)
            v[g[d]] = 1.0
            parts.append(v)
        rows.append(np.hstack(parts))
    return np.array(rows, dtype=np.float32)


class ProcTrainingView(torch.utils.data.Dataset):
    """A subset of a ProcDataset in the form ``training_proc.py:23-32`` consumes: ``aR = dev_1hot[:, :3]``, ``aS = dev_1hot[:, 3:]``,
    ``C12 = inputs[:, 0]``, ``C6 = inputs[:, 1]`` (the reference's column assignment, kept as is), observations ``[C, T]``."""

    def __init__(self, ds: ProcDataset, ids):
        self.ds, self.ids = ds, np.asarray(ids)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        j = int(self.ids[i])
        one = self.ds.dev_1hot[j].float()
        inp = self.ds.inputs[j].float()
        return {"observations": self.ds.observations[j].float(), "aR": one[:3], "aS": one[3:], "C12": inp[0:1], "C6": inp[1:2]}


def build_proc_datasets(data_dir: str, seed: int, folds: int, split: int, heldout: Optional[str] = None, files=None):
    """(train view, validation view, times): utils/proc_dataset.py:164-204 on the plate-reader CSVs found in ``data_dir``."""
    files = [f for f in (files or PROC_FILES) if os.path.exists(os.path.join(data_dir, f))]
    if not files:
        raise FileNotFoundError("no plate-reader CSV (%s ...) in %s" % (PROC_FILES[0], data_dir))
    dmap = {d: float(i) for i, d in enumerate(PROC_DEVICES)}
    parsed = [load_proc_csv(os.path.join(data_dir, f), PROC_DEVICES, dmap, PROC_CONDITIONS, PROC_SIGNALS) for f in files]
    ds = ProcDataset(parsed, normalize=None, subtract_background=True, dev_1hot_fn=proc_cassettes)
    np.random.seed(seed)
    if heldout:
        tr, va = holdout_ids(ds.devices, int(dmap[heldout]))
    else:
        tr, va = kfold_ids(len(ds), folds, split, seed)
    return ProcTrainingView(ds, tr), ProcTrainingView(ds, va), ds.times.float()


# ---- host -> device feeding ------------------------------------------------------------------------------------------------------
class BatchFeeder:
    """Iterates ``dataset`` in minibatches of ``batch_size``: samples are collated straight into one of two PINNED host buffers per
    key and copied to ``device`` with ``non_blocking=True`` on a side stream; the consumer's stream waits on the copy's event, so the
    copy of batch k+1 overlaps whatever the consumer runs on batch k.  ``observations`` keep the ``[B, T, C]`` contiguous layout the
    ELBO step takes natively (the models permute the VIEW, training_cvs.py:18-27).  Without a HIP device (CPU tests) the buffers are
    ordinary tensors and the "copy" is the identity."""

    def __init__(self, dataset, batch_size: int, device: Optional[torch.device] = None, shuffle: bool = False, seed: int = 0,
                 drop_last: bool = False, keys: Optional[Sequence[str]] = None):
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.gen = torch.Generator().manual_seed(seed)
        first = dataset[0]
        self.keys = list(keys) if keys is not None else [k for k, v in first.items() if torch.is_tensor(v)]
        self.on_gpu = self.device.type == "cuda"
        self._bufs: List[Dict[str, torch.Tensor]] = []
        for _ in range(2):
            d = {}
            for k in self.keys:
                v = first[k]
                t = torch.empty((self.bs,) + tuple(v.shape), dtype=v.dtype)
                d[k] = t.pin_memory() if self.on_gpu else t
            self._bufs.append(d)
        self._stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._free_evt = [None, None]   # consumer-side events: buffer may be overwritten once the previous copy out of it finished

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = len(self.ds)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        slot = 0
        for b0 in range(0, n, self.bs):
            ids = order[b0:b0 + self.bs]
            if len(ids) < self.bs and self.drop_last:
                break
            buf = self._bufs[slot]
            if self._free_evt[slot] is not None:
                self._free_evt[slot].synchronize()
            for r, i in enumerate(ids):
                s = self.ds[i]
                for k in self.keys:
                    buf[k][r].copy_(s[k])
            if self.on_gpu:
                with torch.cuda.stream(self._stream):
                    out = {k: buf[k][:len(ids)].to(self.device, non_blocking=True) for k in self.keys}
                    evt = torch.cuda.Event()
                    evt.record(self._stream)
                torch.cuda.current_stream(self.device).wait_event(evt)
                for v in out.values():
                    v.record_stream(torch.cuda.current_stream(self.device))
                self._free_evt[slot] = evt
            else:
                out = {k: buf[k][:len(ids)].clone() for k in self.keys}
            slot ^= 1
            yield out
