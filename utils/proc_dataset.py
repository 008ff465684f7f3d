"""``utils.proc_dataset`` of the reference (utils/proc_dataset.py:7-204) on ``structured_latent_odes_amd.data``: the plate-reader
data set of the proc family and its train / validation split, under the reference's names."""
import numpy as np
import torch

from structured_latent_odes_amd import data as _D
from structured_latent_odes_amd.data import merge_observations  # noqa: F401

__all__ = ["depth", "merge_observations", "scale_data", "get_cassettes", "TimeSeriesDatasetPair", "build_datasets"]


def depth(group_values):
    """Number of distinct group indices (utils/proc_dataset.py:7-8)."""
    return len(set(group_values))


def scale_data(X, args):
    """utils/proc_dataset.py:37-50 with the reference's ``args.data.{normalize, subtract_background}``."""
    return _D.scale_data(X, args.data.normalize, args.data.subtract_background)


def get_cassettes(devices, args):
    """Multi-hot device description from ``args.data.groups`` (utils/proc_dataset.py:53-73)."""
    return _D.proc_cassettes(np.asarray(devices), args.data.groups)


class TimeSeriesDatasetPair(object):
    """utils/proc_dataset.py:130-156: ``.train`` / ``.test`` datasets, their sizes, ``n_species``, ``n_time``, ``depth``,
    ``n_conditions`` and the (non-uniform) ``times`` grid of the merged files."""

    def __init__(self, train_dataset, test_dataset, args, times):
        self.train, self.test = train_dataset, test_dataset
        self.n_train, self.n_test = len(train_dataset), len(test_dataset)
        _, self.n_species, self.n_time = train_dataset.ds.observations.shape
        self.depth = sum(depth(g) for g in args.data.groups.values())
        self.n_conditions = len(args.data.conditions)
        self.times = times


def build_datasets(config):
    """utils/proc_dataset.py:173-204: merged CSVs under ``config.data_path``; a held-out device if ``config.heldout`` else fold
    ``config.split`` of ``config.folds`` (seeded permutation).  Samples carry ``dev_1hot`` / ``inputs`` like the reference's
    (training_proc.py:25-33 slices them) besides the ready-made ``aR, aS, C12, C6``."""
    tr, va, times = _D.build_proc_datasets(config.data_path, config.seed, config.folds if not config.heldout else 1,
                                           config.split if not config.heldout else 1, config.heldout, files=list(config.data.files))
    return TimeSeriesDatasetPair(_RefView(tr), _RefView(va), config, times.numpy())


class _RefView(torch.utils.data.Dataset):
    """A ``ProcTrainingView`` whose samples also hold the reference's raw keys (utils/proc_dataset.py:118-127)."""

    def __init__(self, view):
        self.view, self.ds, self.ids = view, view.ds, view.ids

    def __len__(self):
        return len(self.view)

    def __getitem__(self, i):
        out = self.view[i]
        j = int(self.ids[i])
        out.update(dev_1hot=self.ds.dev_1hot[j].float(), inputs=self.ds.inputs[j].float(), devices=self.ds.devices[j])
        return out
