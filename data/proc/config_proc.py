"""``data.proc.config_proc`` (data/proc/config_proc.py:9-130): ``load_config`` and the ``Config`` device / file tables."""
import os
from collections import OrderedDict

import numpy as np

from structured_latent_odes_amd import data as _D
from structured_latent_odes_amd.configs import AttrDict, load_config_proc

__all__ = ["load_config", "Config"]


class Config(object):
    """:69-130: ``.data`` = device list, component groups, per-group relevance vectors and the device look-up maps."""

    def __init__(self):
        d = AttrDict(groups=AttrDict((k, list(v)) for k, v in _D.PROC_GROUPS.items()), devices=list(_D.PROC_DEVICES), normalize=None,
                     subtract_background=True, conditions=list(_D.PROC_CONDITIONS), files=list(_D.PROC_FILES),
                     signals=list(_D.PROC_SIGNALS), default_devices=dict(), dtype="float32")
        sizes = [len(set(g)) for g in d.groups.values()]
        d.component_maps = OrderedDict((k, OrderedDict(zip(d.devices, g))) for k, g in d.groups.items())
        d.device_depth = sum(sizes)
        d.relevance_vectors = OrderedDict()
        lo = 0
        for k, n in zip(d.groups, sizes):
            rv = np.zeros(d.device_depth, dtype=np.float32)
            rv[lo:lo + n] = 1.0
            if k in d.default_devices:
                rv[lo + d.default_devices[k]] = 0.0
            d.relevance_vectors[k] = rv
            lo += n
        d.device_map = {name: float(i) for i, name in enumerate(d.devices)}
        d.device_idx_to_device_name = dict(enumerate(d.devices))
        d.device_lookup = {v: k for k, v in d.device_map.items()}
        self.data = d


def load_config():
    args = load_config_proc()
    args.data_path = "data/proc/"
    args.output_dir = os.getcwd() + "/"
    args.data = Config().data
    return args
