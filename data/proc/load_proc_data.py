"""``data.proc.load_proc_data.load`` (data/proc/load_proc_data.py:69-125): one plate-reader CSV -> (devices, treatments, times,
observations) restricted to ``args.data.{devices, conditions, signals}``."""
import numpy as np

from structured_latent_odes_amd.data import load_proc_csv

__all__ = ["load"]


def load(csv_file, args):
    d = args.data
    return load_proc_csv(csv_file, d.devices, d.device_map, d.conditions, d.signals, dtype=np.dtype(getattr(d, "dtype", "float32")).type)
