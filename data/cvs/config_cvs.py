"""``data.cvs.config_cvs.load_config`` (data/cvs/config_cvs.py:6-52): the CVS hyper-parameters as an attribute dictionary (the
reference returns a ``munch``; attribute and item access behave the same)."""
import os

from structured_latent_odes_amd.configs import load_config_cvs

__all__ = ["load_config"]


def load_config():
    args = load_config_cvs()
    args.data_path = os.getcwd() + "/data/cvs/"
    return args
