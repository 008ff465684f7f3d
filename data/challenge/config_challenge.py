"""``data.challenge.config_challenge.load_config`` (data/challenge/config_challenge.py:6-49)."""
import os

from structured_latent_odes_amd.configs import load_config_challenge

__all__ = ["load_config"]


def load_config():
    args = load_config_challenge()
    args.data_path = os.getcwd() + "/data/challenge/"
    return args
