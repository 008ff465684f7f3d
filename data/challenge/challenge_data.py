"""``data.challenge.challenge_data`` (data/challenge/challenge_data.py:9-54): the seeded k-fold split of the challenge pickle."""
import os

from structured_latent_odes_amd.data import DatasetPair as TimeSeriesDatasetPair  # noqa: F401
from structured_latent_odes_amd.data import build_challenge_datasets

__all__ = ["TimeSeriesDatasetPair", "build_datasets"]


def build_datasets(config):
    """:30-54.  The reference opens the literal relative path ``data/challenge/data.pkl``; ``config.data_path`` (set by
    ``load_config`` to that same directory under the working directory) is honoured first so the file can live elsewhere."""
    path = os.path.join(getattr(config, "data_path", "data/challenge/"), "data.pkl")
    if not os.path.exists(path):
        path = "data/challenge/data.pkl"
    return build_challenge_datasets(path, config.seed, config.folds, config.split)
