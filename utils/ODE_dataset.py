"""``utils.ODE_dataset`` of the reference (utils/ODE_dataset.py:6-84,171-233) on ``structured_latent_odes_amd.data``: the CVS and
challenge dataset readers and the normalisation transforms, with the reference's constructor signatures.  ``ODEDataSynBio``
(:87-168) is unused by every entry point of the reference and is not provided."""
import os

from structured_latent_odes_amd import data as _D
from structured_latent_odes_amd.data import NormalizeToUnitSegment, NormalizeZScore  # noqa: F401

__all__ = ["ODEDataCSV", "ODEDataChallenge", "NormalizeZScore", "NormalizeToUnitSegment", "create_transforms"]


class ODEDataCSV(_D.CVSDataset):
    """``ODEDataCSV(data_dir, ds_type, seq_len, random_start, transforms=None)`` (utils/ODE_dataset.py:6-56)."""


class ODEDataChallenge(_D.ChallengeDataset):
    """``ODEDataChallenge(data, ds_type, seq_len, random_start, transforms=None)`` (utils/ODE_dataset.py:59-84)."""


def create_transforms(args, data_norm_params=None):
    """utils/ODE_dataset.py:219-233: ``args.norm`` selects the transform; the statistics default to the pickled
    ``data_norm_params.pkl`` under ``args.data_path``."""
    if data_norm_params is None:
        data_norm_params = _D._torch_load(os.path.join(args.data_path, "data_norm_params.pkl"))
    return _D.create_transforms(args.norm, data_norm_params)
