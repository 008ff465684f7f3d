"""``utils.utils`` of the reference (utils/utils.py:6-35): ``set_seed``, ``find_norm_params``."""
from structured_latent_odes_amd.data import find_norm_params  # noqa: F401
from structured_latent_odes_amd.utils.utils import set_seed as _set_seed

__all__ = ["set_seed", "find_norm_params"]


def set_seed(seed, fully_deterministic=True):
    """utils/utils.py:6-13.  ``fully_deterministic`` only toggles a cuDNN flag in the reference; the slode kernels are bitwise
    reproducible by construction (fixed-order reductions, no float atomics), so it is accepted and has nothing to switch."""
    _set_seed(seed)
