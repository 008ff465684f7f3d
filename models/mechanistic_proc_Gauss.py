"""``models.mechanistic_proc_Gauss`` of the reference (models/mechanistic_proc_Gauss.py) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.mechanistic_proc_Gauss import (  # noqa: F401
    MechanisticModelGauss,
)

__all__ = ['MechanisticModelGauss']
