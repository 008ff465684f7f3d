"""``models.mechanistic_cvs`` of the reference (models/mechanistic_cvs.py) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.mechanistic_cvs import (  # noqa: F401
    MechanisticModel,
)

__all__ = ['MechanisticModel']
