"""``models.mechanistic_challenge`` of the reference (models/mechanistic_challenge.py) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.mechanistic_challenge import (  # noqa: F401
    MechanisticModel,
)

__all__ = ['MechanisticModel']
