"""``models.mechanistic_proc`` of the reference (models/mechanistic_proc.py) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.mechanistic_proc import (  # noqa: F401
    MechanisticModel,
)

__all__ = ['MechanisticModel']
