"""``models.blackbox_ode`` of the reference (models/blackbox_ode.py:7-109) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.blackbox_ode import (  # noqa: F401
    Dynamics,
    OdeFunc,
    OdeModel,
)

__all__ = ['Dynamics', 'OdeFunc', 'OdeModel']
