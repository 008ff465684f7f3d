"""``models.encoder_conv`` of the reference (models/encoder_conv.py:6-51) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.encoder_conv import (  # noqa: F401
    EncoderCONV,
    Exp,
)

__all__ = ['EncoderCONV', 'Exp']
