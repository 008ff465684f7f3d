"""``models.decoders`` of the reference (models/decoders.py:8-141) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.decoders import (  # noqa: F401
    Decoder,
    GaussianDecoder,
    VarianceGaussianDecoder,
)

__all__ = ['Decoder', 'GaussianDecoder', 'VarianceGaussianDecoder']
