"""``models.encoder_mlp`` of the reference (models/encoder_mlp.py:9-167) resolved to the slode engine (libslode.so, HIP for gfx950)."""
from structured_latent_odes_amd.models.encoder_mlp import (  # noqa: F401
    ConcatModule,
    EncoderMLP,
    ListOutModule,
    call_nn_op,
)

__all__ = ['ConcatModule', 'EncoderMLP', 'ListOutModule', 'call_nn_op']
