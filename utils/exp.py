"""``utils.exp`` of the reference (utils/exp.py:5-14): the ``Exp`` module."""
from structured_latent_odes_amd.utils.exp import Exp  # noqa: F401

__all__ = ["Exp"]
