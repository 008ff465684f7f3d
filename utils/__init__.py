"""Import-level drop-in for the reference's top-level ``utils`` package (``from utils.utils import set_seed``, training_cvs.py:10):
re-binds the host-side helpers of ``structured_latent_odes_amd`` under the reference's module and function names."""
