"""Import-level drop-in for the reference's top-level ``models`` package (``from models.mechanistic_cvs import MechanisticModel``,
training_cvs.py:14-15): every module here re-binds the engine-backed class of the same name from ``structured_latent_odes_amd.models``.
No arithmetic lives in this package."""
