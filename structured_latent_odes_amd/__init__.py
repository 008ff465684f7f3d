"""slode -- MI355X-native engine for the latent-ODE solve + ELBO path of paidamoyo/structured_latent_ODEs.

The hot path lives in ``libslode.so`` (hand-written HIP for gfx950, C ABI in ``include/slode.h``); this package is the
Python host side mirroring the reference's module API (``models.*``, ``training_*``).  No CPU fallback exists."""
__version__ = "0.1.0"
