"""Host-side mirror of the reference's ``models`` package (same module, class and method names).  Parameters are
``nn.Parameter`` views into ONE flat device vector owned by a :class:`_binding.Binding`; every forward/backward on the hot
path is a call into libslode.so through :mod:`structured_latent_odes_amd.engine`."""
