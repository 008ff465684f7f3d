"""Print the ODE kernel's LDS bytes / threads for the reference configurations (runs without a GPU: host-side helpers of libslode)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structured_latent_odes_amd import _lib as L, engine as E

lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "structured_latent_odes_amd", "libslode.so"))
f_lds = getattr(lib, "_Z19slode_ode_lds_bytesRK11slode_shapeib"); f_lds.restype = C.c_size_t
f_thr = getattr(lib, "_Z17slode_ode_threadsRK11slode_shape"); f_thr.restype = C.c_int


class _E(E.Engine):
    def __init__(self, spec, T):
        self.spec, self.T, self._shapes = spec, T, {}


for name, spec, T in (("cvs C1 rk4", E.cvs_spec(3, 3, 2, solver="rk4"), 200), ("cvs ref midpoint", E.cvs_spec(3, 3, 2, solver="midpoint"), 86),
                      ("challenge C4", E.challenge_spec(gauss=True, solver="rk4"), 300), ("proc C2", E.proc_spec(z_g=10, z_eps=10, solver="rk4"), 100)):
    s = _E(spec, T).shape(1024)
    nt = f_thr(C.byref(s))
    b = f_lds(C.byref(s), nt, C.c_bool(True))   # loop-free form (one workgroup per trajectory)
    print("%-18s threads %4d  LDS %6d B  -> %d workgroups/CU by LDS" % (name, nt, b, (160 * 1024) // b))
