"""ctypes binding of libslode.so (include/slode.h).  The product path has NO fallback: if the shared library is
missing or no gfx950 device is visible, importing the engine / creating a handle raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLODE_LIB_PATH") or os.path.join(_HERE, "libslode.so")  # env override: diagnostics only

MAX_GROUPS, MAX_HEADS, MAX_AUX, MAX_LABELS = 4, 3, 4, 4
AUX_KINDS = {"sigmoid": 0, "softmax": 1, "expexp": 2}
EULER, MIDPOINT, RK4, DOPRI5 = 0, 1, 2, 3
ALD, GAUSS = 0, 1
METHODS = {"euler": EULER, "midpoint": MIDPOINT, "rk4": RK4, "dopri5": DOPRI5}
GRAD_MODES = {"exact": 0, "reference_adjoint": 1}   # slode_grad_mode


class Group(C.Structure):
    _fields_ = [("z_off", C.c_int32), ("z_dim", C.c_int32), ("u_off", C.c_int32), ("u_dim", C.c_int32)]


class Aux(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "z_off", "z_dim", "u_off", "u_dim")]


class Shape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "C", "L", "S", "H", "F", "K", "P", "Hc", "n_u", "n_groups")] + [
        ("groups", Group * MAX_GROUPS), ("method", C.c_int32), ("likelihood", C.c_int32),
        ("quantile_diff", C.c_float), ("rtol", C.c_float), ("atol", C.c_float), ("n_aux", C.c_int32), ("U", C.c_int32),
        ("aux_mult", C.c_float), ("aux", Aux * MAX_AUX), ("aux_in_main", C.c_int32), ("grad_mode", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("conv_w", "conv_b", "lin_w", "lin_b", "zloc_w", "zloc_b", "zls_w", "zls_b",
                                          "ode_begin")] + [
        ("ploc_w", C.c_int32 * MAX_GROUPS), ("ploc_b", C.c_int32 * MAX_GROUPS),
        ("pls_w", C.c_int32 * MAX_GROUPS), ("pls_b", C.c_int32 * MAX_GROUPS)] + [
        (n, C.c_int32) for n in ("init_w1", "init_b1", "init_w2", "init_b2", "dyn_wh", "dyn_bh", "dyn_wg", "dyn_bg",
                                 "dyn_wd", "dyn_bd")] + [
        ("head_w", C.c_int32 * MAX_HEADS)] + [(n, C.c_int32 * MAX_AUX) for n in ("aux_w1", "aux_b1", "aux_w2", "aux_b2", "aux_w3", "aux_b3", "aux_c")] + [
        ("cstd", C.c_int32), ("ode_end", C.c_int32), ("n_params", C.c_int32)]


class Batch(C.Structure):
    """slode_batch: one minibatch as the loader yields it -- observations with strides, the label tensors one by one, optional eps."""
    _fields_ = [("obs", C.c_void_p), ("obs_strides", C.c_int64 * 3), ("n_labels", C.c_int32), ("label_width", C.c_int32 * MAX_LABELS),
                ("labels", C.c_void_p * MAX_LABELS), ("eps", C.c_void_p)]


class AdamArgs(C.Structure):
    """slode_adam"""
    _fields_ = [("n_total", C.c_int64), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int64)]


SVI_MAIN, SVI_AUX = 0, 1

EXPORTS = ["slode_version", "slode_create", "slode_destroy", "slode_last_error", "slode_layout_init",
           "slode_num_stage_times", "slode_workspace_bytes", "slode_stage_times", "slode_encoder_conv_fwd",
           "slode_encoder_conv_bwd", "slode_ode_solve_fwd", "slode_ode_solve_bwd", "slode_decode_heads",
           "slode_elbo_step", "slode_adam_step", "slode_profile_enable", "slode_profile_read", "slode_dynamics_eval", "slode_elbo_adam_step", "slode_aux_step", "slode_adam_region",
           "slode_initialize_state", "slode_prior_nets", "slode_label_heads", "slode_dopri5_step_counts", "slode_decode_heads_bwd",
           "slode_svi_step", "slode_rng_seed", "slode_rng_set_counter", "slode_rng_get", "slode_rng_normal", "slode_sample_normal",
           "slode_grad_payload_floats", "slode_grad_partial", "slode_grad_apply", "slode_fold_invalidate"]

_lib = None


class SlodeError(RuntimeError):
    pass


def load():
    """Load libslode.so (built by ``__graft_entry__.build()`` / ``make -C structured_latent_odes_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlodeError("libslode.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                         "there is no CPU/PyTorch fallback for the hot path" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch BEFORE dlopen so
    # libslode binds to the runtime torch already loaded; loaded the other way round, two runtimes coexist and slode_create reports
    # "no ROCm-capable device" (seen with `python __graft_entry__.py smoke`, where build() loaded the library first).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    P, VP, I64P = C.POINTER, C.c_void_p, C.POINTER(C.c_int64)
    lib.slode_version.restype = C.c_int
    lib.slode_adam_region.argtypes = [VP, C.c_int64, C.c_int64, C.c_int64]
    lib.slode_create.argtypes = [P(VP), C.c_int]
    lib.slode_destroy.argtypes = [VP]
    lib.slode_last_error.argtypes = [VP]
    lib.slode_last_error.restype = C.c_char_p
    lib.slode_layout_init.argtypes = [P(Shape), P(Layout)]
    lib.slode_num_stage_times.argtypes = [P(Shape)]
    lib.slode_workspace_bytes.argtypes = [VP, P(Shape)]
    lib.slode_workspace_bytes.restype = C.c_size_t
    lib.slode_stage_times.argtypes = [VP, P(Shape), VP, VP, VP]
    lib.slode_encoder_conv_fwd.argtypes = [VP, P(Shape), P(Layout), VP, VP, I64P, VP, VP, VP, VP, VP]
    lib.slode_encoder_conv_bwd.argtypes = [VP, P(Shape), P(Layout), VP, VP, I64P, VP, VP, VP, VP, VP, VP, VP, C.c_size_t, VP]
    lib.slode_ode_solve_fwd.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, VP, VP]
    lib.slode_ode_solve_bwd.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, VP, VP, VP, VP, C.c_size_t, VP]
    lib.slode_decode_heads.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, VP]
    lib.slode_decode_heads_bwd.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, VP, VP, VP, VP]
    lib.slode_elbo_step.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, I64P, VP, VP, VP, VP, VP, VP, VP, C.c_size_t, VP]
    lib.slode_adam_step.argtypes = [VP, C.c_int64, VP, VP, VP, VP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, VP]
    lib.slode_elbo_adam_step.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, I64P, VP, VP, VP, VP, VP, C.c_size_t, C.c_int64, VP, VP,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, VP]
    lib.slode_aux_step.argtypes = [VP, P(Shape), P(Layout), VP, VP, I64P, VP, VP, VP, VP, VP, C.c_size_t, C.c_int64, VP, VP,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, VP]
    lib.slode_dynamics_eval.argtypes = [VP, P(Shape), P(Layout), VP, C.c_float, VP, VP, VP, VP]
    lib.slode_initialize_state.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP]
    lib.slode_prior_nets.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP, VP]
    lib.slode_label_heads.argtypes = [VP, P(Shape), P(Layout), VP, VP, VP, VP]
    lib.slode_dopri5_step_counts.argtypes = [VP, P(Shape), P(Layout), VP, C.c_size_t, VP, VP]
    lib.slode_profile_enable.argtypes = [VP, C.c_int]
    lib.slode_profile_read.argtypes = [VP, C.c_int, P(C.c_char_p), P(C.c_float)]
    lib.slode_svi_step.argtypes = [VP, P(Shape), P(Layout), C.c_int, VP, VP, VP, P(Batch), VP, VP, VP, C.c_size_t, P(AdamArgs), VP]
    lib.slode_rng_seed.argtypes = [VP, C.c_uint64, C.c_int64]
    lib.slode_rng_set_counter.argtypes = [VP, C.c_uint64]
    lib.slode_rng_get.argtypes = [VP, P(C.c_uint64), P(C.c_int64), P(C.c_uint64)]
    lib.slode_rng_normal.argtypes = [VP, C.c_uint64, C.c_int32, C.c_int32, VP, VP, VP]
    lib.slode_sample_normal.argtypes = [VP, C.c_int32, C.c_int32, VP, VP, VP, VP]
    lib.slode_fold_invalidate.argtypes = [VP]
    lib.slode_grad_payload_floats.argtypes = [P(Shape), P(Layout), C.c_int]
    lib.slode_grad_payload_floats.restype = C.c_size_t
    lib.slode_grad_partial.argtypes = [VP, P(Shape), P(Layout), C.c_int, VP, VP, VP, P(Batch), VP, VP, C.c_size_t, VP]
    lib.slode_grad_apply.argtypes = [VP, P(Shape), P(Layout), C.c_int, VP, I64P, VP, VP, VP, VP, C.c_size_t, P(AdamArgs), VP]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here == the ABI in include/slode.h is not fully exported
    _lib = lib
    return lib
