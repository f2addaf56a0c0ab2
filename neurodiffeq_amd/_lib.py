"""ctypes binding of libndq.so (C-ABI: include/ndq.h).  The product path has no CPU or torch fallback for these
entry points: if the library is missing or a launch fails, it raises."""
import ctypes
import os

from . import _build

NDQ_ACT_TANH, NDQ_ACT_SIN, NDQ_ACT_SIGMOID, NDQ_ACT_SWISH, NDQ_ACT_APTX = 0, 1, 2, 3, 4
NDQ_ACT_ELU, NDQ_ACT_SOFTPLUS, NDQ_ACT_GELU = 5, 6, 7


class MlpDesc(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int), ("first", ctypes.c_int), ("mask2", ctypes.c_int), ("hidden", ctypes.c_int),
                ("layers", ctypes.c_int), ("act", ctypes.c_int), ("n_out", ctypes.c_int), ("lap", ctypes.c_int),
                ("skip", ctypes.c_int), ("mask3", ctypes.c_int), ("actp", ctypes.c_int), ("widths", ctypes.c_int), ("mono", ctypes.c_int),
                ("mask4", ctypes.c_int)]

    def key(self):
        return (self.d, self.first, self.mask2, self.hidden, self.layers, self.act, self.n_out, self.lap, self.skip,
                self.mask3, self.actp, self.widths, self.mono, self.mask4)


NDQ_SAMPLE_UNIFORM, NDQ_SAMPLE_GRID, NDQ_SAMPLE_SPHERICAL = 0, 1, 2


class SamplerDesc(ctypes.Structure):
    """ndq_sampler_desc of include/ndq.h"""
    _fields_ = [("kind", ctypes.c_int), ("d", ctypes.c_int), ("n", ctypes.c_int * 3), ("lo", ctypes.c_float * 3),
                ("hi", ctypes.c_float * 3), ("noise_std", ctypes.c_float * 3), ("radial", ctypes.c_int)]


FUSED_LAUNCH_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_float, ctypes.c_int, ctypes.c_void_p)


class FusedStep(ctypes.Structure):
    """ndq_fused_step of include/ndq.h"""
    _fields_ = [("launch", ctypes.c_void_p),
                ("n", ctypes.c_int), ("ldc", ctypes.c_int), ("ldj", ctypes.c_int), ("blocks", ctypes.c_int),
                ("n_params", ctypes.c_int), ("seed", ctypes.c_float),
                ("params", ctypes.c_void_p), ("partials", ctypes.c_void_p), ("loss_partials", ctypes.c_void_p),
                ("grad", ctypes.c_void_p), ("loss_slot", ctypes.c_void_p), ("adam_m", ctypes.c_void_p),
                ("adam_v", ctypes.c_void_p),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float),
                ("loss_hist", ctypes.c_void_p), ("best_loss", ctypes.c_void_p), ("best_flat", ctypes.c_void_p),
                ("allreduce", ctypes.c_void_p), ("comm", ctypes.c_void_p),
                ("next_sampler", ctypes.c_void_p), ("next_seed", ctypes.c_ulonglong), ("next_draw", ctypes.c_ulonglong),
                ("next_stream", ctypes.c_uint), ("next_coords", ctypes.c_void_p), ("next_ldc", ctypes.c_int)]


class FusedFit(ctypes.Structure):
    """ndq_fused_fit of include/ndq.h"""
    _fields_ = [("launch", ctypes.c_void_p), ("n_nets", ctypes.c_int), ("net", FusedStep * 4),
                ("valid_coords", ctypes.c_void_p), ("valid_n", ctypes.c_int), ("valid_ldc", ctypes.c_int),
                ("valid_blocks", ctypes.c_int), ("valid_scale", ctypes.c_float),
                ("valid_loss_partials", ctypes.c_void_p), ("valid_hist", ctypes.c_void_p), ("track_best", ctypes.c_int),
                ("pull_ok", ctypes.c_int), ("alt_params", ctypes.c_void_p * 4), ("alt_m", ctypes.c_void_p * 4),
                ("alt_v", ctypes.c_void_p * 4), ("alt_partials", ctypes.c_void_p * 4),
                ("alt_loss_partials", ctypes.c_void_p), ("alt_valid_loss_partials", ctypes.c_void_p),
                ("launch_loop", ctypes.c_void_p), ("loop_ok", ctypes.c_int)]


class NdqError(RuntimeError):
    pass


_LIB = None


def lib():
    """Load libndq.so (building it first if hipcc is available and the sources are newer)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if _build.is_stale():
        try:
            _build.build_lib()
        except Exception as e:  # no hipcc / compile error
            if not os.path.exists(path):
                raise NdqError(f"libndq.so is missing and could not be built: {e}") from e
    L = ctypes.CDLL(path)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    dp = ctypes.POINTER(MlpDesc)
    L.ndq_mlp_supported.argtypes = [dp]
    L.ndq_mlp_num_streams.argtypes = [dp]
    L.ndq_mlp_num_params.argtypes = [dp]
    L.ndq_mlp_bwd_blocks.argtypes = [dp, ci]
    L.ndq_mlp_jet_fwd.argtypes = [dp, vp, ci, ci, vp, vp, ci, vp]
    L.ndq_mlp_jet_bwd.argtypes = [dp, vp, ci, ci, vp, vp, ci, vp, vp]
    L.ndq_reduce_partials.argtypes = [vp, ci, ci, vp, ci, cf, vp]
    L.ndq_adam_step.argtypes = [vp, vp, vp, vp, ci, cf, cf, cf, cf, cf, ci, vp]
    L.ndq_reduce_grad_loss.argtypes = [vp, ci, ci, vp, ci, vp, ci, vp, cf, vp]
    L.ndq_epoch_tail.argtypes = [vp, vp, vp, vp, ci, cf, cf, cf, cf, cf, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp]
    L.ndq_fused_step_run.argtypes = [ctypes.POINTER(FusedStep), vp, ci, ci, ci, vp]
    L.ndq_fused_multi_step_run.argtypes = [ctypes.POINTER(FusedStep), ci, vp, vp, ci, ci, ci, vp]
    L.ndq_fused_fit_run.argtypes = [ctypes.POINTER(FusedFit), ci, vp, ci, ci, ci, ci, vp]
    L.ndq_mlp_register.argtypes = [vp]
    L.ndq_sample.argtypes = [ctypes.POINTER(SamplerDesc), ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, vp, ci, vp]
    L.ndq_oneshot_create.argtypes = [ci, ci, ci, ctypes.POINTER(vp), ctypes.c_char_p]
    L.ndq_oneshot_connect.argtypes = [vp, ctypes.c_char_p]
    L.ndq_oneshot_allreduce.argtypes = [vp, vp, ctypes.c_size_t, ci, ci, vp, vp]
    L.ndq_oneshot_status.argtypes = [vp]
    L.ndq_oneshot_destroy.argtypes = [vp]
    for name in ("ndq_mlp_supported", "ndq_mlp_num_streams", "ndq_mlp_num_params", "ndq_mlp_bwd_blocks",
                 "ndq_mlp_jet_fwd", "ndq_mlp_jet_bwd", "ndq_reduce_partials", "ndq_adam_step", "ndq_reduce_grad_loss",
                 "ndq_epoch_tail", "ndq_fused_step_run", "ndq_sample", "ndq_mlp_register", "ndq_fused_multi_step_run",
                 "ndq_fused_fit_run", "ndq_oneshot_create", "ndq_oneshot_connect", "ndq_oneshot_allreduce", "ndq_oneshot_status",
                 "ndq_oneshot_destroy"):
        getattr(L, name).restype = ci
    _LIB = L
    return L


_LIB64 = None


def lib64():
    """Load libndq64.so (the fp64 build of the stream kernels: ndq64_* of include/ndq.h), building it first if needed."""
    global _LIB64
    if _LIB64 is not None:
        return _LIB64
    if _build.is_stale64():
        try:
            _build.build_lib64()
        except Exception as e:  # no hipcc / compile error
            if not os.path.exists(_build.LIB64):
                raise NdqError(f"libndq64.so is missing and could not be built: {e}") from e
    L = ctypes.CDLL(_build.LIB64)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    dp = ctypes.POINTER(MlpDesc)
    L.ndq64_mlp_supported.argtypes = [dp]
    L.ndq64_mlp_num_streams.argtypes = [dp]
    L.ndq64_mlp_num_params.argtypes = [dp]
    L.ndq64_mlp_bwd_blocks.argtypes = [dp, ci]
    L.ndq64_mlp_jet_fwd.argtypes = [dp, vp, ci, ci, vp, vp, ci, vp]
    L.ndq64_mlp_jet_bwd.argtypes = [dp, vp, ci, ci, vp, vp, ci, vp, vp]
    L.ndq64_reduce_partials.argtypes = [vp, ci, ci, vp, ci, ctypes.c_double, vp]
    L.ndq64_mlp_register.argtypes = [vp]
    cd = ctypes.c_double
    L.ndq64_adam_step.argtypes = [vp, vp, vp, vp, ci, cd, cd, cd, cd, cd, ci, vp]
    L.ndq64_epoch_tail.argtypes = [vp, vp, vp, vp, ci, cd, cd, cd, cd, cd, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp]
    for name in EXPORTS64:
        getattr(L, name).restype = ci
    _LIB64 = L
    return L


EXPORTS64 = ("ndq64_mlp_register", "ndq64_mlp_supported", "ndq64_mlp_num_streams", "ndq64_mlp_num_params",
             "ndq64_mlp_bwd_blocks", "ndq64_mlp_jet_fwd", "ndq64_mlp_jet_bwd", "ndq64_reduce_partials", "ndq64_adam_step",
             "ndq64_epoch_tail")

EXPORTS = ("ndq_mlp_supported", "ndq_mlp_num_streams", "ndq_mlp_num_params", "ndq_mlp_bwd_blocks", "ndq_mlp_jet_fwd",
           "ndq_mlp_jet_bwd", "ndq_reduce_partials", "ndq_adam_step", "ndq_reduce_grad_loss", "ndq_epoch_tail",
           "ndq_fused_step_run", "ndq_sample", "ndq_mlp_register", "ndq_fused_multi_step_run", "ndq_fused_fit_run",
           "ndq_oneshot_create", "ndq_oneshot_connect", "ndq_oneshot_allreduce", "ndq_oneshot_status", "ndq_oneshot_destroy")


def check(rc, what):
    if rc != 0:
        raise NdqError(f"{what} failed with code {rc}" + (" (no kernel for this descriptor)" if rc == -1 else ""))
