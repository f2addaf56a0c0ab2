"""ctypes binding of libndq.so (C-ABI: include/ndq.h).  The product path has no CPU or torch fallback for these
entry points: if the library is missing or a launch fails, it raises."""
import ctypes
import os

from . import _build

NDQ_ACT_TANH, NDQ_ACT_SIN = 0, 1


class MlpDesc(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int), ("first", ctypes.c_int), ("mask2", ctypes.c_int), ("hidden", ctypes.c_int),
                ("layers", ctypes.c_int), ("act", ctypes.c_int), ("n_out", ctypes.c_int)]

    def key(self):
        return (self.d, self.first, self.mask2, self.hidden, self.layers, self.act, self.n_out)


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
    for name in ("ndq_mlp_supported", "ndq_mlp_num_streams", "ndq_mlp_num_params", "ndq_mlp_bwd_blocks",
                 "ndq_mlp_jet_fwd", "ndq_mlp_jet_bwd", "ndq_reduce_partials", "ndq_adam_step"):
        getattr(L, name).restype = ci
    _LIB = L
    return L


EXPORTS = ("ndq_mlp_supported", "ndq_mlp_num_streams", "ndq_mlp_num_params", "ndq_mlp_bwd_blocks", "ndq_mlp_jet_fwd",
           "ndq_mlp_jet_bwd", "ndq_reduce_partials", "ndq_adam_step")


def check(rc, what):
    if rc != 0:
        raise NdqError(f"{what} failed with code {rc}" + (" (no kernel for this descriptor)" if rc == -1 else ""))
