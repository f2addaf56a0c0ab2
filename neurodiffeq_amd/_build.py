"""In-tree build of the gfx950 artefacts (no cmake, no torch headers): hipcc through ``_hipcc.compile_shared`` (device
assembly fix-up pass included).

``libndq.so`` = csrc/ndq_api.hip (+ ndq_mlp.h, ndq_launch.h, ndq_sample.h), the C-ABI declared in include/ndq.h.  The built library travels to
the GPU box with the repository snapshot; ``ensure_built`` recompiles only when a source is newer than the library.
"""
import os

from . import _hipcc

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libndq.so")
SOURCES = [os.path.join(CSRC, "ndq_api.hip")]
HEADERS = [os.path.join(CSRC, h) for h in ("ndq_mlp.h", "ndq_tail.h", "ndq_launch.h", "ndq_wide.h", "ndq_deep.h", "ndq_sample.h", "ndq_oneshot.h")] + \
    [os.path.join(HERE, "..", "include", "ndq.h"), os.path.join(HERE, "_hipcc.py")]
# libndq64.so = csrc/ndq_api64.hip: the same stream kernels compiled for fp64 (ndq64_* entry points of include/ndq.h)
LIB64 = os.path.join(HERE, "libndq64.so")
SOURCES64 = [os.path.join(CSRC, "ndq_api64.hip")]
FLAGS = []
_EXTRA = os.environ.get("NDQ_LIB_FLAGS", "").split()      # tuning experiments: built to a library of their own
if _EXTRA:
    import hashlib
    LIB = os.path.join(HERE, "libndq_" + hashlib.sha1(" ".join(_EXTRA).encode()).hexdigest()[:8] + ".so")
    FLAGS = FLAGS + _EXTRA


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    _hipcc.compile_shared(SOURCES, LIB, FLAGS, verbose=verbose)
    return LIB


def is_stale64():
    if not os.path.exists(LIB64):
        return True
    t = os.path.getmtime(LIB64)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES64 + HEADERS)


def build_lib64(force=False, verbose=False):
    if not force and not is_stale64():
        return LIB64
    _hipcc.compile_shared(SOURCES64, LIB64, [], verbose=verbose)
    return LIB64


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
