"""Hazard canary: does THIS machine show the gfx950 hazards the build pipeline works around, and does the work-around hold?

DESIGN.md 4.6: (1) a packed-fp32 VALU instruction with ``op_sel`` directly followed by a bf16 MFMA can deliver a wrong
low half in lanes 48..63; (2) kernels that spill returned corrupted rows with two waves per SIMD until those instructions
were scalarised -- an observation whose mechanism is not known.  ``_hipcc.py`` rewrites the compiler's assembly for both.
Everything the package ships was validated on the boxes it was developed on; this module re-validates on the box the code
actually lands on:

* ``build()``   (``__graft_entry__.build``): the stand-alone reproducer ``csrc/canary_pk_war.hip`` is built twice -- through
  the fix-up pass ("fixed") and with the compiler's unmodified output ("raw").
* ``check()``   (first ``FusedSystem`` of a process, ~20 ms): runs both.  ``fixed`` must count ZERO wrong results; if it does
  not, the work-around does not cover this machine: a loud ``RuntimeWarning``, and the two-waves-per-SIMD builds of the
  closure kernels (the only configuration hazard (2) was ever seen in) are refused for the rest of the process.  ``raw``
  is the machine's signature: > 0 on every MI355X seen so far (profiles/archive/r01/r01u_pk_mfma_hazard.txt: 5.8 M of 52 M); a box where
  it is 0 is logged, nothing more.
* ``check_two_waves()`` (GPU test; ``NDQ_CANARY=full`` runs it at first use too): an adjoint kernel that spills, built with
  two waves per SIMD, launched repeatedly on the same inputs -- through the pass every launch must return the same bits.
"""
import ctypes
import os
import warnings

import torch

from . import _hipcc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "canary_pk_war.hip")
JIT_DIR = os.path.join(HERE, "_jit")
STATUS = {"checked": False, "fixed": None, "raw": None, "refuse_two_waves": False}


def _so(label):
    return os.path.join(JIT_DIR, f"canary_pk_war_{label}.so")


def build(force=False):
    """Both variants of the reproducer, in-tree (they travel to the GPU box with the snapshot).  Returns the paths."""
    os.makedirs(JIT_DIR, exist_ok=True)
    out = []
    for label, off in (("fixed", "0"), ("raw", "1")):
        so = _so(label)
        stale = force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(SRC), os.path.getmtime(_hipcc.__file__))
        if stale:
            old = os.environ.get("NDQ_NO_PK_MFMA_FIX")
            os.environ["NDQ_NO_PK_MFMA_FIX"] = off
            try:
                _hipcc.compile_shared(SRC, so, ["-DNDQ_PK_WAR_LIB=1", "-Wno-unused-value"])
            finally:
                if old is None:
                    os.environ.pop("NDQ_NO_PK_MFMA_FIX", None)
                else:
                    os.environ["NDQ_NO_PK_MFMA_FIX"] = old
        out.append(so)
    return out


def _count(label, iters):
    lib = ctypes.CDLL(_so(label))
    lib.ndq_pk_war_count.restype = ctypes.c_long
    lib.ndq_pk_war_count.argtypes = [ctypes.c_int]
    return int(lib.ndq_pk_war_count(iters))


def check(iters=50, rebuild=True):
    """Run the canary once per process (see the module docstring).  Returns STATUS."""
    if STATUS["checked"] or os.environ.get("NDQ_CANARY", "1") == "0" or not torch.cuda.is_available():
        return STATUS
    STATUS["checked"] = True
    try:
        if rebuild and not (os.path.exists(_so("fixed")) and os.path.exists(_so("raw"))):
            build()
        STATUS["fixed"], STATUS["raw"] = _count("fixed", iters), _count("raw", iters)
    except Exception as e:      # noqa: BLE001 -- no hipcc and no prebuilt canary: say so, do not block training
        STATUS["error"] = f"{type(e).__name__}: {e}"[:300]
        warnings.warn(f"neurodiffeq_amd: the gfx950 hazard canary could not run ({STATUS['error']}); every closure kernel "
                      "still proves itself on first use (engine.verify_fused)", RuntimeWarning)
        return STATUS
    if STATUS["fixed"] != 0:
        STATUS["refuse_two_waves"] = True
        warnings.warn(f"neurodiffeq_amd: the packed-fp32 -> MFMA hazard reproducer returned {STATUS['fixed']} wrong results "
                      "THROUGH the assembly fix-up pass on this machine -- the work-around of DESIGN.md 4.6 does not cover "
                      "it.  Two-waves-per-SIMD closure-kernel builds are refused for this process; every closure kernel is "
                      "still checked against the three-kernel pipeline on first use.", RuntimeWarning)
    if os.environ.get("NDQ_CANARY") == "full":
        check_two_waves()
    return STATUS


def check_two_waves(launches=30, n=20000):
    """An adjoint kernel that spills (full second-order stream set, 2 inputs), built with two waves per SIMD, ``launches``
    times on the same inputs: returns dict(fixed=launches whose partial rows differ from the first, raw=the same with the
    compiler's unmodified output, spills=...).  ``fixed`` must be 0."""
    from . import _lib, codegen
    desc = _lib.MlpDesc(2, 1, 7, 32, 2, 0, 1, 0, 0, 0)

    class Kernels(ctypes.Structure):
        _fields_ = [("desc", _lib.MlpDesc), ("n_streams", ctypes.c_int), ("n_params", ctypes.c_int),
                    ("bwd_waves", ctypes.c_int), ("lds_bytes", ctypes.c_int), ("fwd", ctypes.c_void_p), ("bwd", ctypes.c_void_p)]
    BWD = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
    out = {}
    gen = torch.Generator(device="cpu").manual_seed(3)
    ld = (n + 63) // 64 * 64
    coords = torch.rand(2, ld, generator=gen).cuda()
    for label, off in (("fixed", "0"), ("raw", "1")):
        old_fix, old_flags = os.environ.get("NDQ_NO_PK_MFMA_FIX"), os.environ.get("NDQ_JIT_FLAGS")
        os.environ["NDQ_NO_PK_MFMA_FIX"] = off
        os.environ["NDQ_JIT_FLAGS"] = ((old_flags + " ") if old_flags else "") + "-DNDQ_BWD_THREADS=512"
        try:
            ext = ctypes.CDLL(codegen.build_mlp_ext(desc))
        finally:
            for k, v in (("NDQ_NO_PK_MFMA_FIX", old_fix), ("NDQ_JIT_FLAGS", old_flags)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        ext.ndq_ext_kernels.restype = ctypes.POINTER(Kernels)
        rec = ext.ndq_ext_kernels().contents
        bwd = BWD(rec.bwd)
        P, NS = rec.n_params, rec.n_streams
        params = (torch.rand(P, generator=gen) - 0.5).cuda()
        gbar = torch.randn(NS, ld, generator=gen).cuda()
        blocks = min(256, (n + 16 * rec.bwd_waves - 1) // (16 * rec.bwd_waves))
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ref, differ = None, 0
        for _ in range(launches):
            part = torch.zeros(blocks, P, device="cuda")
            rc = bwd(coords.data_ptr(), ld, n, params.data_ptr(), gbar.data_ptr(), ld, part.data_ptr(), blocks, stream)
            if rc != 0:
                raise RuntimeError(f"canary adjoint launch failed with code {rc}")
            torch.cuda.synchronize()
            if ref is None:
                ref = part
            elif not torch.equal(ref, part):
                differ += 1
        out[label] = differ
        out[label + "_waves"] = rec.bwd_waves
    STATUS["two_waves"] = out
    if out["fixed"] != 0:
        STATUS["refuse_two_waves"] = True
        warnings.warn(f"neurodiffeq_amd: a spilling adjoint kernel with two waves per SIMD returned different bits in "
                      f"{out['fixed']} of {launches} launches THROUGH the assembly fix-up pass; two-waves-per-SIMD builds are "
                      "refused for this process.", RuntimeWarning)
    return out
