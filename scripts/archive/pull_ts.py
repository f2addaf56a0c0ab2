#!/usr/bin/env python
"""Where the pull prologue of a one-launch epoch spends its time (wall-clock stamps of thread 0 of workgroup 0).
usage: NDQ_JIT_FLAGS=-DNDQ_PHASE_TS python scripts/pull_ts.py [epochs]"""
import ctypes
import os
import sys

os.environ.setdefault("NDQ_JIT_FLAGS", "-DNDQ_PHASE_TS")
os.environ["NDQ_FIT_PULL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neurodiffeq_amd import diff  # noqa: E402
from neurodiffeq_amd.conditions import IVP  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0)
s.fused = "require"
for rep in range(4):
    s.fit(epochs, tqdm_file=None)
    torch.cuda.synchronize()
    fk = s._fused_sys.fusedk
    pt = (ctypes.c_ulonglong * 8)()
    ts = (ctypes.c_ulonglong * (256 * 8))()
    fk.lib.ndq_fused_pull_ts.argtypes = [ctypes.c_void_p]
    fk.lib.ndq_fused_phase_ts.argtypes = [ctypes.c_void_p]
    assert fk.lib.ndq_fused_pull_ts(pt) == 0 and fk.lib.ndq_fused_phase_ts(ts) == 0
    p = np.frombuffer(pt, dtype=np.uint64).astype(np.int64) * 10.0          # ns
    t = np.frombuffer(ts, dtype=np.uint64).reshape(256, 8)[:2, 4:].astype(np.int64) * 10.0
    # NOTE the stamps read here belong to the LAST closure launch with a prologue and to the very last launch (the
    # trailing validation pair has no prologue): only differences inside one array are meaningful
    print("prologue %.2f us | barrier + staging %.2f us   (the previous launch's workgroups 0 / 1 ended %.2f / %.2f us "
          "before this prologue began)" % ((p[2] - p[0]) / 1e3, (p[3] - p[2]) / 1e3, (p[0] - p[4]) / 1e3, (p[0] - p[5]) / 1e3))
    print("last launch, workgroup 0/1: start->staged %s us, loop %s us, epilogue %s us" % (
        (t[:, 1] - t[:, 0]) / 1e3, (t[:, 2] - t[:, 1]) / 1e3, (t[:, 3] - t[:, 2]) / 1e3))
