#!/usr/bin/env python
"""Where a launch of the GROUPED closure kernel (C4) goes: staging | group loop | epilogue per workgroup, and inside the last group
of wave 0 of workgroup 0: phase 1 (forward of the group's tiles -> exchange tile) | phase 2 (per-point program) | phase 3 (forward
with kept states + reverse pass).   usage: NDQ_JIT_FLAGS=-DNDQ_PHASE_TS python scripts/phase_ts_group.py [config[:size]]"""
import ctypes
import os
import sys

os.environ.setdefault("NDQ_JIT_FLAGS", "-DNDQ_PHASE_TS")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tests import configs  # noqa: E402
from neurodiffeq_amd.engine import FusedSystem  # noqa: E402

arg = sys.argv[1] if len(sys.argv) > 1 else "c4"
name, size = (arg.split(":")[0], int(arg.split(":")[1])) if ":" in arg else (arg, None)
torch.manual_seed(0)
cfg = configs.make(name, size)
for net in cfg["nets"]:
    net.to("cuda")
system = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], configs.n_coords(cfg), "cuda", compute_func_val=configs.func_val(cfg))
ex = cfg["gen"].get_examples()
batch = [c.detach().cuda() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
for _ in range(200):
    b, n = system.step(batch, train=True)
torch.cuda.synchronize()
fk = b["fusedk"]
print("threads per workgroup:", fk.threads, "blocks:", b["fused_blocks"], "points:", n)
buf = (ctypes.c_ulonglong * (256 * 8))()
fk.lib.ndq_fused_phase_ts.argtypes = [ctypes.c_void_p]
tt = (ctypes.c_ulonglong * 48)()
fk.lib.ndq_fused_tile_ts.argtypes = [ctypes.c_void_p]
for rep in range(4):
    system.step(batch, train=True)
    torch.cuda.synchronize()
    assert fk.lib.ndq_fused_phase_ts(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8)[:b["fused_blocks"]].astype(np.int64)
    cyc, wall = t[:, :4], t[:, 4:] * 10.0
    d, dc = np.diff(wall, axis=1).mean(0), np.diff(cyc, axis=1).mean(0)
    print("stage %.2f us | group loop %.2f us | epilogue %.2f us (cycles %d / %d / %d); first start -> last end %.2f us, start spread %.2f us"
          % (d[0] / 1e3, d[1] / 1e3, d[2] / 1e3, dc[0], dc[1], dc[2], (wall[:, 3].max() - wall[:, 0].min()) / 1e3,
             (wall[:, 0].max() - wall[:, 0].min()) / 1e3))
    assert fk.lib.ndq_fused_tile_ts(tt) == 0
    g = np.frombuffer(tt, dtype=np.uint64).astype(np.int64).reshape(2, 24)
    for it in (0, 1):
        p = g[it, :4]
        if p[3] > p[0] > 0:
            print("   group (slot %d) of wave 0 / workgroup 0: phase 1 %d | phase 2 %d | phase 3 %d cycles" % (it, p[1] - p[0], p[2] - p[1], p[3] - p[2]))
