#!/usr/bin/env python
"""Where a closure-kernel launch spends its time: per-workgroup timestamps (weight staging | tile loop | reduction
epilogue) and the spread of workgroup start / end times.  Builds the C2 closure kernel with -DNDQ_PHASE_TS.
usage: NDQ_JIT_FLAGS=-DNDQ_PHASE_TS python scripts/phase_ts.py [config[:size]] [threads]"""
import ctypes
import os
import sys

os.environ.setdefault("NDQ_JIT_FLAGS", "-DNDQ_PHASE_TS")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tests import configs  # noqa: E402
from neurodiffeq_amd.engine import FusedSystem, _ptr  # noqa: E402

arg = sys.argv[1] if len(sys.argv) > 1 else "c2"
name, size = (arg.split(":")[0], int(arg.split(":")[1])) if ":" in arg else (arg, None)
torch.manual_seed(0)
cfg = configs.make(name, size)
for net in cfg["nets"]:
    net.to("cuda")
system = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], configs.n_coords(cfg), "cuda",
                     compute_func_val=configs.func_val(cfg))
ex = cfg["gen"].get_examples()
batch = [c.detach().cuda() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
for _ in range(300):
    b, n = system.step(batch, train=True)
torch.cuda.synchronize()
fk = b["fusedk"]
print("threads per workgroup:", fk.threads, "blocks:", b["fused_blocks"], "points:", n)
buf = (ctypes.c_ulonglong * (256 * 8))()
fk.lib.ndq_fused_phase_ts.argtypes = [ctypes.c_void_p]
rows = []
for rep in range(5):
    system.step(batch, train=True)
    torch.cuda.synchronize()
    assert fk.lib.ndq_fused_phase_ts(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8)[:b["fused_blocks"]].astype(np.int64)
    cyc, wall = t[:, :4], t[:, 4:] * 10.0          # wall clock: 100 MHz -> ns
    d = np.diff(wall, axis=1)
    dc = np.diff(cyc, axis=1)
    span = wall[:, 3].max() - wall[:, 0].min()
    rows.append((d.mean(0), dc.mean(0), span, wall[:, 0].max() - wall[:, 0].min(), wall[:, 3].max() - wall[:, 3].min()))
for d, dc, span, s0, s3 in rows:
    print("stage %.2f us | loop %.2f us | epilogue %.2f us  (cycles %d / %d / %d)   first start -> last end %.2f us, "
          "start spread %.2f us, end spread %.2f us" % (d[0] / 1e3, d[1] / 1e3, d[2] / 1e3, dc[0], dc[1], dc[2],
                                                        span / 1e3, s0 / 1e3, s3 / 1e3))

# inside the last tile of wave 0 of workgroup 0: 0 start | 1 forward | 2 output layer | 3 pointwise | 4 output adjoint |
# per hidden layer L..2: act_backward, weight_grad, hbar GEMM | 12 end of tile (first-layer adjoint + sums)
tt = (ctypes.c_ulonglong * 48)()
fk.lib.ndq_fused_tile_ts.argtypes = [ctypes.c_void_p]
assert fk.lib.ndq_fused_tile_ts(tt) == 0
both = np.frombuffer(tt, dtype=np.uint64).astype(np.int64).reshape(2, 24)
names = {0: "start", 1: "forward", 2: "output", 3: "pointwise", 4: "out-adjoint", 12: "first-layer adjoint / end"}
layers = system.descs[0].layers
for k in range(layers - 1):
    names[5 + 3 * k] = f"act_backward L{layers - k}"
    names[6 + 3 * k] = f"weight_grad L{layers - k}"
    names[7 + 3 * k] = f"hbar gemm L{layers - k}"
for which, t in zip(("FIRST tile of wave 0 / workgroup 0", "a LATER tile (the last one)"), both):
    if not t[12]:
        continue
    print(which)
    prev = t[0]
    for k in sorted(names):
        if t[k]:
            print(f"  {names[k]:28s} +{t[k] - prev:6d} cycles")
            prev = t[k]
    print("  tile total", t[12] - t[0], "cycles")
    # epilogue of workgroup 0 (block_reduce_store + loss sums), thread 0: 13 entry | 14 lane sums done | 15 barrier | 16 LDS
    # deposits | 17 barrier | 18 region sums + global stores issued | 19 loss / theta sums
    ep = {13: "epilogue entry", 14: "lane (DPP) sums", 15: "barrier 1", 16: "LDS deposits", 17: "barrier 2",
          18: "region sums + stores", 19: "loss sums"}
    if t[13] and t[19]:
        prev = t[12]
        for k in sorted(ep):
            print(f"  {ep[k]:28s} +{t[k] - prev:6d} cycles")
            prev = t[k]
