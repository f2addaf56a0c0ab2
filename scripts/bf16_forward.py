#!/usr/bin/env python
"""The opt-in "bf16 forward / fp32 gradient" mode BASELINE config 5 names (csrc/ndq_mlp.h NDQ_FWD_BF16X1: the hidden-layer
GEMMs of the forward STREAM kernel on single bf16 operands; selected by building libndq with
NDQ_LIB_FLAGS=-DNDQ_FWD_BF16X1=1): what it costs in accuracy and what it buys, on C5 (three-kernel pipeline).

    python scripts/bf16_forward.py run OUT.npz [config[:size]]     one closure + timed steps with the library the environment selects
    python scripts/bf16_forward.py compare A.npz B.npz             relative differences of B (opt-in) against A (default)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300))


if sys.argv[1] == "compare":
    A, B = np.load(sys.argv[2]), np.load(sys.argv[3])
    print({"loss_rel": abs(float(B["loss"]) - float(A["loss"])) / abs(float(A["loss"])), "residuals_rel_l2": rel(B["resid"], A["resid"]),
           "funcs_rel_l2": rel(B["funcs"], A["funcs"]), "grad_rel_l2": rel(B["grad"], A["grad"]),
           "ms_per_step_default": float(A["ms"]), "ms_per_step_bf16_forward": float(B["ms"]),
           "lib_default": str(A["lib"]), "lib_bf16_forward": str(B["lib"])})
    sys.exit(0)

import torch  # noqa: E402
from tests import configs  # noqa: E402
from neurodiffeq_amd import _build  # noqa: E402
from neurodiffeq_amd.engine import FusedSystem  # noqa: E402

arg = sys.argv[3] if len(sys.argv) > 3 else "c5"
name, size = (arg.split(":")[0], int(arg.split(":")[1])) if ":" in arg else (arg, None)
torch.manual_seed(0)
cfg = configs.make(name, size)
for net in cfg["nets"]:
    net.to("cuda")
fs = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], configs.n_coords(cfg), "cuda", compute_func_val=configs.func_val(cfg))
ex = cfg["gen"].get_examples()
batch = [c.detach().cuda() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
b, n = fs.step(batch, train=True, slot=0, want_funcs=True, want_resid=True)
torch.cuda.synchronize()
out = dict(loss=fs.loss_buf[0].item(), resid=b["resid"][:, :n].cpu().numpy(), funcs=b["funcs"][:, :n].cpu().numpy(),
           grad=np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat]), lib=os.path.basename(_build.LIB))
for _ in range(5):
    fs.step(batch, train=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fs.step(batch, train=True)
torch.cuda.synchronize()
out["ms"] = (time.perf_counter() - t0) / 20 * 1e3
np.savez(sys.argv[2], **out)
print(name, n, "points:", out["lib"], "loss", out["loss"], "ms per closure", round(out["ms"], 4))
