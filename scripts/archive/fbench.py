#!/usr/bin/env python
"""Time the single-launch fused closure kernel of C2 (tuning aid, GPU only).  Variants are selected with
NDQ_JIT_FLAGS (compile-time macros of csrc/ndq_mlp.h); the kernel must have been pre-built on the CPU box
(scripts/fbench.py --build) because the GPU box only gets built .so files.
usage: NDQ_JIT_FLAGS="..." python scripts/fbench.py [--build] [n]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402
from neurodiffeq_amd import codegen  # noqa: E402
from neurodiffeq_amd.engine import FusedSystem, trace_system, _c_vp, _ptr  # noqa: E402

build_only = "--build" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = os.environ.get("FBENCH_CFG", "c2")
grid = int(args[0]) if args else {"c2": 256, "c3": 512}[name]
FLOP = {"c2": 32064, "c3": 198912}[name]
torch.manual_seed(0)
cfg = configs.make(name, grid)
if build_only:
    prog, descs = trace_system(cfg["nets"], cfg["conds"], cfg["pde"], 2)
    print(codegen.build_fused(prog, descs[0]))
    sys.exit(0)
for net in cfg["nets"]:
    net.to("cuda")
ref = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cuda", single_kernel=False)
sysm = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cuda", single_kernel=True)
torch.manual_seed(1)
batch = [c.detach() for c in cfg["gen"].get_examples()]
ref.step(batch, train=True); torch.cuda.synchronize()
g_ref = ref.flat[0].grad.cpu().numpy().copy(); l_ref = ref.loss_buf[0].item()
b, n = sysm.step(batch, train=True); torch.cuda.synchronize()
g = sysm.flat[0].grad.cpu().numpy(); l = sysm.loss_buf[0].item()
fp = sysm.flat[0]
stream = _c_vp(torch.cuda.current_stream().cuda_stream)


def closure():
    b["fusedk"].lib.ndq_fused_launch(sysm._coord_ptr(b, 0), b["ld"], n, _ptr(fp.flat), _ptr(b["fused_partials"]),
                                     _ptr(b["fused_loss_partials"]), None, None, b["ld"], 1.0 / n, 1, stream)


ITERS = 200 if name == "c2" else 60
for _ in range(20):
    closure()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(ITERS):
    closure()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / ITERS
print(json.dumps(dict(flags=os.environ.get("NDQ_JIT_FLAGS", ""), n=n, us=round(us, 2), tflops=round(FLOP * n / us / 1e6, 1),
                      grad_rel=float(np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref)), loss_rel=abs(l - l_ref) / abs(l_ref),
                      blocks=b["fused_blocks"])))
