#!/usr/bin/env python
"""C2 in the reference's default precision (fp64 networks) through run_train_epoch(): the loop bench.py's `c2_fp64` times, on
its own so that a kernel trace shows where an fp64 epoch goes.   usage: scripts/fp64_step.py [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
unchanged_script = len(sys.argv) > 2 and sys.argv[2] == "default"
if unchanged_script:
    # what importing the reference does (neurodiffeq/__init__.py:22): cuda + float64 defaults, then a plain Solver2D with
    # its noisy 256 x 256 generator -- a fresh batch every epoch, drawn on the device, handed out in double
    from neurodiffeq_amd.utils import set_tensor_type
    set_tensor_type(device="cuda", float_bits=64)
torch.manual_seed(0)
solver, cfg = configs.make_solver("c2")
for net in cfg["nets"]:
    net.double()
solver.fused = "require"
torch.manual_seed(1)
if not unchanged_script:
    batch = [c.detach().double().cuda().reshape(-1, 1) for c in cfg["gen"].get_examples()]
    solver.generator["train"].get_examples = lambda: batch
for _ in range(50):
    solver.run_train_epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    solver.run_train_epoch()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps(dict(config="c2_fp64" + ("_unchanged_script_cuda_float64_defaults" if unchanged_script else ""), points=cfg["n_points"], us_per_step=round(dt * 1e6, 2), steps=steps,
                      final_loss=solver.metrics_history["train_loss"][-1])))
