#!/usr/bin/env python
"""Throughput of the other BASELINE configs (parity-test cases, not the headline bench line): run_train_epoch() with
resident pre-sampled batches, one JSON line per config.  usage: scripts/bench_configs.py [c1 c3 c4 c5 ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402
from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator  # noqa: E402

names = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]
for name in names:
    size = None
    if ":" in name:                                   # "c2:2048": the config at another batch size (grid side / point count)
        name, size = name.split(":")[0], int(name.split(":")[1])
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, size)
    solver.fused = "require"
    torch.manual_seed(1)
    solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 4, "cuda"))
    steps = 20 if name == "c5" else 100
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:            # warm-up by wall time: a cold box needs a moment to reach its clocks
        for _ in range(5):
            solver.run_train_epoch()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.run_train_epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sysm = solver._fused_sys
    print(json.dumps(dict(config=name, points=cfg["n_points"], ms_per_step=round(dt * 1e3, 4),
                          points_per_s=round(cfg["n_points"] / dt), single_launch=sysm.fusedk is not None,
                          native_epoch=getattr(sysm, "_fast", None) is not None,
                          final_loss=solver.metrics_history["train_loss"][-1])), flush=True)
