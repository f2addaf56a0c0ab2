#!/usr/bin/env python
"""Where does a C2 run_train_epoch() spend its time on the HOST?  A/B of the per-epoch state watch.
usage: scripts/host_step.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402
from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator  # noqa: E402

torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
torch.manual_seed(1)
solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 2, "cuda"))
for _ in range(300):
    solver.run_train_epoch()
torch.cuda.synchronize()


def measure(tag, k=4000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        solver.run_train_epoch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"case": tag, "host_us_per_epoch": (t1 - t0) / k * 1e6, "wall_us_per_epoch": (t2 - t0) / k * 1e6}


out = [measure("default")]
n_watch = len(solver._eq_watch) if solver._eq_watch is not None else None
t0 = time.perf_counter()
for _ in range(10000):
    solver._eq_watch.dirty()
watch_us = (time.perf_counter() - t0) / 10000 * 1e6
orig = type(solver)._equations_unchanged
type(solver)._equations_unchanged = lambda self, sysm, force=False: True
out.append(measure("no state watch"))
type(solver)._equations_unchanged = orig
out.append(measure("default again"))
print(json.dumps(dict(watch_entries=n_watch, watch_us=watch_us, runs=out)))
