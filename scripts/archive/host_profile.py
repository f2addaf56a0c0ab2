#!/usr/bin/env python
"""cProfile of the host side of run_train_epoch() on the native path (C2, resident batches)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402
from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator  # noqa: E402

torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 8, "cuda"))
for _ in range(600):
    solver.run_train_epoch()
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    solver.run_train_epoch()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host-side enqueue time {t_host / n * 1e6:.1f} us/epoch; wall incl. GPU drain {t_all / n * 1e6:.1f} us/epoch")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    solver.run_train_epoch()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
