#!/usr/bin/env python
"""Host-side cost of run_train_epoch() for an UNCHANGED user script under torch.set_default_device('cuda') (bench.py's
default_generator_cuda leg: plain Solver2D + Generator2D, noise drawn by the device sampler every step): cProfile, top functions.
usage: scripts/host_profile_default_device.py [calls]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
torch.set_default_device("cuda")
torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
for _ in range(300):
    solver.run_train_epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    solver.run_train_epoch()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"per call (2000 calls): enqueue {(t1 - t0) / 2000 * 1e6:.2f} us; with final sync {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    solver.run_train_epoch()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue())
