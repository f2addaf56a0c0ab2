#!/usr/bin/env python
"""Host-side cost of one headline run_train_epoch() call (C2, resident batches): cProfile over N calls, top functions by own
time, plus the uninstrumented per-call time with the GPU kept far behind (enqueue only).
usage: scripts/host_profile.py [calls] > gpurun_out/host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd.generators import Generator2D, ResidentBatchGenerator, SamplerGenerator  # noqa: E402
from tests import configs  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
torch.manual_seed(1)
gen = Generator2D((256, 256), (0, 0), (1, 1), "equally-spaced-noisy")
solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(gen, 8, "cuda", lo=0, hi=65536))
for _ in range(300):
    solver.run_train_epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    solver.run_train_epoch()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue per call (2000 calls, no profiler): {(t1 - t0) / 2000 * 1e6:.2f} us; with final sync "
      f"{(time.perf_counter() - t0) / 2000 * 1e6:.2f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    solver.run_train_epoch()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue())
