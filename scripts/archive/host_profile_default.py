#!/usr/bin/env python
"""cProfile of fit() in the reference's default configuration (Solver1D, 32 noisy points, static validation grid)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import diff  # noqa: E402
from neurodiffeq_amd.conditions import IVP  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D  # noqa: E402

torch.manual_seed(0)
s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0)
s.fit(50, tqdm_file=None)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
s.fit(n, tqdm_file=None)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host {t_host / n * 1e6:.1f} us/epoch, wall {(time.perf_counter() - t0) / n * 1e6:.1f} us/epoch")
pr = cProfile.Profile()
pr.enable()
s.fit(n, tqdm_file=None)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
