#!/usr/bin/env python
"""cProfile of fit() with a per-epoch callback at the reference's default ODE size (host-bound path: one training and one
validation epoch per Python iteration).  usage: scripts/host_profile_fit.py [epochs]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0)
s.fused = "require"
cb = [lambda solver: None]
s.fit(50, tqdm_file=None, callbacks=cb)
torch.cuda.synchronize()
t0 = time.perf_counter()
s.fit(n, tqdm_file=None, callbacks=cb)
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / n * 1e6:.1f} us per fit epoch")
pr = cProfile.Profile()
pr.enable()
s.fit(n, tqdm_file=None, callbacks=cb)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
