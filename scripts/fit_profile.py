#!/usr/bin/env python
"""fit() of the reference's default problems for a kernel trace: python scripts/fit_profile.py <ode|pde|system> [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import diff  # noqa: E402
from neurodiffeq_amd.conditions import IVP, DirichletBVP2D  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D, Solver2D  # noqa: E402

name = sys.argv[1]
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
zero = lambda v: 0 * v
torch.manual_seed(0)
if name == "ode":
    s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0)
elif name == "pde":
    s = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                 [DirichletBVP2D(0, lambda y: torch.sin(3.14159265 * y), 1, zero, 0, zero, 1, zero)], xy_min=(0, 0), xy_max=(1, 1))
else:
    s = Solver1D(lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)], [IVP(0.0, 1.5), IVP(0.0, 1.0)],
                 t_min=0.1, t_max=12.0)
s.fused = "require"
if os.environ.get("NDQ_FIT_TRACE"):
    s._fit_trace = []
s.fit(20, tqdm_file=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
s.fit(epochs, tqdm_file=None)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{name}: {(time.perf_counter() - t0) / epochs * 1e6:.2f} us/epoch (host enqueue {host / epochs * 1e6:.2f})")
if getattr(s, "_fit_trace", None):
    print("chunks: (epochs, draw ms, stage ms, enqueue ms)")
    for k, a, b, c in s._fit_trace:
        print(f"  {k:4d} {a * 1e3:8.3f} {b * 1e3:8.3f} {c * 1e3:8.3f}")
