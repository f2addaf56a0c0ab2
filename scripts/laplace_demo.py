#!/usr/bin/env python
"""End-to-end demonstration (README example of the reference): Laplace's equation on the unit square with
u(0, y) = sin(pi y) and zero on the other edges, default Solver2D set-up apart from the batch (64 x 64 noisy grid),
trained on the fused path; reports wall time and the error against the analytic solution
u = sin(pi y) sinh(pi (1 - x)) / sinh(pi)."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import diff  # noqa: E402
from neurodiffeq_amd.conditions import DirichletBVP2D  # noqa: E402
from neurodiffeq_amd.generators import Generator2D  # noqa: E402
from neurodiffeq_amd.solvers import Solver2D  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
torch.manual_seed(0)
zero = lambda v: 0 * v
solver = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                  [DirichletBVP2D(0, lambda y: torch.sin(math.pi * y), 1, zero, 0, zero, 1, zero)],
                  xy_min=(0, 0), xy_max=(1, 1), train_generator=Generator2D((64, 64), (0, 0), (1, 1), "equally-spaced-noisy"),
                  valid_generator=Generator2D((64, 64), (0, 0), (1, 1), "equally-spaced"), n_batches_valid=1)
solver.fit(10, tqdm_file=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
solver.fit(epochs, tqdm_file=None)
best = solver.lowest_loss
torch.cuda.synchronize()
wall = time.perf_counter() - t0
xs, ys = np.meshgrid(np.linspace(0, 1, 101, dtype=np.float32), np.linspace(0, 1, 101, dtype=np.float32), indexing="ij")
u = solver.get_solution()(xs, ys, to_numpy=True)
exact = np.sin(math.pi * ys) * np.sinh(math.pi * (1 - xs)) / math.sinh(math.pi)
print(json.dumps(dict(epochs=epochs, wall_s=round(wall, 3), us_per_epoch=round(wall / epochs * 1e6, 1), fused=solver.fused_active,
                      best_valid_loss=best, rel_l2_error=float(np.linalg.norm(u - exact) / np.linalg.norm(exact)),
                      max_abs_error=float(np.abs(u - exact).max()))))
