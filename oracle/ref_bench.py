#!/usr/bin/env python
"""TEST INFRASTRUCTURE (oracle/): times the UNMODIFIED reference's training step on this host's CPU cores.

Run as a child process by bench.py's ``cpu_baseline`` leg only (importing the reference changes torch's global defaults,
``neurodiffeq/__init__.py:22``).  Needs ``oracle/_ref/`` (oracle/make_ref.sh).  Workload = BASELINE.json's headline config:
Solver2D, Laplace with DirichletBVP2D, FCNN 2-32-32-1, Generator2D 256 x 256 'equally-spaced-noisy', one
``solver.run_train_epoch()`` per step (sample + forward + diff() sweeps + loss + backward + Adam; SURVEY.md 8(d),
BASELINE.md section 2) after ``set_tensor_type('cpu', 32)``; also the same step on a pre-sampled batch and in the library's
default fp64.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_ref"))
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build(grid, bits, presampled=False):
    from neurodiffeq import diff
    from neurodiffeq.conditions import DirichletBVP2D
    from neurodiffeq.generators import Generator2D, PredefinedGenerator
    from neurodiffeq.networks import FCNN
    from neurodiffeq.solvers import Solver2D
    from neurodiffeq.utils import set_tensor_type
    set_tensor_type(device="cpu", float_bits=bits)
    torch.manual_seed(0)
    pde = lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)]
    conds = [DirichletBVP2D(x_min=0, x_min_val=lambda y: torch.sin(np.pi * y), x_max=1, x_max_val=lambda y: 0,
                            y_min=0, y_min_val=lambda x: 0, y_max=1, y_max_val=lambda x: 0)]
    gen = Generator2D((grid, grid), (0, 0), (1, 1), "equally-spaced-noisy")
    if presampled:
        xs, ys = gen.get_examples()
        gen = PredefinedGenerator(xs.detach(), ys.detach())
    return Solver2D(pde, conds, xy_min=(0, 0), xy_max=(1, 1), nets=[FCNN(2, 1, hidden_units=(32, 32))],
                    train_generator=gen, valid_generator=gen, n_batches_valid=0)


def median_time(fn, min_runs, budget_s, max_runs=200):
    times, t_start = [], time.perf_counter()
    while len(times) < max_runs and (len(times) < min_runs or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 3 * budget_s:
            break
    times.sort()
    return times[len(times) // 2], len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--budget", type=float, default=8.0)
    args = ap.parse_args()
    import neurodiffeq  # noqa: F401
    n = args.grid * args.grid
    solver = build(args.grid, 32)
    ncpu = os.cpu_count() or 1
    best = None
    for t in sorted({1, 4, 8, 16, 32, 64, min(128, ncpu)}):      # the thread count a CPU user would tune to
        if t > ncpu:
            continue
        torch.set_num_threads(t)
        solver.run_train_epoch()
        t0 = time.perf_counter()
        solver.run_train_epoch(); solver.run_train_epoch()
        dt = (time.perf_counter() - t0) / 2
        if best is None or dt < best[0]:
            best = (dt, t)
    torch.set_num_threads(best[1])
    for _ in range(2):
        solver.run_train_epoch()
    med, runs = median_time(solver.run_train_epoch, 8, args.budget)
    pre = build(args.grid, 32, presampled=True)
    pre.run_train_epoch()
    med_pre, runs_pre = median_time(pre.run_train_epoch, 5, args.budget / 2)
    s64 = build(args.grid, 64)
    s64.run_train_epoch()
    med64, runs64 = median_time(s64.run_train_epoch, 3, args.budget / 2)
    rev = open(os.path.join(HERE, "_ref", "REVISION")).read().strip() if os.path.exists(os.path.join(HERE, "_ref", "REVISION")) else "unknown"
    print(json.dumps(dict(points=n, ms_per_step=med * 1e3, value=n / med, runs=runs, threads=best[1],
                          presampled=dict(ms_per_step=med_pre * 1e3, value=n / med_pre, runs=runs_pre),
                          fp64=dict(ms_per_step=med64 * 1e3, value=n / med64, runs=runs64),
                          final_loss=float(solver.metrics_history["train_loss"][-1]), reference_revision=rev,
                          torch=torch.__version__, logical_cpus=ncpu)))


if __name__ == "__main__":
    main()
