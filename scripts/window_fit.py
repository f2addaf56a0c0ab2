#!/usr/bin/env python
"""What a K-step timing window of the headline (bench.py's protocol: synchronize, K x run_train_epoch(), synchronize) costs
beyond K GPU steps: windows of K = 1 ... 400 steps, median of each, least-squares line T(K) = c + g K over the long windows,
the host's enqueue time per step (loop without the closing synchronize), the cost of an empty window, and of a window around
ONE tiny kernel.
usage: scripts/window_fit.py > gpurun_out/window_fit.json"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd.generators import Generator2D, ResidentBatchGenerator, SamplerGenerator  # noqa: E402
from tests import configs  # noqa: E402

torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
torch.manual_seed(1)
gen = Generator2D((256, 256), (0, 0), (1, 1), "equally-spaced-noisy")
solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(gen, 8, "cuda", lo=0, hi=65536))
for _ in range(300):
    solver.run_train_epoch()
sync = torch.cuda.synchronize


def med_window(k, reps, body=None):
    body = body or solver.run_train_epoch
    ts, hs = [], []
    for _ in range(reps):
        sync(); sync()
        t0 = time.perf_counter()
        for _ in range(k):
            body()
        t1 = time.perf_counter()
        sync(); sync()
        t2 = time.perf_counter()
        ts.append(t2 - t0)
        hs.append(t1 - t0)
    return statistics.median(ts) * 1e6, statistics.median(hs) * 1e6


out = {"windows_us": {}, "host_enqueue_us_per_step": {}}
for k in (1, 2, 3, 5, 10, 20, 40, 100, 200, 400):
    t, h = med_window(k, max(30, 4000 // k))
    out["windows_us"][k] = round(t, 2)
    out["host_enqueue_us_per_step"][k] = round(h / k, 2)
ks = [40, 100, 200, 400]
xs, ys = ks, [out["windows_us"][k] for k in ks]
n = len(xs)
g = (n * sum(x * y for x, y in zip(xs, ys)) - sum(xs) * sum(ys)) / (n * sum(x * x for x in xs) - sum(xs) ** 2)
c = (sum(ys) - g * sum(xs)) / n
out["fit"] = {"gpu_step_us": round(g, 3), "fixed_us": round(c, 2), "from_windows": ks}
out["per_step_us"] = {k: round(v / k, 2) for k, v in out["windows_us"].items()}
out["empty_window_us"] = round(med_window(0, 300)[0], 2)
x = torch.zeros(64, device="cuda")
out["one_tiny_kernel_window_us"] = round(med_window(1, 300, body=lambda: x.add_(1.0))[0], 2)
out["twenty_tiny_kernels_window_us"] = round(med_window(20, 300, body=lambda: x.add_(1.0))[0], 2)
print(json.dumps(out, indent=1))
