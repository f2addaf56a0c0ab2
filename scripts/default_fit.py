#!/usr/bin/env python
"""Wall time of solver.fit() in the reference's default configuration (README-style script: default networks, default
generators -- 32 noisy points for training (32 x 32 in 2-D), 4 static validation batches per epoch).
usage: scripts/default_fit.py [epochs] [--f64]

Per problem: 'fit' = fit(n) as a user calls it (whole chunks of epochs per native call), 'per_epoch' = the same epochs
with a no-op callback (one training + one validation epoch per call: the round-2 behaviour of the host loop), 'off' =
the reference's closure on torch autograd on the same GPU.  The loss histories of 'fit' and 'per_epoch' are compared bit
for bit."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import diff  # noqa: E402
from neurodiffeq_amd.conditions import IVP, DirichletBVP2D  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D, Solver2D  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
if "--f64" in sys.argv:       # the reference's import defaults as well: cuda default device AND float64 (neurodiffeq/__init__.py:22)
    from neurodiffeq_amd.utils import set_tensor_type
    set_tensor_type(device="cuda", float_bits=64)
zero = lambda v: 0 * v
PROBLEMS = {
    "ode": lambda: Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0),
    "pde": lambda: Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                            [DirichletBVP2D(0, lambda y: torch.sin(3.14159265 * y), 1, zero, 0, zero, 1, zero)],
                            xy_min=(0, 0), xy_max=(1, 1)),
    # the README's Lotka-Volterra system: two default networks, one per unknown
    "system": lambda: Solver1D(lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)],
                               [IVP(0.0, 1.5), IVP(0.0, 1.0)], t_min=0.1, t_max=12.0),
}
out = {"command": " ".join(["python"] + sys.argv), "epochs": epochs}
for name, make in PROBLEMS.items():
    hist = {}
    for mode in ("fit", "per_epoch", "off"):
        n = epochs if mode != "off" else max(epochs // 10, 50)
        torch.manual_seed(0)
        s = make()
        s.fused = "off" if mode == "off" else "require"
        cbs = [lambda solver: None] if mode == "per_epoch" else ()
        s.fit(20, tqdm_file=None, callbacks=cbs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.fit(n, tqdm_file=None, callbacks=cbs)
        t_host = time.perf_counter() - t0
        _ = s.metrics_history["valid_loss"][-1]
        torch.cuda.synchronize()
        out[f"{name}_{mode}_us_per_epoch"] = round((time.perf_counter() - t0) / n * 1e6, 2)
        out[f"{name}_{mode}_host_us_per_epoch"] = round(t_host / n * 1e6, 2)
        out[f"{name}_{mode}_final_valid_loss"] = s.metrics_history["valid_loss"][-1]
        hist[mode] = (list(s.metrics_history["train_loss"]), list(s.metrics_history["valid_loss"]))
    out[f"{name}_fit_history_bit_identical_to_per_epoch"] = hist["fit"] == hist["per_epoch"]
print(json.dumps(out))
