#!/usr/bin/env python
"""BUILD CONTAINER ONLY (needs /root/reference): how far the oracle's port of the training step (oracle/autograd_ref.py
TrainLoop -- what bench.py times as `cpu_baseline`, kind "port") is from the UNMODIFIED reference on the same host,
config C2 (65 536 points, fp32), alternating runs.  The result is quoted in BASELINE.md so that `cpu_baseline.kind:
"port"` can be read as "the reference within that ratio"."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "golden", "_refshim"), "/root/reference", ROOT]
os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import neurodiffeq  # noqa: E402,F401
from neurodiffeq import diff  # noqa: E402
from neurodiffeq.conditions import DirichletBVP2D  # noqa: E402
from neurodiffeq.generators import Generator2D  # noqa: E402
from neurodiffeq.networks import FCNN  # noqa: E402
from neurodiffeq.solvers import Solver2D  # noqa: E402
from neurodiffeq.utils import set_tensor_type  # noqa: E402

set_tensor_type(device="cpu", float_bits=32)
torch.set_num_threads(min(8, os.cpu_count() or 1))
from oracle import autograd_ref as R  # noqa: E402

torch.manual_seed(0)
gen = Generator2D((256, 256), (0, 0), (1, 1), "equally-spaced-noisy")
ref = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
               [DirichletBVP2D(0, lambda y: torch.sin(np.pi * y), 1, lambda y: 0, 0, lambda x: 0, 1, lambda x: 0)],
               nets=[FCNN(2, 1, hidden_units=(32, 32))], train_generator=gen, valid_generator=gen, n_batches_valid=0,
               xy_min=(0, 0), xy_max=(1, 1))
torch.manual_seed(0)
cfg = R.build_config("c2", 256)
port = R.TrainLoop(cfg["nets"], cfg["enforcers"], cfg["pde"], cfg["sampler"])
for _ in range(2):
    ref.run_train_epoch(); port.epoch()
t_ref, t_port = [], []
for _ in range(8):
    t0 = time.perf_counter(); ref.run_train_epoch(); t_ref.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); port.epoch(); t_port.append(time.perf_counter() - t0)
out = dict(threads=torch.get_num_threads(), cpus=os.cpu_count(), reference_ms_median=float(np.median(t_ref) * 1e3),
           port_ms_median=float(np.median(t_port) * 1e3), port_over_reference=float(np.median(t_port) / np.median(t_ref)),
           reference_ms=[round(t * 1e3, 1) for t in t_ref], port_ms=[round(t * 1e3, 1) for t in t_port])
print(json.dumps(out))
