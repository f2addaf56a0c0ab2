#!/usr/bin/env python
"""The tracer fuzzer's random equations (tests/test_tracer_fuzz.py) as SOLVERS: five epochs of fit() on the fused path and on plain
torch modules (fused="off", custom-op seam off) from one seed; prints the worst relative difference of the loss histories and of
the final parameters per seed.   usage: python scripts/fuzz_trajectories.py [n_seeds]"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import autograd_ops  # noqa: E402
from neurodiffeq_amd.generators import Generator1D, Generator2D  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D, Solver2D  # noqa: E402
from tests.test_tracer_fuzz import _system  # noqa: E402


def run(seed, fused):
    system, src = _system(seed)
    torch.manual_seed(1000 + seed)
    nets, conds, pde = system.product()
    for n in nets:
        n.cuda()
    if system.n_coords == 1:
        lo, hi = system.box[0]
        s = Solver1D(pde, conds, nets=nets, train_generator=Generator1D(64, lo, hi, method="equally-spaced-noisy"),
                     valid_generator=Generator1D(32, lo, hi, method="equally-spaced"))
    else:
        (x0, x1), (y0, y1) = system.box
        s = Solver2D(pde, conds, nets=nets, train_generator=Generator2D((12, 12), (x0, y0), (x1, y1), method="equally-spaced-noisy"),
                     valid_generator=Generator2D((8, 8), (x0, y0), (x1, y1), method="equally-spaced"))
    s.fused = fused
    torch.manual_seed(2000 + seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if fused == "off":
            with autograd_ops.native_autograd(False):
                s.fit(5)
        else:
            s.fit(5)
    h = s.metrics_history
    flat = torch.cat([p.detach().reshape(-1) for n in s.nets for p in n.parameters()]).double().cpu().numpy()
    return np.array(h["train_loss"]), np.array(h["valid_loss"]), flat, s.fused_active, src


rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))
worst = 0.0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    ft, fv, fp, active, src = run(seed, "auto")
    pt, pv, pp, _, _ = run(seed, "off")
    e = max(rel(ft, pt), rel(fv, pv), float(np.linalg.norm(fp - pp) / np.linalg.norm(pp)))
    worst = max(worst, e)
    print(f"seed {seed:2d} fused={active} train {rel(ft, pt):.1e} valid {rel(fv, pv):.1e} params {np.linalg.norm(fp - pp) / np.linalg.norm(pp):.1e}"
          + ("   <-- " + src[:150] if e > 1e-4 else ""), flush=True)
print("worst", worst)
