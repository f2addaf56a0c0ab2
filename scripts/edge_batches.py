#!/usr/bin/env python
"""Edge batches through the fused path and through plain torch (fused="off") from one seed: 1 / 15 / 17 points, no points at all, huge
/ tiny / non-finite coordinates.  Prints the loss histories side by side.   usage: python scripts/edge_batches.py"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd import autograd_ops, diff  # noqa: E402
from neurodiffeq_amd.conditions import IVP  # noqa: E402
from neurodiffeq_amd.generators import PredefinedGenerator  # noqa: E402
from neurodiffeq_amd.networks import FCNN  # noqa: E402
from neurodiffeq_amd.solvers import Solver1D  # noqa: E402

CASES = {
    "one_point": torch.tensor([0.7]),
    "fifteen": torch.linspace(0.0, 2.0, 15),
    "seventeen": torch.linspace(0.0, 2.0, 17),
    "empty": torch.zeros(0),
    "huge": torch.tensor([0.0, 1.0, 1e6, 1e12, 1e30]),
    "tiny": torch.tensor([0.0, 1e-30, 1e-38, 1e-44, -1e-20]),
    "negative": torch.linspace(-50.0, -1.0, 33),
    "with_nan": torch.tensor([0.0, 1.0, float("nan"), 2.0]),
    "with_inf": torch.tensor([0.0, 1.0, float("inf"), 2.0]),
    "repeated": torch.full((40,), 1.25),
}


def run(fused, pts):
    torch.manual_seed(2)
    s = Solver1D(lambda u, t: [diff(u, t, order=2) + u * diff(u, t) - torch.sin(t)], [IVP(0.0, 1.0, 0.5)],
                 nets=[FCNN(1, 1, hidden_units=(32, 32)).cuda()], train_generator=PredefinedGenerator(pts.clone()),
                 valid_generator=PredefinedGenerator(pts.clone()))
    s.fused = fused
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            if fused == "off":
                with autograd_ops.native_autograd(False):
                    s.fit(4)
            else:
                s.fit(4)
        except Exception as e:      # noqa: BLE001
            return f"{type(e).__name__}: {str(e)[:80]}", None, s
    flat = torch.cat([p.detach().reshape(-1) for p in s.nets[0].parameters()]).double().cpu().numpy()
    return np.array(s.metrics_history["train_loss"]), flat, s


for name, pts in CASES.items():
    a, pa, sa = run("auto", pts)
    b, pb, sb = run("off", pts)
    if isinstance(a, str) or isinstance(b, str):
        print(f"{name:12s} fused: {a if isinstance(a, str) else a.tolist()}\n{'':12s} plain: {b if isinstance(b, str) else b.tolist()}")
        continue
    same_nan = np.array_equal(np.isnan(a), np.isnan(b))
    fin = np.isfinite(a) & np.isfinite(b)
    rel = float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-30))) if fin.any() else 0.0
    pn = np.array_equal(np.isnan(pa), np.isnan(pb))
    pfin = np.isfinite(pa) & np.isfinite(pb)
    prel = float(np.linalg.norm(pa[pfin] - pb[pfin]) / max(np.linalg.norm(pb[pfin]), 1e-30)) if pfin.any() else 0.0
    flag = "ok" if (same_nan and pn and rel < 1e-4 and prel < 1e-4) else "**DIFFER**"
    print(f"{flag:10s} {name:12s} fused_active={sa.fused_active} loss rel {rel:.1e} nan-pattern same={same_nan}; params rel {prel:.1e} nan-pattern same={pn}; fused {a.tolist()} plain {b.tolist()}")
