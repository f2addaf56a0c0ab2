#!/usr/bin/env python
"""Wide single-hidden-layer networks (csrc/ndq_wide.h) at BASELINE C2's grid: closure-kernel launch time (HIP events, back to
back), run_train_epoch() and fit() step time.  usage: scripts/wide_bench.py [w16:256 w17:256 ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tests import configs  # noqa: E402
from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator  # noqa: E402

for spec in (sys.argv[1:] or ["w16:256", "w17:256"]):
    name, size = spec.split(":")[0], int(spec.split(":")[1])
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, size)
    solver.fused = "require"
    torch.manual_seed(1)
    solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 2, "cuda"))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(5):
            solver.run_train_epoch()
        torch.cuda.synchronize()
    sysm = solver._fused_sys
    times = bench.timed_windows(solver.run_train_epoch, 50, torch.cuda.synchronize)
    dt = times[(len(times) - 1) // 2] / 50
    solver.fit(200, tqdm_file=None)
    tf = bench.timed_windows(lambda: solver.fit(200, tqdm_file=None), 1, torch.cuda.synchronize)
    dt_fit = tf[(len(tf) - 1) // 2] / 200
    batch = solver._batch["train"]
    brk = bench.fused_breakdown(sysm, [c.detach() for c in batch]) if sysm.fusedk is not None else \
        dict(fused_closure=dict(us=0.0, blocks=0))
    kb = bench.kernel_breakdown(sysm, [c.detach() for c in batch]) if len(sysm.flat) == 1 else {}
    n = cfg["n_points"]
    print(json.dumps(dict(config=name, points=n, us_per_step=round(dt * 1e6, 2), us_per_step_in_fit=round(dt_fit * 1e6, 2),
                          points_per_s=round(n / min(dt, dt_fit)), closure_us=round(brk["fused_closure"]["us"], 2),
                          blocks=brk["fused_closure"]["blocks"], threads=sysm.fusedk.threads if sysm.fusedk is not None else 0,
                          fwd_us=round(kb.get("mlp_jet_fwd", {}).get("us", 0), 2), bwd_us=round(kb.get("mlp_jet_bwd", {}).get("us", 0), 2),
                          pw_us=round(kb.get("pointwise", {}).get("us", 0), 2),
                          flags=os.environ.get("NDQ_JIT_FLAGS", ""), final_loss=solver.metrics_history["train_loss"][-1])), flush=True)
