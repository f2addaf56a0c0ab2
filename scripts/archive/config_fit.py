#!/usr/bin/env python
"""bench.py's per-config record (run_train_epoch() loop and fit(k)) for a few configs only.
usage: scripts/config_fit.py c1 [c4 ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for name in sys.argv[1:] or ["c1"]:
    r = bench.config_record(name)
    print(json.dumps({"config": name, **{k: r[k] for k in ("points", "ms_per_step", "ms_per_step_run_train_epoch",
                                                              "ms_per_step_in_fit", "launches_per_step", "final_loss")}}), flush=True)
