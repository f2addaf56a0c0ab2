#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per kernel from the counter_collection csv files under a directory."""
import csv
import collections
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        short = ("fused_closure" if "fused_closure" in k else "bwd" if "jet_bwd" in k else "fwd" if "jet_fwd" in k
                 else "reduce_tail" if "reduce_tail" in k else "pointwise" if "ndq_pw" in k
                 else "reduce" if "reduce_partials" in k else None)
        if short is None:
            continue
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for name, vals in sorted(d.items()):
        vals = vals[len(vals) // 4:]          # skip warm-up dispatches
        print(f"   {name:28s} n={len(vals):4d} mean={sum(vals) / len(vals):14.1f}")
