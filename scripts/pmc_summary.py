#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per kernel from the counter_collection csv files under a directory.
Kernels of this package are listed by a short name that keeps the template arguments (Cfg<...>) apart."""
import csv
import collections
import glob
import os
import re
import sys

root = sys.argv[1]


def short(k):
    m = re.search(r"(fused_group_closure|fused_multi_closure|fused_closure|mlp_jet_bwd|mlp_jet_fwd)_kernel<ndq::Cfg<([^>]*)>", k)
    if m:
        return f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
    m = re.search(r"(wide_closure_tv|wide_closure|wide_jet_fwd|wide_jet_bwd)_kernel<ndq::WideCfg<([^>]*)>", k)
    if m:
        return f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
    m = re.search(r"(deep_gemm_bf|deep_fwd_gemm|deep_bwd_gemm|deep_wgrad_gemm|deep_wgrad_bf|deep_head_fwd|deep_head_bwd)<ndq::DeepCfg<([^>]*)>((?:, \w+)*)", k)
    if m:
        return f"{m.group(1)}<{m.group(2).replace(' ', '')}>{(m.group(3) or '').replace(', ', '/')}"
    if "deep_prep_planes" in k:
        return "deep_prep_planes"
    for name in ("deep_reduce_all", "deep_prep", "reduce_tail_multi", "reduce_tail", "reduce_partials", "reduce_grad_loss", "epoch_tail", "ndq_pw_kernel",
                 "sample_kernel", "calib_read4", "calib_read16", "calib_write4"):
        if name in k:
            return name
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = short(row.get("Kernel_Name", "?"))
        if s is None:
            continue
        acc[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
        grid[s] = (row.get("Grid_Size"), row.get("Workgroup_Size"))
for k, d in sorted(acc.items()):
    print(k, "grid/wg =", grid.get(k))
    for name, vals in sorted(d.items()):
        vals = vals[len(vals) // 4:]          # skip warm-up dispatches
        print(f"   {name:28s} n={len(vals):5d} mean={sum(vals) / len(vals):16.1f}")
