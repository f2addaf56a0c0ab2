#!/usr/bin/env python
"""profiles/traffic_c2.json from a scripts/gpu_pmc.sh summary: HBM-side bytes per launch of the headline kernels,
corrected with the calibration kernels of the SAME run (scripts/pmc_calib.hip: known byte counts in this package's
access patterns).  usage: scripts/traffic_from_pmc.py gpurun_out/<tag>/summary.txt <tag> > profiles/traffic_c2.json"""
import json
import re
import sys

text = open(sys.argv[1]).read()
tag = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
blocks = {}
cur = None
for line in text.splitlines():
    if not line.startswith(" "):
        cur = line.split(" grid/wg")[0].strip()
        blocks[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", line)
        if m:
            blocks[cur][m.group(1)] = float(m.group(3))
calib_bytes = 512 * 1024 * 1024
fx = {k: calib_bytes / (blocks[k]["FETCH_SIZE"] * 1024) for k in ("calib_read4", "calib_read16") if k in blocks}
wx = calib_bytes / (blocks["calib_write4"]["WRITE_SIZE"] * 1024) if "calib_write4" in blocks else 1.0
out = {"source": f"profiles/{tag}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean per dispatch, "
                 "counters in KiB; corrected with calibration kernels of the same run: a 512 MiB streaming read reports "
                 f"FETCH_SIZE x {1 / fx.get('calib_read4', 2):.3f} at 4 B/lane and x {1 / fx.get('calib_read16', 2):.3f} at 16 B/lane "
                 f"(the guide's 1/2), a 512 MiB streaming write reports WRITE_SIZE x {1 / wx:.3f})",
       "fetch_correction": fx.get("calib_read4", 2.0), "write_correction": wx, "kernels": {}}
for name, key, algo in (("fused_closure", "fused_closure<2,1,5u,2,2", 65536 * 8 + 256 * 1185 * 4 + 256 * 4),
                        ("pointwise", "ndq_pw_kernel", 65536 * 40)):
    k = next((b for b in blocks if b.startswith(key)), None)
    if k is None or "FETCH_SIZE" not in blocks[k]:
        continue
    f, w = blocks[k]["FETCH_SIZE"], blocks[k]["WRITE_SIZE"]
    out["kernels"][name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                            "hbm_bytes": f * 1024 * out["fetch_correction"] + w * 1024 * wx, "algorithmic_bytes": algo}
print(json.dumps(out, indent=1))
