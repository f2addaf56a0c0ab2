#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as a per-kernel stats table (markdown/CSV-ish).
usage: scripts/rocpd_stats.py <trace_results.db> [> profiles/xxx_kernel_stats.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("| kernel | calls | total_us | avg_us | min_us | max_us | % | vgpr | agpr | lds_B | scratch_B | grid_x | wg_x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, calls, tot, avg, mn, mx, vg, ag, lds, scr, gx, wx in rows:
    nm = name if len(name) < 110 else name[:107] + "..."
    print(f"| {nm} | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} | "
          f"{vg} | {ag} | {lds} | {scr} | {gx} | {wx} |")
