#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as a per-kernel stats table (markdown/CSV-ish).
usage: scripts/rocpd_stats.py <trace_results.db> [> profiles/xxx_kernel_stats.md]

One row per kernel VARIANT: the generated closure modules all instantiate templates whose pointwise program ``PW`` sits in an
anonymous namespace, so the demangled name alone is the same for every generated closure of one ``Cfg`` (4- / 8-wave builds,
other PDEs, other batch sizes).  Rows are therefore keyed on (name, workgroup size, grid, LDS bytes, scratch bytes, VGPRs,
AGPRs): the headline launch -- 65 536 points, its own grid and register allocation -- gets a row of its own and
``frac`` can be recomputed from this table alone."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "vgpr_count, accum_vgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                  "from kernels group by name, workgroup_x, grid_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count "
                  "order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("| kernel | calls | total_us | avg_us | min_us | max_us | % | vgpr | agpr | lds_B | scratch_B | grid_x | wg_x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, calls, tot, avg, mn, mx, vg, ag, lds, scr, gx, wx in rows:
    nm = name if len(name) < 110 else name[:107] + "..."
    print(f"| {nm} | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} | "
          f"{vg} | {ag} | {lds} | {scr} | {gx} | {wx} |")
