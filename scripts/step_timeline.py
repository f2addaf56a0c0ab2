#!/usr/bin/env python
"""Where a headline training step spends its time on the GPU, from a rocprofv3 kernel trace (rocpd sqlite) of bench.py:
median duration of the closure launch, of the sums / tail launch, and of the two idle gaps between them, over every
steady-state step (closure -> tail -> closure -> ...) of the trace.
usage: scripts/step_timeline.py <trace_results.db> [closure-name-fragment] [tail-name-fragment] > profiles/xxx_step_timeline.json"""
import json
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
closure_key = sys.argv[2] if len(sys.argv) > 2 else "fused_closure_kernel<ndq::Cfg<2, 1, 5u, 2, 2"
tail_key = sys.argv[3] if len(sys.argv) > 3 else "reduce_tail_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
t0 = "start" if "start" in cols else "start_timestamp"
t1 = "end" if "end" in cols else "end_timestamp"
rows = db.execute(f'select name, "{t0}", "{t1}" from kernels order by "{t0}"').fetchall()
steps = []
i = 0
while i + 2 < len(rows):
    a, b, c = rows[i], rows[i + 1], rows[i + 2]
    if closure_key in a[0] and tail_key in b[0] and closure_key in c[0]:
        steps.append((a[2] - a[1], b[1] - a[2], b[2] - b[1], c[1] - b[2], c[1] - a[1]))
        i += 2
    else:
        i += 1
# a window boundary (host synchronisation) shows as a long second gap: keep the steps issued back to back
steady = [s for s in steps if s[3] < 20000] or steps


def med(k):
    return round(statistics.median(s[k] for s in steady) / 1e3, 3) if steady else None


out = {"steps_in_trace": len(steps), "steady_state_steps": len(steady),
       "closure_us": med(0), "gap_closure_to_tail_us": med(1), "tail_us": med(2), "gap_tail_to_closure_us": med(3),
       "step_us": med(4),
       "note": "medians over back-to-back steps (closure launch -> sums/tail launch -> next closure launch) of the traced "
               "bench run; gaps = end of one kernel to start of the next on the same stream (dispatch of a dependent kernel)"}
print(json.dumps(out, indent=1))
