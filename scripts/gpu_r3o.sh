#!/bin/bash
# round 3, visit O: WG_TR with the swizzled plane images -- C2 parity, bench A/B, per-phase cycles
set -u
OUT=gpurun_out/${1:-r3o}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "c2 and (golden or at_size or near_convergence or trajectory)" > $OUT/c2_tests.log 2>&1; echo "c2 tests rc=$?"; tail -n 3 $OUT/c2_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_tr.json 2> $OUT/bench_tr.err; echo "bench rc=$?"
NDQ_JIT_FLAGS="-DNDQ_WG_TR=0" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench (f32 weight gradients) rc=$?"
python - $OUT <<'PY'
import json, sys
for f in ("bench_tr", "bench_f32"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{f}.json").read().strip().splitlines()[-1])
        print(f, "ms_per_step", d["ms_per_step"], "closure_us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], "final_loss", d.get("final_loss"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
NDQ_JIT_FLAGS="-DNDQ_PHASE_TS" timeout 300 python scripts/phase_ts.py c2 > $OUT/ts_tr_8w.log 2>&1; echo "== tr 8-wave"; tail -n 11 $OUT/ts_tr_8w.log; grep "stage " $OUT/ts_tr_8w.log | tail -n 2
NDQ_FUSED_WIDE_MIN=1000000000 NDQ_JIT_FLAGS="-DNDQ_PHASE_TS" timeout 300 python scripts/phase_ts.py c2 > $OUT/ts_tr_4w.log 2>&1; echo "== tr 4-wave"; tail -n 11 $OUT/ts_tr_4w.log
