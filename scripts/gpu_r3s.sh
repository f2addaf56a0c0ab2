#!/bin/bash
# round 3, visit S: H = 64 weight gradients through transposing reads (Cfg::WG_TR64) -- C3 / C5 parity and timing
set -u
OUT=gpurun_out/${1:-r3s}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "(c3 or c5 or c1) and (golden or at_size or near_convergence or trajectory or reproducible)" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -n 6 $OUT/tests.log
for cfg in c3 c5 c1; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "bench $cfg rc=$?"
  python - $OUT/bench_$cfg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms_per_step %.5f" % d["ms_per_step"], "value %.4g" % d["value"], {k: v for k, v in d.get("roofline", {}).items() if k in ("avg_launch_us", "frac", "kernel")})
except Exception as e:
    print("unreadable:", e)
PY
done
NDQ_JIT_FLAGS="-DNDQ_WG_TR=0" timeout 300 python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_c3_f32.json 2> $OUT/bench_c3_f32.err
python - $OUT/bench_c3_f32.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("c3 without WG_TR64: ms_per_step %.5f" % d["ms_per_step"], "value %.4g" % d["value"])
except Exception as e:
    print("unreadable:", e)
PY
