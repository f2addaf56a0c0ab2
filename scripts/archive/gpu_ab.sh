#!/bin/bash
# A/B of the C2 closure kernel under different NDQ_JIT_FLAGS: usage gpu_ab.sh TAG "flags A" "flags B" ...
set -u
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for flags in "$@"; do
  NDQ_JIT_FLAGS="$flags" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - $OUT/bench_$i.json "$flags" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s]" % sys.argv[2], "ms_per_step %.5f" % d["ms_per_step"], "closure_us %.3f" % d["roofline"]["avg_launch_us"], "frac %.4f" % d["roofline"]["frac"], "final_loss", d.get("final_loss"))
except Exception as e:
    print("[%s]" % sys.argv[2], "unreadable:", e)
PY
  i=$((i+1))
done
