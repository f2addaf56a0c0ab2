#!/bin/bash
# round 3, visit K: C2 with 512 closure workgroups (two 8-wave workgroups per CU, one tile per wave) against 256
set -u
OUT=gpurun_out/r3k; mkdir -p $OUT
for rep in 1 2; do for cap in 256 512; do
  flags=""; [ $cap = 512 ] && flags="-DNDQ_MAX_BLOCKS=512"
  NDQ_JIT_FLAGS="$flags" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_cap${cap}_$rep.json 2> $OUT/bench_cap${cap}_$rep.err
  python -c "
import json; d=json.loads(open('$OUT/bench_cap${cap}_$rep.json').read().strip().splitlines()[-1]); print('cap $cap', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done; done
tail -n 2 $OUT/bench_cap512_1.err
