#!/bin/bash
# round 3, visit J: training-only epochs on the plain training kernel against the combined train + validation kernel
set -u
OUT=gpurun_out/r3j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -p no:cacheprovider > $OUT/fit_tests.log 2>&1; echo "fit tests rc=$?"; tail -n 3 $OUT/fit_tests.log
for rep in 1 2; do for tv in 0 1; do
  if [ $tv = 1 ]; then export NDQ_TV_ALWAYS=1; else unset NDQ_TV_ALWAYS; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_tv${tv}_$rep.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$OUT/bench_tv${tv}_$rep.json').read().strip().splitlines()[-1]); print('combined kernel always' if $tv else 'plain kernel', d['value'], d['ms_per_step'], d['in_fit']['ms_per_step'])"
done; done
unset NDQ_TV_ALWAYS
timeout 600 python scripts/config_fit.py c3 c4 | tee $OUT/configs_plain.json
NDQ_TV_ALWAYS=1 timeout 600 python scripts/config_fit.py c3 c4 | tee $OUT/configs_tv.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --cold-start 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('cold_start'))"
