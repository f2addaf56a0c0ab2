#!/bin/bash
# Round 5: the BASELINE configs one by one through bench.py --config (scaling_run at world size 1), the wide single-layer
# networks, the headline under the driver protocol, then the GPU suite.   usage: scripts/gpu_r5_cfgs.sh [TAG]
set -u
TAG=${1:-r05i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in c5 c3 c4 c1; do
  steps=50; [ $cfg = c5 ] && steps=5
  for rep in 1 2; do
    timeout 300 python bench.py --config $cfg --steps $steps --warmup 5 > $OUT/${cfg}_$rep.json 2>$OUT/${cfg}_$rep.err
    python - $OUT/${cfg}_$rep.json $cfg $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]} rep {sys.argv[3]}: ms_per_step={d['ms_per_step']:.5f} value={d['value']:.4g} frac={d.get('frac_of_fp32_mfma_peak_per_gpu', 0):.3f}")
PY
  done
done
timeout 300 python scripts/wide_bench.py w16:256 w17:256 w18:256 > $OUT/wide.jsonl 2>$OUT/wide.err; cut -c1-230 $OUT/wide.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_driver_protocol.json 2>$OUT/benchq.err
python - $OUT/bench_driver_protocol.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(f"headline: value={d['value']:.4g} ms_per_step={d['ms_per_step']:.5f} closure_us={d['roofline']['avg_launch_us']:.2f}")
PY
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log | cut -c1-200
du -sh $OUT
