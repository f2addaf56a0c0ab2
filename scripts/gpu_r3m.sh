#!/bin/bash
# round 3, visit M: hidden-layer weight gradients through transposing LDS reads (Cfg::WG_TR) -- lane map of
# ds_read_b64_tr_b16, parity of the C2 closure, bench A/B against the exact-f32 weight-gradient route
set -u
OUT=gpurun_out/r3m; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 scripts/ubench_tr16.hip -o /tmp/ubench_tr16 2>/dev/null && /tmp/ubench_tr16 > $OUT/tr16.log 2>&1; echo "tr16 rc=$?"; tail -n 12 $OUT/tr16.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "c2 and (golden or at_size or near_convergence or trajectory)" > $OUT/c2_tests.log 2>&1; echo "c2 tests rc=$?"; tail -n 15 $OUT/c2_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_tr.json 2> $OUT/bench_tr.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_tr.json
NDQ_JIT_FLAGS="-DNDQ_WG_TR=0" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic --no-cold-start > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench (f32 weight gradients) rc=$?"; cut -c1-400 $OUT/bench_f32.json
python - <<'PY'
import json
for f in ("bench_tr", "bench_f32"):
    try:
        d = json.loads(open(f"gpurun_out/r3m/{f}.json").read().strip().splitlines()[-1])
        print(f, "ms_per_step", d["ms_per_step"], "closure_us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], "final_loss", d.get("final_loss"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
