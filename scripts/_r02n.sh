mkdir -p gpurun_out/r02n
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r02n/pytest.log 2>&1; tail -4 gpurun_out/r02n/pytest.log
python scripts/bench_configs.py c2 c2:128 c2:512 c1 c3 c4 2>&1 | grep config | cut -c1-120
timeout 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r02n/bench_short.json 2>/dev/null; python - <<'P'
import json
d=json.loads(open("gpurun_out/r02n/bench_short.json").read().strip().splitlines()[-1])
print("step us", d["ms_per_step"]*1e3, d["timing"], "kernel", d["roofline"]["avg_launch_us"], d["roofline"]["in_situ_us"], "frac", d["roofline"]["frac"])
P
bash scripts/gpu_pmc.sh r02n_pmc_c2 > gpurun_out/r02n/pmc_c2.log 2>&1; grep -A22 "fused_closure<2,1,5u,2,2" gpurun_out/r02n_pmc_c2/summary.txt | grep "FETCH_SIZE\|WRITE_SIZE" | head -2
