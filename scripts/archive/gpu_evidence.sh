# round-2 evidence: bench line, kernel trace over the bench (all configs), PMC passes for C2 (headline), C3, C4, C5
TAG=${1:-evidence}
mkdir -p gpurun_out/${TAG}
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > gpurun_out/${TAG}/env.log 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}/bench.json 2> gpurun_out/${TAG}/bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/${TAG}/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/${TAG}/bench_driver_protocol.json 2>/dev/null; cut -c1-600 gpurun_out/${TAG}/bench_driver_protocol.json
export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/${TAG}/prof" -o trace -- python "$REPO/bench.py" --steps 2000 --warmup 200 --no-cpu-baseline > "$REPO/gpurun_out/${TAG}/prof_bench.json" 2> "$REPO/gpurun_out/${TAG}/prof.err"); echo "rocprof rc=$?"
find gpurun_out/${TAG}/prof -name "*kernel_trace.csv" -size +20M -delete
bash scripts/gpu_pmc.sh ${TAG}_pmc_c2 > gpurun_out/${TAG}/pmc_c2.log 2>&1; tail -3 gpurun_out/${TAG}/pmc_c2.log
for c in c3 c4 c5; do bash scripts/gpu_pmc.sh ${TAG}_pmc_$c python scripts/bench_configs.py $c > gpurun_out/${TAG}/pmc_$c.log 2>&1; tail -2 gpurun_out/${TAG}/pmc_$c.log; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5000 --warmup 500 --no-configs > gpurun_out/${TAG}/bench_dist1.json 2> gpurun_out/${TAG}/bench_dist1.err; echo "dist rc=$?"; tail -n 1 gpurun_out/${TAG}/bench_dist1.json | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 50 --warmup 10 --config c3 --scaling strong > gpurun_out/${TAG}/bench_c3_strong_1rank.json 2> gpurun_out/${TAG}/bench_c3.err; echo "c3 strong rc=$?"; tail -n 1 gpurun_out/${TAG}/bench_c3_strong_1rank.json | cut -c1-300
du -sh gpurun_out/${TAG}*
