#!/bin/bash
# bench under torch.distributed.run with ONE rank: direct RCCL communicator vs torch.distributed all-reduce
OUT=gpurun_out/${1:-dist1}
mkdir -p "$OUT"
for mode in 1 0; do
  NDQ_RCCL_DIRECT=$mode timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$mode bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline > "$OUT/bench_direct$mode.json" 2> "$OUT/bench_direct$mode.err"
  echo "direct=$mode rc=$?"; tail -n 1 "$OUT/bench_direct$mode.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','n_gpus','final_loss')}, d['config'].get('parallelism'))"
  grep -v "amdgpu.ids\|hostname of the client" "$OUT/bench_direct$mode.err" | tail -n 5
done
