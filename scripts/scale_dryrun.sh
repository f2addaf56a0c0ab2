#!/bin/bash
# Dry run of the driver's multi-GPU bench protocol on ONE GPU: bench.py --gpus {2,4,8} under
# torch.distributed.run exactly as the driver launches it, except that every rank uses device 0 (NDQ_BENCH_DEVICE=0) and the
# process group is gloo (NDQ_BENCH_BACKEND=gloo) -- RCCL cannot put two ranks on one device.  The one-shot all-reduce (HIP-IPC
# inboxes, csrc/ndq_oneshot.h) DOES run between the processes, so the exchange, the sharding, the rendezvous and the JSON line
# are exercised; the numbers are NOT scaling figures (N ranks time-share one GPU).
#   usage: scripts/scale_dryrun.sh [outdir]     env: GPUS="2 4 8"
set -u
OUT=${1:-gpurun_out/scale_dryrun}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
port=29700
for n in ${GPUS:-2 4 8}; do
  port=$((port + 1))
  NDQ_BENCH_DEVICE=0 NDQ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic \
    > "$OUT/dryrun_$n.out" 2> "$OUT/dryrun_$n.err"
  rc=$?
  grep "^{" "$OUT/dryrun_$n.out" | tail -n 1 > "$OUT/scale_dryrun_$n.json"
  python - "$OUT/scale_dryrun_$n.json" $n $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"n={sys.argv[2]} rc={sys.argv[3]} value={d['value']:.4g} ms_per_step={d['ms_per_step']:.4f} allreduce={d['config'].get('allreduce')} "
          f"flag_timeouts={d.get('allreduce_flag_timeouts')} scaling={d.get('scaling')} n_gpus={d.get('n_gpus')}")
except Exception as e:
    print(f"n={sys.argv[2]} rc={sys.argv[3]} NO JSON LINE ({e})")
PY
done
# ... and the strong-scaling form of a big config (bench.py --config c3 --scaling strong: 262 144 points cut into 8 shards)
port=$((port + 1))
NDQ_BENCH_DEVICE=0 NDQ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
  --master-addr 127.0.0.1 --master-port $port bench.py --gpus 8 --config c3 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline \
  > "$OUT/dryrun_c3_strong_8.out" 2> "$OUT/dryrun_c3_strong_8.err"
echo "c3 strong 8 ranks rc=$?"; grep "^{" "$OUT/dryrun_c3_strong_8.out" | tail -n 1 | tee "$OUT/scale_dryrun_c3_strong_8.json" | cut -c1-400
