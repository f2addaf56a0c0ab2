#!/bin/bash
# The 1 -> 8 GPU scaling table of DESIGN.md section 7, one command on an 8-GPU MI355X node:
#   weak scaling of the headline (C2, 65 536 points per GPU) and STRONG scaling of the two configs that have work to share
#   (C3 262 144 points, C5 1 048 576 points) at 1 / 2 / 4 / 8 ranks, one rank per GPU over RCCL / xGMI
#   (torch.distributed.run, 127.0.0.1 rendezvous).  Every run prints bench.py's JSON line; the table is computed from
#   those lines.  usage: scripts/scale.sh [outdir]       env: GPUS="1 2 4 8"  STEPS / WARMUP per config below
# Nothing here needs more than the driver's own SCALE run does; it exists so that the curve is one command for a human.
set -u
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
GPUS=${GPUS:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
port=29600
run() {  # name scaling config steps warmup
  local name=$1 scaling=$2 cfg=$3 steps=$4 warm=$5
  for n in $GPUS; do
    port=$((port + 1))
    if [ "$n" = "1" ] && [ "$cfg" = "c2" ]; then
      timeout 900 python bench.py --gpus 1 --steps $steps --warmup $warm --no-cpu-baseline --no-configs --no-traffic \
        > "$OUT/${name}_$n.json" 2> "$OUT/${name}_$n.err"
    else
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps $steps --warmup $warm --config $cfg --scaling $scaling --no-cpu-baseline --no-configs --no-traffic \
        > "$OUT/${name}_$n.json" 2> "$OUT/${name}_$n.err"
    fi
    echo "$name n=$n rc=$?"
  done
}
run c2_weak weak c2 2000 200
run c3_strong strong c3 200 20
run c5_strong strong c5 20 3
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for name in ("c2_weak", "c3_strong", "c5_strong"):
    base = None
    for n in (1, 2, 4, 8):
        f = os.path.join(out, f"{name}_{n}.json")
        if not os.path.exists(f):
            continue
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not lines:
            continue
        d = json.loads(lines[-1])
        base = base or d["value"]
        rows.append((name, n, d["value"], d["ms_per_step"], d["value"] / base, d["value"] / base / n,
                     d.get("config", {}).get("allreduce", ""), d.get("allreduce_flag_timeouts", "")))
with open(os.path.join(out, "scaling_table.md"), "w") as fh:
    fh.write("| run | GPUs | points/s | ms/step | speed-up vs 1 GPU | efficiency | all-reduce | flag timeouts |\n|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| {r[0]} | {r[1]} | {r[2]:.4g} | {r[3]:.4f} | {r[4]:.2f} | {r[5]:.2f} | {r[6]} | {r[7]} |\n")
print(open(os.path.join(out, "scaling_table.md")).read())
PY
