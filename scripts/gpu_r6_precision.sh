#!/bin/bash
# Round 6: which product count drives the error of the GENERIC adjoint kernels (ndq_mlp_jet_bwd with random adjoint seeds)?
# libndq.so variants built with NDQ_LIB_FLAGS="-DNDQ_HBAR_NPROD=a -DNDQ_WG_NPROD=b" (neurodiffeq_amd/_build.py), the kernel
# tests of tests/test_gpu_parity.py against the fp64 jet oracle.   usage: scripts/gpu_r6_precision.sh TAG "a b" "a b" ...
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  set -- $v
  flags="-DNDQ_HBAR_NPROD=$1 -DNDQ_WG_NPROD=$2"
  key="hbar$1_wg$2"
  NDQ_LIB_FLAGS="$flags" NDQ_DIAG_DIR=$OUT/diag_$key timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider \
    -k "test_mlp_jet_bwd_matches_jet_oracle or test_bwd_is_linear_in_gbar" > $OUT/pytest_$key.log 2>&1
  echo "$key: $(tail -n 1 $OUT/pytest_$key.log)"
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
table = {}
for d in sorted(glob.glob(os.path.join(out, "diag_*"))):
    key = os.path.basename(d)[5:]
    worst = {}
    for f in glob.glob(os.path.join(d, "bwd_*.json")):
        v = json.load(open(f))
        name = os.path.basename(f)[4:-5]
        worst[name] = max(x for x in v.values() if isinstance(x, (int, float)))
    table[key] = worst
json.dump(table, open(os.path.join(out, "precision_map.json"), "w"), indent=1)
names = sorted({n for w in table.values() for n in w}, key=lambda s: (s.rsplit("_", 1)[0], int(s.rsplit("_", 1)[1])))
print(f"{'case':18s}" + "".join(f"{k:>14s}" for k in table))
for n in names:
    if n.endswith(("_1000", "_4099")):
        print(f"{n:18s}" + "".join(f"{table[k].get(n, float('nan')):14.2e}" for k in table))
PY
