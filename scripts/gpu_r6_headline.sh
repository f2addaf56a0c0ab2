#!/bin/bash
# Round 6: the headline closure kernel's A/B runs on one MI355X box (profiles/r06_headline_ab.md).
#   usage: scripts/gpu_r6_headline.sh TAG "flags1" "flags2" ...     ("" = the default build)
# Per flag set (NDQ_JIT_FLAGS; variants pre-built with scripts/prebuild.py): launch time of the C2 / C3 closure kernels
# (scripts/variants.py) and the parity numbers of the golden / at-size / near-convergence closure tests (their diag files).
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp NDQ_BUILD_NO_PRUNE=1
(rocminfo | grep -E "Marketing|gfx" | head -4) > $OUT/env.log 2>&1
for flags in "$@"; do
  key=$(echo "${flags:-default}" | tr -c 'A-Za-z0-9=\n' '_')
  echo "=== $key" | tee -a $OUT/times.txt
  for cfg in c2 c3:512; do
    timeout 600 python scripts/variants.py $cfg 512 "$flags" 2>&1 | tee -a $OUT/times.txt
  done
  if [[ "${PARITY:-1}" == "1" ]]; then
    NDQ_JIT_FLAGS="$flags" NDQ_DIAG_DIR=$OUT/diag_$key timeout 1500 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider \
      -k "(test_fused_closure_matches_reference_golden or test_fused_closure_matches_oracle_at_size or test_near_convergence) and (c2 or c3) and 1k" \
      > $OUT/parity_$key.log 2>&1
    tail -n 3 $OUT/parity_$key.log | cut -c1-300
  fi
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for d in sorted(glob.glob(os.path.join(out, "diag_*"))):
    key = os.path.basename(d)[5:]
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        v = json.load(open(f))
        name = os.path.basename(f)[:-5]
        if "error" in v:
            v = {k: (v["error"][k], v["bound"][k]) for k in v["error"]}
        rows.setdefault(name, {})[key] = v
json.dump(rows, open(os.path.join(out, "parity.json"), "w"), indent=1)
for name, per in rows.items():
    for key, v in per.items():
        worst = max((x[0] / x[1] if isinstance(x, (list, tuple)) else x / 1e-5) for x in v.values())
        print(f"{name:44s} {key:28s} worst error / bound = {worst:8.3f}   " +
              " ".join(f"{k}={(x[0] if isinstance(x, (list, tuple)) else x):.2e}" for k, x in v.items() if k in ("loss", "grad", "residuals", "residual")))
PY
du -sh $OUT
