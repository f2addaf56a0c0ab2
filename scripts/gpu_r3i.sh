#!/bin/bash
# round 3, visit I: C1 with four tile slots per workgroup (two waves per SIMD), with and without pull mode
set -u
OUT=gpurun_out/r3i; mkdir -p $OUT
for g in 2 4; do for pull in 1 0; do
  flags=""; [ $g = 4 ] && flags="-DNDQ_MULTI_G2=4"
  echo "G=$g pull=$pull"
  NDQ_JIT_FLAGS="$flags" NDQ_FIT_PULL=$pull timeout 600 python scripts/config_fit.py c1 2> $OUT/c1_g${g}_pull$pull.err | tee $OUT/c1_g${g}_pull$pull.json
done; done
tail -n 3 $OUT/c1_g4_pull1.err
