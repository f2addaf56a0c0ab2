#!/bin/bash
# round 4: first timings of the wide single-hidden-layer kernels (csrc/ndq_wide.h)
mkdir -p gpurun_out
python scripts/wide_bench.py w16:256 w17:256 > gpurun_out/r04a_wide.jsonl 2> gpurun_out/r04a_wide.err
NDQ_JIT_FLAGS="-DNDQ_WIDE_THREADS=512" python scripts/wide_bench.py w16:256 >> gpurun_out/r04a_wide.jsonl 2>> gpurun_out/r04a_wide.err
python scripts/wide_bench.py w16:1024 w16:64 >> gpurun_out/r04a_wide.jsonl 2>> gpurun_out/r04a_wide.err
cat gpurun_out/r04a_wide.jsonl; tail -5 gpurun_out/r04a_wide.err
