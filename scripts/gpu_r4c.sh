#!/bin/bash
# round 4: deep wide kernels after prefetch / 16-byte loads / parallel reductions: tests, timings, kernel trace
mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_wide.py -q -p no:cacheprovider -k "128 or 256 or 100-1-tanh-lap-2 or 512-3 or 65-1-sin or 80-2 or 144 or 96-1-aptx or 72-1 or w18 or w19" 2>&1 | tail -5
python scripts/wide_bench.py w18:256 w19:256 > gpurun_out/r04c/deep.jsonl 2> gpurun_out/r04c/deep.err; cat gpurun_out/r04c/deep.jsonl
bash scripts/gpu_r4b.sh w18:256 2>&1 | tail -14
