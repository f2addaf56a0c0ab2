mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r02e/pytest.log 2>&1; tail -3 gpurun_out/r02e/pytest.log
for f in "" "-DNDQ_KEEP_H=0"; do
  echo "== flags: $f"; NDQ_JIT_FLAGS="$f" python scripts/bench_configs.py c2 c2:128 c2:512 2>&1 | grep config | cut -c1-120
done
python scripts/bench_configs.py c1 c3 c4 c4:1048576 2>&1 | grep config | cut -c1-120
python scripts/phase_ts.py c2 2>&1 | tail -12
python scripts/phase_ts.py c2:128 2>&1 | tail -12
