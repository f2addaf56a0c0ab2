mkdir -p gpurun_out/r02d
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "grouped or c4" > gpurun_out/r02d/pytest.log 2>&1; tail -3 gpurun_out/r02d/pytest.log
for f in "" "-DNDQ_GROUP_U1=2" "-DNDQ_GROUP_U1=4" "-DNDQ_GROUP_U1=2 -DNDQ_GROUP_U3=2" "-DNDQ_GROUP_U1=4 -DNDQ_GROUP_U3=2"; do
  echo "== flags: $f"; NDQ_JIT_FLAGS="$f" python scripts/bench_configs.py c4 c4:1048576 2>&1 | grep config | cut -c1-120
done
python scripts/phase_ts.py c2 2>&1 | tail -14
python scripts/phase_ts.py c2:128 2>&1 | tail -14
