mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r02f/pytest.log 2>&1; tail -3 gpurun_out/r02f/pytest.log
python scripts/bench_configs.py c2 c2:128 c2:512 c1 c3 c4 c4:1048576 c5 2>&1 | grep config | cut -c1-120
python scripts/phase_ts.py c2 2>&1 | tail -11
python scripts/phase_ts.py c2:128 2>&1 | tail -11
