mkdir -p gpurun_out/r02j
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "closure_based or custom_loss or falls_back" > gpurun_out/r02j/pytest.log 2>&1; tail -5 gpurun_out/r02j/pytest.log
echo "== 8-wave build, 65536 points"; python scripts/ablate.py c2 512 2>&1 | tail -10
echo "== 4-wave build, 16384 points"; python scripts/ablate.py c2:128 256 2>&1 | tail -10
echo "== 4-wave build, 65536 points"; NDQ_FUSED_WIDE=0 python scripts/ablate.py c2 256 2>&1 | tail -10
