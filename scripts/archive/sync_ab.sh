#!/bin/bash
# A/B of the host-side wait mode under the driver's bench protocol (20-step windows, each closed by a synchronisation):
# how much of a window is the wake-up of the waiting host thread?   usage: scripts/sync_ab.sh [outdir]
OUT=${1:-gpurun_out/sync_ab}; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-traffic 2>/dev/null | grep "^{" | tail -n 1 > $OUT/$name.json
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:28s} value={d['value']/1e9:.3f} G  ms_per_step={d['ms_per_step']*1e3:.2f} us  window_min={d['timing']['window_ms_min']*1e3:.1f} us  default_gen_cuda={d.get('default_generator_cuda',{}).get('value',0)/1e9:.3f} G")
except Exception as e:
    print(sys.argv[2], "no json", e)
PY
}
run default NDQ_DUMMY=1
run hsa_interrupt_0 HSA_ENABLE_INTERRUPT=0
run active_wait_1000 ROC_ACTIVE_WAIT_TIMEOUT=1000
run active_wait_100 ROC_ACTIVE_WAIT_TIMEOUT=100
run both HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000
run default_again NDQ_DUMMY=1
python - <<'PY'
import time, torch, ctypes
x = torch.zeros(1024, device="cuda")
def lat(n=2000):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); x.add_(1.0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts)//2] * 1e6, ts[len(ts)//10] * 1e6
print("launch + synchronize of one tiny kernel, median / p10 us:", lat())
hip = ctypes.CDLL("libamdhip64.so")
print("hipSetDeviceFlags(spin) rc", hip.hipSetDeviceFlags(1))
print("after hipDeviceScheduleSpin:", lat())
PY
