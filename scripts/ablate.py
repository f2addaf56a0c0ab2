#!/usr/bin/env python
"""Marginal cost of the pieces of the closure kernel, in situ: the C2 closure kernel built with one piece switched off
at a time (csrc/ndq_mlp.h: NDQ_ABL; results are wrong by design) and timed back to back.
usage: python scripts/ablate.py [config[:size]] [threads]   -- builds every variant (hipcc) unless pre-built"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("baseline", 0), ("no weight-grad GEMMs", 1), ("no hbar GEMM", 2), ("no weight-grad, no hbar GEMM", 1 | 2),
            ("no forward GEMM", 4), ("no act_backward", 8), ("no operand splitting", 32),
            ("no GEMMs at all", 1 | 2 | 4), ("no GEMMs, no split, no act_backward", 1 | 2 | 4 | 8 | 32)]
if os.environ.get("ABLATE_F32"):      # the exact-f32 weight-gradient route (NDQ_WG_TR=0) has its LDS transposes as a piece of its own
    VARIANTS.insert(2, ("no LDS transposes", 16))
CHILD = r"""
import sys, torch, ctypes
sys.path.insert(0, %r)
from tests import configs
from neurodiffeq_amd.engine import FusedSystem, _ptr, _c_vp
arg, threads = sys.argv[1], int(sys.argv[2])
name, size = (arg.split(":")[0], int(arg.split(":")[1])) if ":" in arg else (arg, None)
torch.manual_seed(0)
cfg = configs.make(name, size)
for net in cfg["nets"]:
    net.to("cuda")
import neurodiffeq_amd.engine as E
system = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], configs.n_coords(cfg), "cuda", compute_func_val=configs.func_val(cfg))
system._self_check = False
if threads == 256:
    system.fusedk_wide = False
ex = cfg["gen"].get_examples()
batch = [c.detach().cuda() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
b, n = system.upload(batch)
fk = b["fusedk"]
fp = system.flat[0]
stream = _c_vp(torch.cuda.current_stream().cuda_stream)
def launch():
    fk.lib.ndq_fused_launch(system._coord_ptr(b, 0), b["ld"], n, _ptr(fp.flat), _ptr(b["fused_partials"]),
                            _ptr(b["fused_loss_partials"]), None, None, b["ld"], 1.0 / n, 1, stream)
for _ in range(3000):
    launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(3000):
    launch()
e1.record()
torch.cuda.synchronize()
print("RESULT", fk.threads, n, e0.elapsed_time(e1) / 3000 * 1e3)
""" % ROOT

if __name__ == "__main__":
    arg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    threads = sys.argv[2] if len(sys.argv) > 2 else "512"
    base = None
    for label, bits in VARIANTS:
        env = dict(os.environ, NDQ_JIT_FLAGS=f"-DNDQ_ABL={bits}" if bits else "", NDQ_BUILD_NO_PRUNE="1")
        if os.environ.get("ABLATE_BUILD_ONLY"):
            code = ("import sys; sys.path.insert(0, %r); import torch; from tests import configs; from neurodiffeq_amd import codegen; "
                    "from neurodiffeq_amd.engine import trace_system; torch.manual_seed(0); a = sys.argv[1]; "
                    "cfg = configs.make(a.split(':')[0], 8); p, d = trace_system(cfg['nets'], cfg['conds'], cfg['pde'], configs.n_coords(cfg), configs.func_val(cfg)); "
                    "print(codegen.build_fused(p, d[0])); print(codegen.build_fused(p, d[0], threads=512))" % ROOT)
            subprocess.run([sys.executable, "-c", code, arg], env=env, check=True)
            continue
        out = subprocess.run([sys.executable, "-c", CHILD, arg, threads], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(label, "FAILED", out.stderr[-400:])
            continue
        _, thr, n, us = line[0].split()
        us = float(us)
        base = base or us
        print(f"{label:40s} {us:7.2f} us   ({us - base:+6.2f} us vs baseline; {thr} threads, {n} points)", flush=True)
