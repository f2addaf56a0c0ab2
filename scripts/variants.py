#!/usr/bin/env python
"""Time the closure kernel of a config built with different NDQ_JIT_FLAGS (one child process per variant).
usage: python scripts/variants.py cfg[:size] threads "flags1" "flags2" ...   (ABLATE_BUILD_ONLY=1: just build them)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate  # noqa: E402

arg, threads = sys.argv[1], sys.argv[2]
base = None
for flags in sys.argv[3:]:
    env = dict(os.environ, NDQ_JIT_FLAGS=flags, NDQ_BUILD_NO_PRUNE="1")
    if os.environ.get("ABLATE_BUILD_ONLY"):
        code = ("import sys; sys.path.insert(0, %r); import torch; from tests import configs; from neurodiffeq_amd import codegen; "
                "from neurodiffeq_amd.engine import trace_system; torch.manual_seed(0); a = sys.argv[1]; "
                "cfg = configs.make(a.split(':')[0], 8); p, d = trace_system(cfg['nets'], cfg['conds'], cfg['pde'], configs.n_coords(cfg), configs.func_val(cfg)); "
                "print(codegen.build_fused(p, d[0])); print(codegen.build_fused(p, d[0], threads=512))" % ablate.ROOT)
        subprocess.run([sys.executable, "-c", code, arg], env=env, check=True)
        continue
    out = subprocess.run([sys.executable, "-c", ablate.CHILD, arg, threads], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(flags, "FAILED", out.stderr[-600:])
        continue
    _, thr, n, us = line[0].split()
    us = float(us)
    base = base or us
    print(f"{flags or '(default)':44s} {us:7.2f} us   ({us - base:+6.2f} us; {thr} threads, {n} points)", flush=True)
