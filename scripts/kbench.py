#!/usr/bin/env python
"""Micro-benchmark of the MLP kernels of one or more libndq builds (tuning aid, GPU only).
usage: scripts/kbench.py lib1.so [lib2.so ...]   -> one JSON line per library (C2 shapes, N = 65536)"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurodiffeq_amd._lib import MlpDesc  # noqa: E402
from oracle import jet_ref as J  # noqa: E402

N = int(os.environ.get("KBENCH_N", 65536))
CFGS = {"c2": ((2, 32, 32, 1), "tanh", MlpDesc(2, 1, 5, 32, 2, 0, 1, 0), [(), (0,), (1,), (0, 0), (1, 1)], 10688)}


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    rng = np.random.default_rng(0)
    for path in sys.argv[1:]:
        L = ctypes.CDLL(os.path.abspath(path))
        for name, (dims, act, d, streams, fflop) in CFGS.items():
            if not L.ndq_mlp_supported(ctypes.byref(d)):
                continue
            P = L.ndq_mlp_num_params(ctypes.byref(d))
            parts = []
            for a, b in zip(dims[:-1], dims[1:]):
                k = 1 / np.sqrt(a)
                parts += [rng.uniform(-k, k, a * b), rng.uniform(-k, k, b)]
            flat = np.concatenate(parts).astype(np.float32)
            coords = rng.uniform(0, 1, (dims[0], N)).astype(np.float32)
            gbar = (rng.standard_normal((len(streams), N)) * float(os.environ.get("KBENCH_GSCALE", "1"))).astype(np.float32)
            c, p, g = torch.from_numpy(coords).cuda(), torch.from_numpy(flat).cuda(), torch.from_numpy(gbar).cuda()
            jets = torch.zeros(len(streams), N, device="cuda")
            nb = L.ndq_mlp_bwd_blocks(ctypes.byref(d), N)
            part = torch.zeros(nb, P, device="cuda"); out = torch.zeros(P, device="cuda")
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            vp = ctypes.c_void_p
            fwd = lambda: L.ndq_mlp_jet_fwd(ctypes.byref(d), vp(c.data_ptr()), N, N, vp(p.data_ptr()), vp(jets.data_ptr()), N, st)
            bwd = lambda: L.ndq_mlp_jet_bwd(ctypes.byref(d), vp(c.data_ptr()), N, N, vp(p.data_ptr()), vp(g.data_ptr()), N, vp(part.data_ptr()), st)
            red = lambda: L.ndq_reduce_partials(vp(part.data_ptr()), nb, P, vp(out.data_ptr()), 0, ctypes.c_float(1.0), st)
            assert fwd() == 0 and bwd() == 0 and red() == 0
            torch.cuda.synchronize()
            m = min(N, 8192)
            wj = J.mlp_jets(flat.astype(np.float64), dims, act, list(coords[:, :m].astype(np.float64)), streams)
            ferr = max(rel(jets[s, :m].cpu().numpy(), wj[mm][:, 0]) for s, mm in enumerate(streams))
            wg = J.mlp_jets_vjp(flat.astype(np.float64), dims, act, list(coords.astype(np.float64)),
                                {mm: gbar[s].astype(np.float64)[:, None] for s, mm in enumerate(streams)})
            berr = rel(out.cpu().numpy(), wg)
            tf, tb, tr = timeit(fwd), timeit(bwd), timeit(red)
            print(json.dumps(dict(lib=os.path.basename(path), cfg=name, n=N, fwd_us=round(tf, 2), bwd_us=round(tb, 2),
                                  red_us=round(tr, 2), bwd_blocks=nb, fwd_tflops=round(fflop * N / tf / 1e6, 1),
                                  bwd_tflops=round(2 * fflop * N / tb / 1e6, 1), fwd_err=ferr, bwd_err=berr)), flush=True)


if __name__ == "__main__":
    main()
