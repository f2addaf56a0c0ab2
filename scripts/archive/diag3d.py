"""GPU diagnostic: per-parameter gradient error of the single-launch closure kernel on the 3-coordinate zoo systems,
repeated to check determinism.  Usage: python scripts/diag3d.py [names...]"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import zoo
from oracle import autograd_ref as R
from neurodiffeq_amd.engine import FusedSystem

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

names = sys.argv[1:] or ["poisson3d", "hessian3d", "shell", "helmholtz_xy"]
for name in names:
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(3001, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.to("cuda")
    for mode in ("1k", "3k"):
        fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=(mode == "1k"))
        if mode == "1k" and fs.fusedk is None:
            continue
        prev = None
        for rep in range(3):
            b, n = fs.step([c.float() for c in coords], train=True, slot=0, want_funcs=True, want_resid=True)
            torch.cuda.synchronize()
            g = np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat])
            out = {"all": rel(g, want_grad), "same_as_prev": None if prev is None else bool((g == prev).all())}
            off = 0
            for i, p in enumerate(fs.flat[0].params):
                k = p.numel()
                out[f"p{i}{tuple(p.shape)}"] = rel(g[off:off + k], want_grad[off:off + k])
                off += k
            prev = g
            print(name, mode, rep, json.dumps(out), flush=True)
        if mode == "1k":
            w = want_grad[:96].reshape(32, 3); gg = g[:96].reshape(32, 3)
            print("  W1 col err:", [rel(gg[:, a], w[:, a]) for a in range(3)] if system.n_coords == 3 else None)
            bad = np.argsort(-np.abs(g - want_grad))[:8]
            print("  worst idx:", bad.tolist(), (g - want_grad)[bad].tolist(), want_grad[bad].tolist())
