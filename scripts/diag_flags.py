"""GPU diagnostic: closure-kernel partial gradients of a zoo system under several NDQ_JIT_FLAGS builds, against the
NDQ_FUSED_SAMPLING=0 build.  With --prebuild only compiles (no GPU needed)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import zoo

VARIANTS = [v for v in os.environ.get("DIAG_VARIANTS", "").split(";")]
name = os.environ.get("DIAG_SYSTEM", "poisson3d")
prebuild = "--prebuild" in sys.argv

def parts(flags, reps=4):
    os.environ["NDQ_JIT_FLAGS"] = flags
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    if prebuild:
        from neurodiffeq_amd import codegen
        from neurodiffeq_amd.engine import trace_system
        program, descs = trace_system(nets, conds, pde, system.n_coords)
        codegen.build(program); print(flags, "->", codegen.build_fused(program, descs[0]), flush=True)
        return None
    from neurodiffeq_amd.engine import FusedSystem
    coords = system.sample(3001, seed=5)
    for net in nets:
        net.to("cuda")
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=True)
    out = []
    for _ in range(reps):
        b, n = fs.step([c.float() for c in coords], train=True, slot=0)
        torch.cuda.synchronize()
        out.append(b["fused_partials"].cpu().numpy().copy())
    return out

ref = parts("-DNDQ_FUSED_SAMPLING=0")
for v in VARIANTS:
    got = parts(v)
    if got is None:
        continue
    det = all((g == got[0]).all() for g in got)
    bad = sorted({int(i) for g in got for i in np.argwhere(g != ref[0])[:, 1]})
    err = max(float(np.abs(g - ref[0]).max()) for g in got)
    print(f"[{v}] deterministic={det} max|diff vs ref|={err:.3e} differing param idx={bad[:12]}{'...' if len(bad) > 12 else ''}", flush=True)
