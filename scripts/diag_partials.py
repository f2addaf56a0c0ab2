"""GPU diagnostic: per-block partial gradients of the poisson3d closure kernel, default build vs NDQ_FUSED_SAMPLING=0."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import zoo
from neurodiffeq_amd.engine import FusedSystem

name = sys.argv[1] if len(sys.argv) > 1 else "poisson3d"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 3001
def parts(flags, reps=4):
    os.environ["NDQ_JIT_FLAGS"] = flags
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    coords = system.sample(npts, seed=5)
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
print("reference build deterministic:", all((r == ref[0]).all() for r in ref), ref[0].shape)
got = parts("")
for i, g in enumerate(got):
    bad = np.argwhere(g != ref[0])
    print(f"run {i}: {len(bad)} differing entries")
    for blk, idx in bad[:40]:
        print(f"   block {blk} idx {idx}: got {g[blk, idx]!r} want {ref[0][blk, idx]!r} diff {g[blk, idx] - ref[0][blk, idx]:.3e}")
