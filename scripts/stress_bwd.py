"""Determinism stress of the standalone adjoint kernel: many launches on identical inputs must be bit-identical."""
import sys, os, zlib, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_parity as T
from neurodiffeq_amd import _lib
L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name in ("c2", "c2full", "c3", "c1"):
    dims, act, _, _, streams = T.ARCH[name]
    for n in (4099, 8195, 65536):
        rng = np.random.default_rng(7)
        flat = T._params(name, rng)
        coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
        gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
        ld = (n + 63) // 64 * 64
        c = torch.zeros(dims[0], ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
        g = torch.zeros(len(streams), dims[-1], ld, device="cuda"); g[:, :, :n] = torch.from_numpy(gbar)
        p = torch.from_numpy(flat).cuda()
        d = T._desc(name)
        nb = L.ndq_mlp_bwd_blocks(ctypes.byref(d), n)
        P = L.ndq_mlp_num_params(ctypes.byref(d))
        parts = torch.zeros(reps, nb, P, device="cuda")
        st = T._stream()
        for r in range(reps):
            rc = L.ndq_mlp_jet_bwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, parts[r].data_ptr(), st)
            assert rc == 0
        torch.cuda.synchronize()
        bad = int((parts != parts[0:1]).any(dim=2).any(dim=1).sum().item())
        print(name, n, "blocks", nb, "launches", reps, "differing from the first:", bad, flush=True)
        if bad:
            tot = parts.sum(dim=1)              # [reps][P] (sum over blocks, not the fixed-order second stage)
            ref = tot[0]
            k = int(((parts != parts[0:1]).any(dim=2).any(dim=1)).nonzero()[0].item())
            diffblocks = (parts[k] != parts[0]).any(dim=1).nonzero().reshape(-1).tolist()
            cols = (parts[k] != parts[0]).any(dim=0).nonzero().reshape(-1).tolist()
            groups = {g: sum(1 for cc in cols if a <= cc < b) for g, a, b in T._groups(name)}
            rel = ((parts[k] - parts[0]).abs().max() / parts[0].abs().max()).item()
            print("   launch", k, "blocks differing", diffblocks[:12], "n =", len(diffblocks), "cols per group", groups, "max rel", rel)
