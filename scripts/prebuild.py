#!/usr/bin/env python
"""Pre-compile (hipcc, no GPU needed) the generated kernels of some test configs under the CURRENT environment -- e.g. with
NDQ_JIT_FLAGS="-DNDQ_DEEP_XCD_REMAP=0" for an A/B run -- so that the GPU box finds them in neurodiffeq_amd/_jit instead of
compiling them inside a timed gpurun call.   usage: [NDQ_JIT_FLAGS=...] scripts/prebuild.py w18 w19:256 ...
Not part of build(): __graft_entry__.build() prunes modules it did not build itself, so run this AFTER build() (or with
NDQ_BUILD_NO_PRUNE=1 set for the next build())."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from neurodiffeq_amd import _hipcc, codegen  # noqa: E402
from neurodiffeq_amd.engine import trace_system  # noqa: E402
from tests import configs  # noqa: E402

with _hipcc.deferred():
    for spec in sys.argv[1:]:
        name = spec.split(":")[0]
        size = int(spec.split(":")[1]) if ":" in spec else None
        torch.manual_seed(0)
        cfg = configs.make(name, size)
        program, descs = trace_system(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg),
                                      compute_func_val=configs.func_val(cfg))
        print(name, "pointwise:", codegen.build(program), flush=True)
        for d in descs.values():   # wide / deep networks: the extension module with their stream / adjoint kernels
            if d.hidden > 64 and codegen.mlp_ext_allowed(d):
                print(name, "mlp ext:", codegen.build_mlp_ext(d), flush=True)
        if codegen.can_fuse(program, descs):
            print(name, "closure:", codegen.build_fused(program, descs[0]), flush=True)
            if os.environ.get("PREBUILD_WIDE"):
                print(name, "closure (8 waves):", codegen.build_fused(program, descs[0], threads=512), flush=True)
