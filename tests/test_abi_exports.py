"""libndq.so loads (no GPU needed) and exports every symbol include/ndq.h declares; descriptor queries work on CPU."""
import ctypes
import os
import re

from neurodiffeq_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    header = open(os.path.join(ROOT, "include", "ndq.h")).read()
    declared = set(re.findall(r"^\s*int\s+(ndq_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    declared64 = set(re.findall(r"^\s*int\s+(ndq64_\w+)\s*\(", header, flags=re.M))      # libndq64.so: the fp64 build
    assert declared64 == set(_lib.EXPORTS64), declared64 ^ set(_lib.EXPORTS64)
    L64 = _lib.lib64()
    for name in declared64:
        assert getattr(L64, name) is not None
    d = _lib.MlpDesc(2, 1, 7, 32, 2, _lib.NDQ_ACT_TANH, 1, 0)
    assert L64.ndq64_mlp_supported(ctypes.byref(d)) == 1 and L64.ndq64_mlp_num_streams(ctypes.byref(d)) == 6


def test_descriptor_queries_without_gpu():
    L = _lib.lib()
    c2 = _lib.MlpDesc(2, 1, 5, 32, 2, _lib.NDQ_ACT_TANH, 1, 0)
    assert L.ndq_mlp_supported(ctypes.byref(c2)) == 1
    assert L.ndq_mlp_num_streams(ctypes.byref(c2)) == 5
    assert L.ndq_mlp_num_params(ctypes.byref(c2)) == 2 * 32 + 32 + 32 * 32 + 32 + 32 + 1 == 1185
    assert L.ndq_mlp_bwd_blocks(ctypes.byref(c2), 65536) >= 1
    c2lap = _lib.MlpDesc(2, 1, 5, 32, 2, _lib.NDQ_ACT_TANH, 1, 1)        # Laplacian stream: value, x, y, xx+yy
    assert L.ndq_mlp_supported(ctypes.byref(c2lap)) == 1 and L.ndq_mlp_num_streams(ctypes.byref(c2lap)) == 4
    c3 = _lib.MlpDesc(2, 1, 1, 64, 3, _lib.NDQ_ACT_TANH, 1, 0)
    assert L.ndq_mlp_num_params(ctypes.byref(c3)) == 8577
    bad = _lib.MlpDesc(2, 1, 5, 80, 2, 0, 1, 0)                          # wider than the kernels go
    assert L.ndq_mlp_supported(ctypes.byref(bad)) == 0
    assert L.ndq_mlp_num_params(ctypes.byref(bad)) == -1


def test_fused_system_refuses_cpu():
    import pytest
    import torch
    from neurodiffeq_amd.engine import FusedSystem
    from tests import configs
    torch.manual_seed(0)
    cfg = configs.make("c2", 8)
    with pytest.raises(_lib.NdqError):
        FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cpu")
