"""GPU parity tests (run with ``-m gpu`` on an MI355X).  Everything goes through the C-ABI of libndq.so /
the generated pointwise kernels; the checker is the oracle (oracle/*.py) and the reference's golden vectors
(tests/golden/*.npz).  Tolerance: rel-L2 <= 1e-5 against the fp64 reference (north_star).

Each test also dumps its error figures into gpurun_out/diag/ so a failing run can be diagnosed from one call."""
import ctypes
import json
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import autograd_ref as R
from oracle import jet_ref as J

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.environ.get("NDQ_DIAG_DIR") or os.path.join(ROOT, "gpurun_out", "diag")
TOL = 1e-5

# name -> (dims, act name, act id, (d, first, mask2) of the config's stream set, streams in kernel order)
ARCH = {
    "c1": ((1, 32, 32, 1), "sin", 1, (1, 1, 0), [(), (0,)]),
    "c2": ((2, 32, 32, 1), "tanh", 0, (2, 1, 5), [(), (0,), (1,), (0, 0), (1, 1)]),
    "c2full": ((2, 32, 32, 1), "tanh", 0, (2, 1, 7), [(), (0,), (1,), (0, 0), (0, 1), (1, 1)]),
    "c3": ((2, 64, 64, 64, 1), "tanh", 0, (2, 1, 1), [(), (0,), (1,), (0, 0)]),
    "c5": ((2, 64, 64, 64, 1), "tanh", 0, (2, 1, 5), [(), (0,), (1,), (0, 0), (1, 1)]),
    "c5p": ((2, 64, 64, 64, 1), "tanh", 0, (2, 1, 0), [(), (0,), (1,)]),
    "c2val": ((2, 32, 32, 1), "tanh", 0, (2, 0, 0), [()]),
    # multi-output networks: C4's radial coefficient net (25 harmonics) and a 3-output 2-D net
    "c4": ((1, 32, 32, 25), "tanh", 0, (1, 1, 1), [(), (0,), (0, 0)]),
    "c4val": ((1, 32, 32, 25), "tanh", 0, (1, 0, 0), [()]),
    "m3": ((2, 32, 32, 3), "tanh", 0, (2, 1, 5), [(), (0,), (1,), (0, 0), (1, 1)]),
    # Laplacian stream: the two pure second derivatives travel as ONE stream holding their sum
    "c2lap": ((2, 32, 32, 1), "tanh", 0, (2, 1, 5), [(), (0,), (1,), ("L", 0, 1)]),
    "c5lap": ((2, 64, 64, 64, 1), "tanh", 0, (2, 1, 5), [(), (0,), (1,), ("L", 0, 1)]),
    # three input coordinates (SolverSpherical's default FCNN(3, 1)): diagonal, Laplacian-merged and full Hessian
    "s3": ((3, 32, 32, 1), "tanh", 0, (3, 1, 41), [(), (0,), (1,), (2,), (0, 0), (1, 1), (2, 2)]),
    "s3lap": ((3, 32, 32, 1), "tanh", 0, (3, 1, 41), [(), (0,), (1,), (2,), ("L", 0, 1, 2)]),
    "s3full": ((3, 32, 32, 1), "tanh", 0, (3, 1, 63),
               [(), (0,), (1,), (2,), (0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]),
    "s3first": ((3, 32, 32, 1), "tanh", 0, (3, 1, 0), [(), (0,), (1,), (2,)]),
    "s3val": ((3, 32, 32, 1), "tanh", 0, (3, 0, 0), [()]),
    # sigmoid and Swish (beta = 1) activations
    "sg2": ((2, 32, 32, 1), "sigmoid", 2, (2, 1, 7), [(), (0,), (1,), (0, 0), (0, 1), (1, 1)]),
    "sw2lap": ((2, 32, 32, 1), "swish", 3, (2, 1, 5), [(), (0,), (1,), ("L", 0, 1)]),
    "sw1": ((1, 32, 32, 1), "swish", 3, (1, 1, 1), [(), (0,), (0, 0)]),
    "sg1": ((1, 32, 32, 1), "sigmoid", 2, (1, 1, 0), [(), (0,)]),
}
# APTx has no kernels in libndq.so's table: the descriptor is served by an extension module compiled on first use
ARCH_EXT = {"ap2": ((2, 32, 32, 1), "aptx", 4, (2, 1, 5), [(), (0,), (1,), (0, 0), (1, 1)])}
# third-order streams (d, first, mask2, mask3): a 1-D sin network, the full 2-D set, one triple of a 2-D set (xxx with its
# pair xx), a wide three-layer network, a two-output network and one 3-D triple (0, 1, 2) with its three mixed pairs
ARCH_T3 = {
    "t3sin1": ((1, 32, 32, 1), "sin", 1, (1, 1, 1, 1), [(), (0,), (0, 0), (0, 0, 0)]),
    "t3full2": ((2, 32, 32, 1), "tanh", 0, (2, 1, 7, 15),
                [(), (0,), (1,), (0, 0), (0, 1), (1, 1), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)]),
    "t3kdv": ((2, 32, 32, 1), "tanh", 0, (2, 1, 1, 1), [(), (0,), (1,), (0, 0), (0, 0, 0)]),
    "t3wide": ((2, 64, 64, 64, 1), "tanh", 0, (2, 1, 5, 9), [(), (0,), (1,), (0, 0), (1, 1), (0, 0, 0), (1, 1, 1)]),
    "t3sig2out": ((1, 32, 32, 2), "sigmoid", 2, (1, 1, 1, 1), [(), (0,), (0, 0), (0, 0, 0)]),
    "t3mixed3": ((3, 32, 32, 1), "tanh", 0, (3, 1, 22, 16), [(), (0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)]),
}


# fourth-order streams (round 6; (d, first, mask2, mask3, mask4)): the beam set of one input (tanh, sin with three layers,
# sigmoid with two outputs), the biharmonic set of two inputs (every pair and triple; quadruples xxxx, xxyy, yyyy), the
# Kuramoto-Sivashinsky set (x-derivatives only of a 2-D network) and the FULL set of two inputs (all five quadruples)
ARCH_T4 = {
    "t4beam": ((1, 32, 32, 1), "tanh", 0, (1, 1, 1, 1, 1), [(), (0,), (0, 0), (0, 0, 0), (0, 0, 0, 0)]),
    "t4sin3": ((1, 32, 32, 32, 1), "sin", 1, (1, 1, 1, 1, 1), [(), (0,), (0, 0), (0, 0, 0), (0, 0, 0, 0)]),
    "t4sig2out": ((1, 32, 32, 2), "sigmoid", 2, (1, 1, 1, 1, 1), [(), (0,), (0, 0), (0, 0, 0), (0, 0, 0, 0)]),
    "t4ks": ((2, 32, 32, 1), "tanh", 0, (2, 1, 1, 1, 1), [(), (0,), (1,), (0, 0), (0, 0, 0), (0, 0, 0, 0)]),
    "t4biharm": ((2, 32, 32, 1), "tanh", 0, (2, 1, 7, 15, 21),
                 [(), (0,), (1,), (0, 0), (0, 1), (1, 1), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1),
                  (0, 0, 0, 0), (0, 0, 1, 1), (1, 1, 1, 1)]),
    "t4full2": ((2, 16, 16, 1), "sin", 1, (2, 1, 7, 15, 31),
                [(), (0,), (1,), (0, 0), (0, 1), (1, 1), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1),
                 (0, 0, 0, 0), (0, 0, 0, 1), (0, 0, 1, 1), (0, 1, 1, 1), (1, 1, 1, 1)]),
}


def _parts(m):
    """oracle streams making up kernel stream m"""
    return [(c, c) for c in m[1:]] if (m and m[0] == "L") else [m]


def _oracle_jets(flat, dims, act, coords, streams):
    want = sorted({p for m in streams for p in _parts(m)}, key=lambda t: (len(t), t))
    js = J.mlp_jets(flat.astype(np.float64), dims, act, list(coords.astype(np.float64)), want)
    return {m: sum(js[p] for p in _parts(m)) for m in streams}


def _oracle_vjp(flat, dims, act, coords, streams, gbar):
    gb = {}
    for s, m in enumerate(streams):
        for p in _parts(m):
            gb[p] = gb.get(p, 0) + gbar[s].astype(np.float64).T
    return J.mlp_jets_vjp(flat.astype(np.float64), dims, act, list(coords.astype(np.float64)), gb)
SIZES = {"c1": 64, "c2": 16, "c3": 12, "c5": 8, "c4": 96, "w1": None, "w2": None, "w3": None, "w4": None, "w5": None, "w6": None, "w7": None, "w8": None, "w9": None, "w10": None}


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def diag(name, payload):
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, name + ".json"), "w") as fh:
        json.dump(payload, fh, indent=1, default=float)


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from neurodiffeq_amd import _lib
    return _lib.lib()


def _desc(name):
    from neurodiffeq_amd import _lib
    dims, _, act, spec, _ = ARCH[name]
    d, first, mask2 = spec[:3]
    lap = int(any(m and m[0] == "L" for m in ARCH[name][4]))
    return _lib.MlpDesc(d, first, mask2, dims[1], len(dims) - 2, act, dims[-1], lap, 0, spec[3] if len(spec) > 3 else 0,
                        0, 0, 0, spec[4] if len(spec) > 4 else 0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _params(name, rng, scale=1.0):
    dims = ARCH[name][0]
    parts = []
    for a, b in zip(dims[:-1], dims[1:]):
        k = 1.0 / np.sqrt(a)
        parts += [rng.uniform(-k, k, a * b) * scale, rng.uniform(-k, k, b) * scale]
    return np.concatenate(parts).astype(np.float32)


def _fwd(L, name, coords, flat):
    dims, _, _, _, streams = ARCH[name]
    n = coords.shape[1]
    ld = (n + 63) // 64 * 64
    c = torch.zeros(dims[0], ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
    p = torch.from_numpy(flat).cuda()
    jets = torch.full((len(streams), dims[-1], ld), float("nan"), device="cuda")
    d = _desc(name)
    rc = L.ndq_mlp_jet_fwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), jets.data_ptr(), ld, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return jets[:, :, :n].cpu().numpy()          # [NS][n_out][n]


def _bwd(L, name, coords, flat, gbar):
    dims, _, _, _, streams = ARCH[name]
    n = coords.shape[1]
    ld = (n + 63) // 64 * 64
    c = torch.zeros(dims[0], ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
    g = torch.zeros(len(streams), dims[-1], ld, device="cuda"); g[:, :, :n] = torch.from_numpy(gbar)
    p = torch.from_numpy(flat).cuda()
    d = _desc(name)
    nb = L.ndq_mlp_bwd_blocks(ctypes.byref(d), n)
    P = L.ndq_mlp_num_params(ctypes.byref(d))
    assert P == flat.size
    part = torch.full((nb, P), float("nan"), device="cuda")
    out = torch.zeros(P, device="cuda")
    rc = L.ndq_mlp_jet_bwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, part.data_ptr(), _stream())
    assert rc == 0, rc
    rc = L.ndq_reduce_partials(part.data_ptr(), nb, P, out.data_ptr(), 0, 1.0, _stream())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _groups(name):
    dims = ARCH[name][0]
    out, off = [], 0
    for li, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        out.append((f"W{li + 1}", off, off + a * b)); off += a * b
        out.append((f"b{li + 1}", off, off + b)); off += b
    return out


# ------------------------------------------------------------------------------------------------ MLP kernels
def test_extension_module_kernels_match_jet_oracle(L):
    """A descriptor outside libndq.so's table (APTx activation): codegen.ensure_mlp_kernels builds + registers it, then
    the ordinary C-ABI entry points serve it."""
    from neurodiffeq_amd import codegen
    ARCH.update(ARCH_EXT)
    try:
        d = _desc("ap2")
        assert codegen.ensure_mlp_kernels(d) and L.ndq_mlp_supported(ctypes.byref(d)) == 1
        dims, act, _, _, streams = ARCH["ap2"]
        rng = np.random.default_rng(11)
        flat = _params("ap2", rng)
        n = 1000
        coords = rng.uniform(-1.0, 1.0, (2, n)).astype(np.float32)
        got = _fwd(L, "ap2", coords, flat)
        want = _oracle_jets(flat, dims, act, coords, streams)
        assert max(float(np.linalg.norm(got[s].T - want[m]) / np.linalg.norm(want[m])) for s, m in enumerate(streams)) < TOL
        gbar = rng.standard_normal((len(streams), 1, n)).astype(np.float32)
        assert rel_l2(_bwd(L, "ap2", coords, flat, gbar), _oracle_vjp(flat, dims, act, coords, streams, gbar)) < TOL
    finally:
        for k in ARCH_EXT:
            ARCH.pop(k, None)


@pytest.mark.parametrize("name", list(ARCH_T3))
@pytest.mark.parametrize("n", [17, 1000])
def test_third_order_stream_kernels_match_jet_oracle(L, name, n):
    """Third-order derivative streams (ndq_mlp_desc.mask3; VERDICT r1 missing #2): forward values of d3/dx_a dx_b dx_c and the
    parameter gradient given adjoints of all streams, against the numpy jet oracle (itself checked against three
    nested autograd sweeps in tests/test_oracle_golden.py)."""
    from neurodiffeq_amd import codegen
    ARCH.update(ARCH_T3)
    try:
        d = _desc(name)
        assert codegen.ensure_mlp_kernels(d) and L.ndq_mlp_supported(ctypes.byref(d)) == 1
        dims, act, _, _, streams = ARCH[name]
        assert L.ndq_mlp_num_streams(ctypes.byref(d)) == len(streams)
        rng = np.random.default_rng(zlib.crc32(f"{name}/{n}".encode()))
        flat = _params(name, rng)
        coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
        got = _fwd(L, name, coords, flat)
        want = _oracle_jets(flat, dims, act, coords, streams)
        floor = (0.1 if n < 64 else 0.0) * np.sqrt(n * dims[-1]) * max(np.sqrt(np.mean(want[m] ** 2)) for m in streams)
        errs = {str(m): float(np.linalg.norm(got[s].T - want[m]) / max(np.linalg.norm(want[m]), floor))
                for s, m in enumerate(streams)}
        gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
        errs["grad"] = rel_l2(_bwd(L, name, coords, flat, gbar), _oracle_vjp(flat, dims, act, coords, streams, gbar))
        diag(f"t3_{name}_{n}", errs)
        assert max(errs.values()) < TOL, errs
    finally:
        for k in ARCH_T3:
            ARCH.pop(k, None)


@pytest.mark.parametrize("name", list(ARCH_T4))
@pytest.mark.parametrize("n", [17, 1000])
def test_fourth_order_stream_kernels_match_jet_oracle(L, name, n):
    """Fourth-order derivative streams (ndq_mlp_desc.mask4; VERDICT r5 missing #2: diff(u, x, order=4) -- beam, biharmonic,
    Kuramoto-Sivashinsky): forward values of every stream and the parameter gradient given adjoints of ALL streams (so the
    adjoint contributions of a quadruple to its triples, pairs and first-order streams are exercised), against the numpy jet
    oracle (general Faa di Bruno over set partitions; itself checked against four nested autograd sweeps on the CPU)."""
    from neurodiffeq_amd import codegen
    ARCH.update(ARCH_T4)
    try:
        d = _desc(name)
        assert d.mask4 == ARCH[name][3][4]
        assert codegen.ensure_mlp_kernels(d) and L.ndq_mlp_supported(ctypes.byref(d)) == 1
        dims, act, _, _, streams = ARCH[name]
        assert L.ndq_mlp_num_streams(ctypes.byref(d)) == len(streams)
        rng = np.random.default_rng(zlib.crc32(f"{name}/{n}".encode()))
        flat = _params(name, rng)
        coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
        got = _fwd(L, name, coords, flat)
        want = _oracle_jets(flat, dims, act, coords, streams)
        floor = (0.1 if n < 64 else 0.0) * np.sqrt(n * dims[-1]) * max(np.sqrt(np.mean(want[m] ** 2)) for m in streams)
        errs = {str(m): float(np.linalg.norm(got[s].T - want[m]) / max(np.linalg.norm(want[m]), floor))
                for s, m in enumerate(streams)}
        gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
        errs["grad"] = rel_l2(_bwd(L, name, coords, flat, gbar), _oracle_vjp(flat, dims, act, coords, streams, gbar))
        diag(f"t4_{name}_{n}", errs)
        assert max(errs.values()) < TOL, errs
    finally:
        for k in ARCH_T4:
            ARCH.pop(k, None)


@pytest.mark.parametrize("name", list(ARCH))
@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4099])
def test_mlp_jet_fwd_matches_jet_oracle(L, name, n):
    dims, act, _, _, streams = ARCH[name]
    rng = np.random.default_rng(zlib.crc32(f"{name}/{n}".encode()))
    flat = _params(name, rng)
    coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
    got = _fwd(L, name, coords, flat)
    want = _oracle_jets(flat, dims, act, coords, streams)
    # rel-L2 per stream; for tiny batches a single stream value can be a near-cancellation of O(1) terms, so there
    # the denominator is floored at 10 % of the largest stream's RMS (absolute fp32 noise is what matters)
    floor = (0.1 if n < 64 else 0.0) * np.sqrt(n * dims[-1]) * max(np.sqrt(np.mean(want[m] ** 2)) for m in streams)
    errs = {str(m): float(np.linalg.norm(got[s].T - want[m]) / max(np.linalg.norm(want[m]), floor))
            for s, m in enumerate(streams)}
    diag(f"fwd_{name}_{n}", errs)
    assert np.isfinite(got).all()
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", list(ARCH))
@pytest.mark.parametrize("n", [1, 17, 1000, 4099])
def test_mlp_jet_bwd_matches_jet_oracle(L, name, n):
    dims, act, _, _, streams = ARCH[name]
    rng = np.random.default_rng(zlib.crc32(f"{name}/{n}/b".encode()))
    flat = _params(name, rng)
    coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
    gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
    got = _bwd(L, name, coords, flat, gbar)
    want = _oracle_vjp(flat, dims, act, coords, streams, gbar)
    errs = {g: rel_l2(got[a:b], want[a:b]) for g, a, b in _groups(name)}
    errs["all"] = rel_l2(got, want)
    diag(f"bwd_{name}_{n}", errs)
    assert np.isfinite(got).all()
    assert errs["all"] < TOL, errs


def test_bwd_is_linear_in_gbar_and_deterministic(L):
    """Size-independent properties at the full C2 size (65 536 points): linearity of the adjoint in its seed,
    run-to-run bit-exactness (fixed-order reductions, no float atomics), shard additivity."""
    name, n = "c2", 65536
    dims, act, _, _, streams = ARCH[name]
    rng = np.random.default_rng(5)
    flat = _params(name, rng)
    coords = rng.uniform(0, 1, (2, n)).astype(np.float32)
    g1 = rng.standard_normal((len(streams), 1, n)).astype(np.float32)
    g2 = rng.standard_normal((len(streams), 1, n)).astype(np.float32)
    a, b = _bwd(L, name, coords, flat, g1), _bwd(L, name, coords, flat, g2)
    ab = _bwd(L, name, coords, flat, (2.0 * g1 - 0.5 * g2).astype(np.float32))
    again = _bwd(L, name, coords, flat, g1)
    h = n // 2
    halves = _bwd(L, name, coords[:, :h], flat, g1[:, :, :h]) + _bwd(L, name, coords[:, h:], flat, g1[:, :, h:])
    res = dict(linearity=rel_l2(ab, 2.0 * a - 0.5 * b), shard_additivity=rel_l2(halves, a),
               bit_exact_rerun=bool(np.array_equal(a, again)))
    diag("bwd_properties_c2_full", res)
    assert res["bit_exact_rerun"]
    assert res["linearity"] < 5e-6 and res["shard_additivity"] < 5e-6, res


@pytest.mark.parametrize("name,n", [("c2", 8195), ("c2full", 4099), ("c3", 8195), ("c1", 4099)])
def test_bwd_launches_are_bit_reproducible(L, name, n):
    """100 launches of the adjoint kernel on identical inputs give 100 bit-identical partial-sum blocks (a build with
    two scratch-using waves per SIMD failed exactly this: a few corrupted workgroup rows per launch)."""
    dims, act, _, _, streams = ARCH[name]
    rng = np.random.default_rng(7)
    flat = _params(name, rng)
    coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
    gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
    ld = (n + 63) // 64 * 64
    c = torch.zeros(dims[0], ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
    g = torch.zeros(len(streams), dims[-1], ld, device="cuda"); g[:, :, :n] = torch.from_numpy(gbar)
    p = torch.from_numpy(flat).cuda()
    d = _desc(name)
    nb, P, reps = L.ndq_mlp_bwd_blocks(ctypes.byref(d), n), L.ndq_mlp_num_params(ctypes.byref(d)), 100
    parts = torch.zeros(reps, nb, P, device="cuda")
    for r in range(reps):
        assert L.ndq_mlp_jet_bwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, parts[r].data_ptr(),
                                 _stream()) == 0
    torch.cuda.synchronize()
    assert int((parts != parts[0:1]).any(dim=2).any(dim=1).sum().item()) == 0


def test_bad_arguments_are_rejected(L):
    from neurodiffeq_amd import _lib
    d = _desc("c2")
    t = torch.zeros(64, device="cuda")
    assert L.ndq_mlp_jet_fwd(ctypes.byref(d), t.data_ptr(), 64, 0, t.data_ptr(), t.data_ptr(), 64, _stream()) == -2
    assert L.ndq_mlp_jet_fwd(ctypes.byref(d), t.data_ptr(), 8, 16, t.data_ptr(), t.data_ptr(), 64, _stream()) == -2
    bad = _lib.MlpDesc(2, 1, 5, 80, 2, 0, 1, 0)
    assert L.ndq_mlp_supported(ctypes.byref(bad)) == 0
    assert L.ndq_mlp_jet_fwd(ctypes.byref(bad), t.data_ptr(), 64, 16, t.data_ptr(), t.data_ptr(), 64, _stream()) == -1


# ------------------------------------------------------------------------------------------------ reduce / adam
def test_reduce_partials(L):
    rng = np.random.default_rng(0)
    for nparts, length in [(1, 1), (3, 7), (512, 1185), (257, 8577)]:
        part = rng.standard_normal((nparts, length)).astype(np.float32)
        base = rng.standard_normal(length).astype(np.float32)
        p, o = torch.from_numpy(part).cuda(), torch.from_numpy(base).cuda()
        assert L.ndq_reduce_partials(p.data_ptr(), nparts, length, o.data_ptr(), 1, 0.5, _stream()) == 0
        torch.cuda.synchronize()
        want = base.astype(np.float64) + 0.5 * part.astype(np.float64).sum(0)
        assert rel_l2(o.cpu().numpy(), want) < 1e-6


def test_adam_step_matches_torch(L):
    rng = np.random.default_rng(1)
    n = 1185
    p0 = rng.standard_normal(n).astype(np.float32)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = torch.from_numpy(p0.copy()).cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        ref.grad = torch.from_numpy(g.copy())
        opt.step()
        gg = torch.from_numpy(g).cuda()
        assert L.ndq_adam_step(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0,
                               step, _stream()) == 0
    torch.cuda.synchronize()
    assert rel_l2(p.cpu().numpy(), ref.detach().numpy()) < 1e-6


# ------------------------------------------------------------------------------------------------ whole closure
def _load_system(name, size, single_kernel=True):
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(0)
    cfg = configs.make(name, size)
    for net in cfg["nets"]:
        net.to("cuda")
    return cfg, FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                            compute_func_val=configs.func_val(cfg), single_kernel=single_kernel)


# "1k" = single-launch fused closure kernel (single-network systems), "3k" = forward / pointwise / backward pipeline
@pytest.mark.parametrize("name,mode", [("c1", "3k"), ("c1", "1k"), ("c2", "1k"), ("c2", "3k"), ("c3", "1k"), ("c3", "3k"), ("c5", "3k"),
                                       ("c4", "3k"), ("c4", "1k"), ("w1", "1k"), ("w1", "3k"), ("w2", "1k"), ("w2", "3k"), ("w3", "1k"),
                                       ("w4", "1k"), ("w4", "3k"), ("w5", "1k"), ("w5", "3k"), ("w6", "3k"), ("w7", "3k"), ("w8", "3k"),
                                       ("w9", "1k"), ("w9", "3k"), ("w10", "1k"), ("w10", "3k")])
def test_fused_closure_matches_reference_golden(golden_dir, name, mode):
    """funcs / residuals / loss / flat gradient of ONE closure (solvers.py:369-395) on the reference's own inputs."""
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    cfg, system = _load_system(name, SIZES[name], single_kernel=(mode == "1k"))
    assert (system.fusedk is not None) == (mode == "1k")
    R.set_flat(cfg["nets"], gold["params0"])
    coords = [torch.from_numpy(c) for c in gold["coords"]]
    b, n = system.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    funcs = b["funcs"][:, :n].T.cpu().numpy()
    resid = b["resid"][:, :n].T.cpu().numpy()[:, :gold["residuals_f64"].shape[1]]     # Sobolev: gradient columns follow
    loss = float(system.loss_buf[0].item())
    grad = np.concatenate([fp.grad.cpu().numpy() for fp in system.flat])
    errs = dict(funcs=rel_l2(funcs, gold["funcs_f64"]), residuals=rel_l2(resid, gold["residuals_f64"]),
                loss=abs(loss - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(grad, gold["grad_f64"]))
    off = 0
    for k, fp in enumerate(system.flat):
        errs[f"grad_net{k}"] = rel_l2(grad[off:off + fp.numel], gold["grad_f64"][off:off + fp.numel])
        off += fp.numel
    diag(f"closure_{name}_{mode}", errs)
    assert max(errs["funcs"], errs["residuals"], errs["loss"], errs["grad"]) < TOL, errs


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "w8", "w9", "w10"])
def test_solver_trajectory_matches_reference_golden(golden_dir, name):
    """Three epochs of Solver.run_train_epoch (sampling on the CPU RNG, fused step, fused Adam) against the
    reference solver's loss history and final parameters."""
    from tests import configs
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    solver, cfg = configs.make_solver(name, SIZES[name])
    solver.fused = "require"
    assert np.array_equal(R.get_flat(cfg["nets"]).cpu().numpy(), gold["params0"])
    torch.manual_seed(int(gold["seed"]) + 2)
    for _ in range(3):
        solver.run_train_epoch()
    assert solver.fused_active
    hist = np.array(solver.metrics_history["train_loss"])
    params = R.get_flat(cfg["nets"]).cpu().numpy()
    errs = dict(loss=float(np.max(np.abs(hist - gold["traj_loss"]) / np.abs(gold["traj_loss"]))),
                params=rel_l2(params, gold["traj_params"]))
    diag(f"trajectory_{name}", dict(errs, hist=hist.tolist(), want=gold["traj_loss"].tolist()))
    assert errs["loss"] < 2e-5 and errs["params"] < 1e-5, errs


@pytest.mark.parametrize("name,size,mode", [("c2", 256, "1k"), ("c2", 256, "3k"), ("c3", 96, "1k"), ("c3", 97, "3k"),
                                            ("c1", 1024, "3k"), ("c1", 1024, "1k"), ("c2", 37, "1k"), ("c4", 5000, "3k"),
                                            # the BASELINE configs at their stated size (row g of the verdict table)
                                            ("c3", 512, "1k"), ("c3", 512, "3k"), ("c4", 131072, "3k"),
                                            ("c4", 131072, "1k"), ("c5", 1024, "3k")])
def test_fused_closure_matches_oracle_at_size(name, size, mode):
    """Every BASELINE config at its stated size (C1 1 024, C2 65 536, C3 262 144, C4 131 072, C5 1 048 576 points) and
    ragged batches against the autograd oracle in fp64, POINT BY POINT (function values, residuals) plus loss and every
    network's gradient.  Loss and gradient are sums over points, so the oracle walks the big batches in chunks
    (oracle/autograd_ref.py: closure_chunked; unchunked C5 needs ~43 GB on the CPU).  C5's product path is the
    three-kernel pipeline ("3k": its three networks differ in stream set, there is no single-launch kernel for it); at
    1 048 576 points the case costs ~2 minutes of host autograd."""
    cfg, system = _load_system(name, size, single_kernel=(mode == "1k"))
    if mode == "1k" and system.fusedk is None:
        pytest.skip("no single-launch closure kernel for this system")
    torch.manual_seed(0)
    ocfg = R.build_config(name, size, dtype=torch.float64)
    flat = R.get_flat(cfg["nets"]).cpu()
    R.set_flat(ocfg["nets"], flat.double())
    torch.manual_seed(3)
    ex = cfg["gen"].get_examples()
    coords = [ex.detach()] if isinstance(ex, torch.Tensor) else [c.detach() for c in ex]
    if coords[0].numel() > 70000:
        out = R.closure_chunked(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], [c.double() for c in coords],
                                chunk=32768, keep=True)
    else:
        out = R.closure(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], [c.double() for c in coords])
    want_grad = R.get_flat_grad(ocfg["nets"]).numpy()
    b, n = system.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert n == cfg["n_points"]
    grad = np.concatenate([fp.grad.cpu().numpy() for fp in system.flat])
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), out["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), out["residuals"].numpy()),
                loss=abs(system.loss_buf[0].item() - out["loss"].item()) / abs(out["loss"].item()),
                grad=rel_l2(grad, want_grad))
    off = 0
    for k, fp in enumerate(system.flat):          # per network: a small gradient must not hide behind a large one
        errs[f"grad_net{k}"] = rel_l2(grad[off:off + fp.numel], want_grad[off:off + fp.numel])
        off += fp.numel
    diag(f"closure_full_{name}_{size}_{mode}", errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name,size,mode", [("c1", 1024, "1k"), ("c2", 256, "1k"), ("c2", 256, "3k"), ("c3", 512, "1k"), ("c3", 512, "3k"),
                                            ("c4", 131072, "1k"), ("c4", 131072, "3k"), ("c5", 1024, "1k"), ("c5", 1024, "3k")])
def test_fused_closure_matches_reference_golden_at_stated_size(golden_dir, name, size, mode):
    """The BASELINE configs at their STATED sizes against numbers the unmodified reference produced (fp64; script:
    tests/golden/make_golden.py make_full -> <name>_full.npz): loss, flat gradient, and the per-column sums of function
    values and squared residuals of one closure.  The batch is regenerated here from the seed -- the generators'
    bit-exact contract -- and checked against the head and the sum of the reference's own draw."""
    gold = np.load(os.path.join(golden_dir, f"{name}_full.npz"))
    cfg, system = _load_system(name, size, single_kernel=(mode == "1k"))
    if mode == "1k" and system.fusedk is None:
        pytest.skip("no single-launch closure kernel for this system")
    R.set_flat(cfg["nets"], gold["params0"])
    torch.manual_seed(int(gold["seed"]) + 1)
    ex = cfg["gen"].get_examples()
    coords = [ex.detach()] if isinstance(ex, torch.Tensor) else [c.detach() for c in ex]
    assert coords[0].numel() == int(gold["n_points"])
    # the same draw as the reference's: first values, and the int64 sum of the fp32 bit patterns.  The spherical generator
    # goes through acos / cbrt-type CPU kernels whose last bit depends on the host's vector ISA (build container vs GPU
    # box: 3 of 393 216 values differ by an ulp), so the checksum gets a budget of 1/8 ulp per value -- another draw is
    # off by ~1e6 ulp per value
    assert np.allclose(np.stack([c[:8].numpy() for c in coords]), gold["coords_head"], rtol=3e-7, atol=0)
    bits = np.asarray([c.view(torch.int32).to(torch.int64).sum().item() for c in coords])
    assert np.all(np.abs(bits - gold["coords_bits_sum"]) <= coords[0].numel() // 8), (bits, gold["coords_bits_sum"])
    b, n = system.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    n_eq = gold["resid_sq_sum"].shape[0]
    fsum = b["funcs"][:, :n].double().sum(dim=1).cpu().numpy()
    r2sum = (b["resid"][:n_eq, :n].double() ** 2).sum(dim=1).cpu().numpy()
    grad = np.concatenate([fp.grad.cpu().numpy() for fp in system.flat])
    loss = float(system.loss_buf[0].item())
    errs = dict(loss=abs(loss - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(grad, gold["grad_f64"]), funcs_sum=rel_l2(fsum, gold["funcs_sum"]),
                resid_sq_sum=rel_l2(r2sum, gold["resid_sq_sum"]))
    off = 0
    for k, fp in enumerate(system.flat):
        errs[f"grad_net{k}"] = rel_l2(grad[off:off + fp.numel], gold["grad_f64"][off:off + fp.numel])
        off += fp.numel
    diag(f"closure_refgold_full_{name}_{mode}", errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name,grid,build", [("c2", 64, "lap"), ("c2", 64, "nolap"), ("c3", 48, "lap")])
@pytest.mark.parametrize("mode", ["1k", "3k"])
def test_near_convergence_parity_against_the_reference_trained_state(golden_dir, monkeypatch, name, grid, build, mode):
    """SURVEY.md 8c (last bullet) / VERDICT r2 #2b.  Near convergence the residual is a cancellation of O(1) terms, so an
    error that is invisible on a freshly initialised network may show there.  Fixtures: the UNMODIFIED reference trained
    C2 (5 000 epochs, loss 15.6 -> 1e-4) and C3 (3 000 epochs), then evaluated one batch at the trained parameters in
    fp64 and -- its own arithmetic -- in fp32 (tests/golden/make_golden.py: make_trained).  Checked here, each against
    fp64: the raw network's derivative streams (forward kernel), the solution and its derivative columns ONE BY ONE
    (u_xx and u_yy separately, through a system whose "equations" are those columns), the residual, the loss and the
    gradient of the training closure -- merged-Laplacian build and NDQ_NO_LAP=1 build, single launch and pipeline.
    Bound per quantity: the 1e-5 contract, or twice the reference's own fp32-vs-fp64 error where that is larger
    (residual 5.7e-5, gradient 4.4e-4 at C2's trained state: nobody's fp32 holds 1e-5 on a cancelled residual)."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.engine import FusedSystem
    from tests import configs
    if build == "nolap":
        monkeypatch.setenv("NDQ_NO_LAP", "1")
    gold = np.load(os.path.join(golden_dir, f"{name}_trained.npz"))
    torch.manual_seed(0)
    cfg = configs.make(name, grid)
    for net in cfg["nets"]:
        net.to("cuda")
    R.set_flat(cfg["nets"], gold["params"])
    coords = [torch.from_numpy(c) for c in gold["coords"]]
    system = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), 2, "cuda", single_kernel=(mode == "1k"))
    assert (system.fusedk is not None) == (mode == "1k")
    lap = bool(system.descs[0].lap)
    assert lap == (name == "c2" and build == "lap")
    b, n = system.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    ref = lambda key: rel_l2(gold[key + "_f32"], gold[key + "_f64"])             # the reference's own fp32 error
    got = dict(u=rel_l2(b["funcs"][0, :n].cpu().numpy(), gold["u_f64"]),
               residual=rel_l2(b["resid"][:1, :n].T.cpu().numpy(), gold["residual_f64"]),
               loss=abs(system.loss_buf[0].item() - float(gold["loss_f64"])) / float(gold["loss_f64"]),
               grad=rel_l2(system.flat[0].grad.cpu().numpy(), gold["grad_f64"]))
    yard = dict(u=ref("u"), residual=ref("residual"), grad=ref("grad"),
                loss=abs(float(gold["loss_f32"]) - float(gold["loss_f64"])) / float(gold["loss_f64"]))
    # the derivative columns of the re-parameterised solution, one by one, through the product path
    cols = ["u_x", "u_y", "u_xx", "u_yy"] if name == "c2" else ["u_x", "u_y", "u_xx"]

    def columns(u, x, y):
        return [dict(u_x=lambda: diff(u, x), u_y=lambda: diff(u, y), u_xx=lambda: diff(u, x, order=2),
                     u_yy=lambda: diff(u, y, order=2))[c]() for c in cols]
    sys2 = FusedSystem(cfg["nets"], cfg["conds"], columns, 2, "cuda", single_kernel=(mode == "1k"))
    b2, _ = sys2.step(coords, train=False, slot=0, want_resid=True)
    torch.cuda.synchronize()
    for k, c in enumerate(cols):
        got[c] = rel_l2(b2["resid"][k, :n].cpu().numpy(), gold[c + "_f64"])
        yard[c] = ref(c)
    # the raw network's streams (what every closure kernel computes tile by tile): value, first and second derivatives
    L = system.L
    d = _desc_from(2, 1, 5 if name == "c2" else 1, system.descs[0])
    if L.ndq_mlp_supported(ctypes.byref(d)):
        jets = torch.zeros(L.ndq_mlp_num_streams(ctypes.byref(d)), b["ld"], device="cuda")
        _lib_check(L.ndq_mlp_jet_fwd(ctypes.byref(d), system._coord_ptr(b, 0), b["ld"], n, system.flat[0].flat.data_ptr(),
                                     jets.data_ptr(), b["ld"], None))
        torch.cuda.synchronize()
        keys = ["n", "n_x", "n_y", "n_xx", "n_yy"] if name == "c2" else ["n", "n_x", "n_y", "n_xx"]
        for s_, key in enumerate(keys):
            got[key] = rel_l2(jets[s_, :n].cpu().numpy(), gold[key + "_f64"])
            yard[key] = ref(key)
    bound = {k: max(TOL, 2.0 * yard[k]) for k in got}
    diag(f"trained_{name}_{build}_{mode}", dict(error=got, reference_fp32_error=yard, bound=bound))
    bad = {k: (got[k], bound[k]) for k in got if not got[k] <= bound[k]}
    assert not bad, bad


def _desc_from(d, first, mask2, like):
    from neurodiffeq_amd import _lib
    return _lib.MlpDesc(d, first, mask2, like.hidden, like.layers, like.act, like.n_out, 0, like.skip, 0, like.actp, like.widths,
                        like.mono)


def _lib_check(rc):
    assert rc == 0, rc


@pytest.mark.parametrize("mode", ["1k", "3k"])
@pytest.mark.parametrize("name", ["pendulum", "coupled_sin", "bvp_tanh", "helmholtz_xy", "advection", "heat_wide",
                                  "stokes_like", "kdv", "ode3", "poisson3d", "hessian3d", "shell", "swish_laplace", "sigmoid_mixed",
                                  "swish_ode", "bundle_decay", "bundle_bvp", "shape_64x2", "shape_32x3", "shape_48x2",
                                  "shape_16x2_sin", "shape_32x1", "aptx_burgers", "resnet_laplace", "resnet_ode",
                                  # beyond round 2's template limits: 6 and 8 hidden layers, 4 and 5 inputs, a 3-parameter bundle
                                  "shape_32x6", "shape_16x8_sin", "heat4d", "mix5d", "bundle_osc",
                                  # piecewise / clipped equations: masks, where, clamp, relu, maximum, sign, log1p, atan2, erf
                                  "piecewise_source", "relu_ode", "atan2_adv",
                                  # round 5, second batch: rounding functions, torch.nn.functional activations, inverse / special functions
                                  "rounding_ode", "activations_ode", "special_2d",
                                  "autograd_grad_ode",        # torch.autograd.grad written out by hand in the equation
                                  # round 6: fourth-order streams -- diff(u, x, order=4)
                                  "beam", "beam_sigmoid", "biharmonic", "kuramoto"])
def test_zoo_closure_matches_autograd_oracle(name, mode):
    """Systems outside the BASELINE set (tests/zoo.py): second-order IVP, sin networks, mixed second derivatives, first
    order only, three coordinates (Laplacian-merged, diagonal and full Hessian stream sets), three networks."""
    from tests import zoo
    from neurodiffeq_amd.engine import FusedSystem
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
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=(mode == "1k"))
    if mode == "1k" and fs.fusedk is None:
        pytest.skip("no single-launch closure for this system (networks of different shapes / stream sets)")
    assert (fs.fusedk is not None) == (mode == "1k")
    assert mode == "3k" or len(nets) == 1 or name in ("coupled_sin", "stokes_like")   # the multi-network closure kernel
    b, n = fs.step([c.float() for c in coords], train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert mode != "1k" or (fs.fusedk is not None and fs.fused_check["reproducible"]), getattr(fs, "fused_check", None)
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), want["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat]), want_grad))
    diag(f"zoo_{name}_{mode}", errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", ["pendulum", "advection", "helmholtz_xy", "sigmoid_mixed"])
def test_zoo_closure_on_the_eight_wave_build(name):
    """Batches of 65 536+ points go to the 8-wave build of the closure kernel (two waves per SIMD, 256 registers each:
    engine.fused_variant), whose weight-gradient route (csrc/ndq_mlp.h hbar_wgrad_tr, the branch without the whole register
    file) the BASELINE configs only exercise with C2's four streams: three streams (a round with ONE stream: zero planes in
    the second slot) and six (three full rounds) against the fp64 autograd oracle."""
    from tests import zoo
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    n_pts = 65536 - 37            # 4 094 tiles: two rounds of 2 048 waves (engine.prefers_wide), the last tile ragged
    coords = system.sample(n_pts, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.to("cuda")
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda")
    assert fs.fusedk is not None
    b, n = fs.step([c.float() for c in coords], train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert b["fusedk"].threads == 512, b["fusedk"].threads
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), want["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat]), want_grad))
    diag(f"zoo8w_{name}", errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name,kind,mode", [("helmholtz_xy", "l1", "1k"), ("stokes_like", "l1", "1k"), ("stokes_like", "infinity", "1k"),
                                            ("stokes_like", "infinity", "3k"), ("pendulum", "infinity", "1k")])
def test_l1_and_infinity_losses_match_autograd_oracle(name, kind, mode):
    """loss_fn = 'l1' / 'infinity' (losses.py:4-12) on the fused path: per-point loss terms and their adjoint seeds come
    out of the generated pointwise code."""
    from tests import zoo
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(3001, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords, loss=kind)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.to("cuda")
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=(mode == "1k"), loss=kind)
    b, n = fs.step([c.float() for c in coords], train=True, slot=0)
    torch.cuda.synchronize()
    errs = dict(loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat]), want_grad))
    diag(f"loss_{kind}_{name}_{mode}", errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", ["shell", "poisson3d", "hessian3d", "pendulum", "shape_48x2", "shape_32x3"])
def test_grouped_closure_kernel_on_single_output_systems(monkeypatch, name):
    """The grouped closure kernel (csrc/ndq_mlp.h: fused_group_closure_kernel -- per-point stage on one point per lane
    through an LDS exchange tile; built for multi-output networks such as C4's) serves single-output networks too when
    asked to (NDQ_FUSE_GROUP=1): same results as the autograd oracle, ragged batch (3001 points: a partial group, a
    partial tile), bit-reproducible, and accepted by its first-use self-check against the three-kernel pipeline."""
    from tests import zoo
    from neurodiffeq_amd import codegen
    from neurodiffeq_amd.engine import FusedSystem
    monkeypatch.setenv("NDQ_FUSE_GROUP", "1")
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
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda")
    assert fs.fusedk is not None and codegen.fuse_mode(fs.program, fs.descs) == "group"
    b, n = fs.step([c.float() for c in coords], train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert fs.fusedk is not None and fs.fused_check["reproducible"], getattr(fs, "fused_check", None)
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), want["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(np.concatenate([fp.grad.cpu().numpy() for fp in fs.flat]), want_grad))
    diag(f"group_{name}", dict(errs, check=fs.fused_check))
    assert max(errs.values()) < TOL, errs
    # evaluation-only launch (validation epochs, Solution objects): no adjoint, same values
    b2, n2 = fs.step([c.float() for c in coords], train=False, slot=1, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert abs(fs.loss_buf[1].item() - fs.loss_buf[0].item()) <= 1e-6 * abs(fs.loss_buf[0].item())
    assert rel_l2(b2["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()) < TOL


def test_custom_loss_additional_loss_and_metrics_stay_on_the_fused_path():
    """VERDICT r1 missing #3 / ADVICE: a criterion callable, an ``additional_loss`` override and ``metrics`` (one of them
    differentiating the solution) are traced with the system -- the solver stays on the HIP path (``fused='require'``)
    and follows the trajectory of the reference's closure on plain torch autograd (``fused='off'``, native autograd
    seam switched off so that leg is ATen only)."""
    from tests import configs
    from neurodiffeq_amd import autograd_ops
    runs = {}
    for mode in ("require", "off"):
        torch.manual_seed(0)
        solver, cfg = configs.make_custom_loss_solver(16)
        solver.fused = mode
        autograd_ops.set_native_autograd(mode == "require")
        try:
            torch.manual_seed(7)
            for _ in range(3):
                solver.run_train_epoch()
        finally:
            autograd_ops.set_native_autograd(True)
        assert solver.fused_active == (mode == "require")
        if mode == "require":
            assert solver._fused_sys.program.loss == "custom" and solver._fused_sys.n_metrics == 2
        runs[mode] = (np.array(solver.metrics_history["train_loss"]), np.array(solver.metrics_history["train__energy"]),
                      np.array(solver.metrics_history["train__mean_u"]), R.get_flat(cfg["nets"]).cpu().numpy())
    got, want = runs["require"], runs["off"]
    errs = dict(loss=float(np.max(np.abs(got[0] - want[0]) / np.abs(want[0]))),
                energy=float(np.max(np.abs(got[1] - want[1]) / np.abs(want[1]))),
                mean_u=float(np.max(np.abs(got[2] - want[2]) / np.abs(want[2]))), params=rel_l2(got[3], want[3]))
    diag("custom_loss_solver", dict(errs, loss_hist=got[0].tolist()))
    assert errs["loss"] < 2e-5 and errs["energy"] < 2e-5 and errs["mean_u"] < 2e-5 and errs["params"] < 1e-5, errs


def test_epoch_dependent_custom_loss_leaves_the_fused_path_loudly():
    """A traced loss is frozen into the generated kernel.  A callable that follows solver state (here: a penalty weight
    that grows with ``solver.global_epoch``) is re-probed on its second use and periodically afterwards; when it traces
    to a different term the solver warns and continues on the reference's closure -- and therefore matches a run that
    never used the fused path."""
    from tests import configs

    def run(mode):
        torch.manual_seed(0)
        solver, cfg = configs.make_solver("c2", 8)
        solver.loss_fn = lambda r, f, x: (r ** 2).mean() * (1.0 + 0.5 * solver.global_epoch)
        solver.fused = mode
        torch.manual_seed(4)
        if mode == "auto":
            with pytest.warns(RuntimeWarning, match="changed between epochs"):
                for _ in range(4):
                    solver.run_train_epoch()
            assert not solver.fused_active
        else:
            for _ in range(4):
                solver.run_train_epoch()
        return np.array(solver.metrics_history["train_loss"])
    got, want = run("auto"), run("off")
    assert np.allclose(got, want, rtol=2e-5), (got, want)


def test_loss_outside_the_traced_family_falls_back_loudly():
    """A loss the tracer cannot express (batch sum) or user code that does something a traced column cannot: under
    fused='auto' the solver takes the composite path with a RuntimeWarning that names the cost; 'require' raises."""
    from tests import configs
    from neurodiffeq_amd import _lib
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2", 8, loss_fn=lambda r, f, x: (r ** 2).sum() / r.shape[0])
    with pytest.warns(RuntimeWarning, match="NOT on the fused MI355X path"):
        solver.run_train_epoch()
    assert not solver.fused_active and len(solver.metrics_history["train_loss"]) == 1
    # a per-point data column is an input row of the kernels since round 3: fused, and the same loss as the composite path
    losses = {}
    for mode in ("require", "off"):
        torch.manual_seed(0)
        data = torch.linspace(0, 1, 64, device="cuda").reshape(-1, 1)
        solver, cfg = configs.make_solver("c2", 8)
        solver.fused = mode
        solver.diff_eqs = lambda u, x, y: [configs.diff(u, x, order=2) + configs.diff(u, y, order=2) - data]
        torch.manual_seed(1)
        solver.run_train_epoch()
        assert solver.fused_active == (mode == "require")
        losses[mode] = solver.metrics_history["train_loss"][0]
    assert abs(losses["require"] - losses["off"]) <= 2e-5 * abs(losses["off"])
    # ... a matrix of data is not
    torch.manual_seed(0)
    table = torch.ones(64, 2, device="cuda")
    solver, cfg = configs.make_solver("c2", 8)
    solver.diff_eqs = lambda u, x, y: [configs.diff(u, x, order=2) + (configs.diff(u, y, order=2) * table).sum(dim=1, keepdim=True)]
    with pytest.warns(RuntimeWarning, match="NOT on the fused MI355X path"):
        solver.run_train_epoch()
    assert not solver.fused_active
    solver.fused = "require"
    solver._fused_key = None
    with pytest.raises(_lib.NdqError):
        solver.run_train_epoch()


def test_closure_based_optimizer_runs_on_the_fused_path():
    """LBFGS-style optimisers (solvers.py:397-400: ``optimizer.step(closure)`` once per batch) on the fused path: every
    closure call is one fused evaluation; same loss history as the reference's closure on torch autograd."""
    from tests import configs
    from neurodiffeq_amd import autograd_ops
    runs = {}
    for mode in ("require", "off"):
        torch.manual_seed(0)
        cfg = configs.make("c2", 16)
        nets = [n.to("cuda") for n in cfg["nets"]]
        opt = torch.optim.LBFGS([p for n in nets for p in n.parameters()], lr=0.5, max_iter=4, history_size=5)
        from neurodiffeq_amd.solvers import Solver2D
        with pytest.warns(RuntimeWarning):          # the reference's warning about n_batches_valid = 0 with LBFGS
            solver = Solver2D(cfg["pde"], cfg["conds"], xy_min=(0, 0), xy_max=(1, 1), nets=nets, optimizer=opt,
                              train_generator=cfg["gen"], valid_generator=cfg["gen"], n_batches_valid=0)
        solver.fused = mode
        autograd_ops.set_native_autograd(mode == "require")
        try:
            torch.manual_seed(7)
            for _ in range(3):
                solver.run_train_epoch()
        finally:
            autograd_ops.set_native_autograd(True)
        assert solver.fused_active == (mode == "require")
        runs[mode] = np.array(solver.metrics_history["train_loss"])
    err = float(np.max(np.abs(runs["require"] - runs["off"]) / np.abs(runs["off"])))
    diag("lbfgs", dict(err=err, fused=runs["require"].tolist(), autograd=runs["off"].tolist()))
    assert err < 1e-3 and runs["require"][-1] < runs["require"][0]


def test_one_multi_output_network_shared_by_several_conditions():
    """The reference's single_net / ith_unit mode (ode.py:276-280, conditions.py:54-55): ``nets=[net, net]`` with
    ``set_impose_on`` -- ONE parameter set, one stream array, one gradient on the fused path (shared networks
    were treated as two), also under weight decay, where a duplicated parameter set would decay twice."""
    from neurodiffeq_amd import diff, autograd_ops
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.optim import FusedAdam
    from neurodiffeq_amd.solvers import Solver1D
    runs = {}
    for mode in ("require", "off"):
        torch.manual_seed(0)
        net = FCNN(1, 2, hidden_units=(32, 32)).to("cuda")
        conds = [IVP(0.0, 1.5), IVP(0.0, 1.0)]
        with pytest.warns(DeprecationWarning):
            conds[0].set_impose_on(0)
            conds[1].set_impose_on(1)
        opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-2) if mode == "require" else \
            torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-2)
        solver = Solver1D(lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)], conds, t_min=0.1, t_max=4.0,
                          nets=[net, net], optimizer=opt, n_batches_valid=0)
        solver.fused = mode
        autograd_ops.set_native_autograd(mode == "require")
        try:
            torch.manual_seed(3)
            for _ in range(5):
                solver.run_train_epoch()
        finally:
            autograd_ops.set_native_autograd(True)
        assert solver.fused_active == (mode == "require")
        if mode == "require":
            assert len(solver._fused_sys.flat) == 1 and solver._fused_sys.descs[0].n_out == 2
        runs[mode] = (np.array(solver.metrics_history["train_loss"]), R.get_flat([net]).cpu().numpy())
    errs = dict(loss=float(np.max(np.abs(runs["require"][0] - runs["off"][0]) / np.abs(runs["off"][0]))),
                params=rel_l2(runs["require"][1], runs["off"][1]))
    diag("shared_net", errs)
    assert errs["loss"] < 2e-5 and errs["params"] < 1e-5, errs


def test_sobolev_loss_fused_for_first_and_second_order_systems():
    """loss_fn = 'h1' (losses.py:17-26): first-order systems trace with second-order streams, second-order PDEs with
    third-order streams (ndq_mlp_desc.mask3); both follow the autograd path's trajectory.  Round 6: h1 of a THIRD-order
    equation needs fourth-order streams (mask4) and is fused as well; a fifth-order requirement (h1 of a fourth-order
    equation) is refused."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP, NoCondition
    from neurodiffeq_amd.solvers import Solver1D, Solver2D

    def run(mode, kind):
        torch.manual_seed(0)
        s = Solver1D(lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)], [IVP(0.0, 1.5), IVP(0.0, 1.0)],
                     t_min=0.1, t_max=4.0, loss_fn=kind, n_batches_valid=1)
        s.fused = mode
        s.fit(25, tqdm_file=None)
        return s
    for kind in ("h1", "h1 semi"):
        a, b = run("require", kind), run("off", kind)
        assert a.fused_active and not b.fused_active
        assert np.allclose(a.metrics_history["train_loss"], b.metrics_history["train_loss"], rtol=3e-4), kind
        assert np.allclose(a.metrics_history["valid_loss"], b.metrics_history["valid_loss"], rtol=3e-4), kind
    def run2(mode):
        torch.manual_seed(0)
        s2 = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2) - u], [NoCondition()], xy_min=(0, 0),
                      xy_max=(1, 1), loss_fn="h1", n_batches_valid=0)
        s2.fused = mode
        s2.fit(5, tqdm_file=None)
        return s2
    a, b = run2("require"), run2("off")
    assert a.fused_active and not b.fused_active
    assert a._fused_sys.descs[0].mask3 == 0b1111          # xxx, xxy, xyy, yyy
    assert np.allclose(a.metrics_history["train_loss"], b.metrics_history["train_loss"], rtol=3e-4)
    def run3(mode):
        torch.manual_seed(0)
        s3 = Solver1D(lambda u, t: [diff(u, t, order=3) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=1.0, loss_fn="h1", n_batches_valid=0)
        s3.fused = mode
        s3.fit(5, tqdm_file=None)
        return s3
    a, b = run3("require"), run3("off")
    assert a.fused_active and not b.fused_active and a._fused_sys.descs[0].mask4 == 1
    assert np.allclose(a.metrics_history["train_loss"], b.metrics_history["train_loss"], rtol=3e-4)
    torch.manual_seed(0)
    s4 = Solver1D(lambda u, t: [diff(u, t, order=4) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=1.0, loss_fn="h1", n_batches_valid=0)
    s4.fused = "require"
    with pytest.raises(_lib_error()):
        s4.fit(1, tqdm_file=None)


def _lib_error():
    from neurodiffeq_amd import _lib
    return _lib.NdqError


def test_solver_with_l1_loss_stays_on_the_fused_path():
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.solvers import Solver1D

    def run(mode):
        torch.manual_seed(0)
        s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0, loss_fn="l1", n_batches_valid=1)
        s.fused = mode
        s.fit(30, tqdm_file=None)
        return s
    a, b = run("require"), run("off")
    assert a.fused_active and not b.fused_active
    assert np.allclose(a.metrics_history["train_loss"], b.metrics_history["train_loss"], rtol=2e-4)
    assert np.allclose(a.metrics_history["valid_loss"], b.metrics_history["valid_loss"], rtol=2e-4)


def test_spherical_solver_with_default_network_runs_fused():
    """SolverSpherical with its default FCNN(3, 1) (solvers.py:761-976 of the reference) trains on the d = 3 kernels."""
    from tests import zoo
    from neurodiffeq_amd.solvers import SolverSpherical
    torch.manual_seed(0)
    pde, conds = zoo.spherical_solver_problem()
    solver = SolverSpherical(pde, conds, 0.5, 2.0, n_batches_valid=1)
    solver.fused = "require"
    solver.fit(200, tqdm_file=None)
    assert solver.fused_active
    h = solver.metrics_history
    assert len(h["train_loss"]) == 200 and len(h["valid_loss"]) == 200
    assert h["train_loss"][-1] < 0.85 * h["train_loss"][0] and h["valid_loss"][-1] < 0.85 * h["valid_loss"][0]
    r = torch.full((5,), 0.5)
    th, ph = torch.linspace(0.3, 2.8, 5), torch.zeros(5)
    assert torch.allclose(solver.get_solution()(r, th, ph).cpu(), torch.cos(th), atol=1e-6)      # inner boundary exact


def test_static_host_batches_are_cached_and_invalidated():
    """A host batch that comes back unchanged (static validation sets) is uploaded once and read in place; an
    in-place change (torch version counter) or a new tensor is seen."""
    cfg, system = _load_system("c2", 16)
    torch.manual_seed(3)
    x, y = torch.rand(256), torch.rand(256)

    def loss_of(sysm, batch):
        sysm.step(batch, train=False, slot=0)
        return sysm.loss_buf[0].item()
    first = [loss_of(system, [x, y]) for _ in range(4)]
    assert len(system._static) == 1 and len(set(first)) == 1
    other = [torch.rand(256), torch.rand(256)]                  # a different batch of the same size in between
    l_other = loss_of(system, other)
    assert loss_of(system, [x, y]) == first[0] and l_other != first[0]
    x.mul_(0.5)                                                 # in-place edit: must be uploaded again
    edited = loss_of(system, [x, y])
    _, fresh = _load_system("c2", 16)
    assert edited != first[0] and edited == loss_of(fresh, [x.clone(), y.clone()])
    # ... and an edit through `.data` / a numpy view, which bumps no version counter (a callback shifting or rescaling a static
    # grid in place): seen by the first / last value that are part of a batch's identity
    assert loss_of(system, [x, y]) == edited
    v = y._version
    y.data.mul_(0.5)
    x.numpy()[:] += 0.25
    assert y._version == v
    again = loss_of(system, [x, y])
    assert again != edited and again == loss_of(fresh, [x.clone(), y.clone()])


def test_bundle_solver_trains_fused_and_matches_autograd_path():
    """BundleSolver1D (solvers.py:1189-1420): bundle inputs are extra network coordinates without derivative streams.
    The fused zero-sync fit and the reference's autograd closure (fused='off', same GPU) walk the same trajectory."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import BundleIVP
    from neurodiffeq_amd.solvers import BundleSolver1D

    def run(mode):
        torch.manual_seed(0)
        solver = BundleSolver1D(lambda u, t, lam: [diff(u, t) + lam * u],
                                [BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0})], t_min=0.0, t_max=1.0,
                                theta_min=(0.5, 0.5), theta_max=(2.0, 2.0), eq_param_index=(1,), n_batches_valid=1)
        solver.fused = mode
        solver.fit(40, tqdm_file=None)
        return solver

    a, b = run("require"), run("off")
    assert a.fused_active and not b.fused_active
    ha, hb = a.metrics_history, b.metrics_history
    assert np.allclose(ha["train_loss"], hb["train_loss"], rtol=2e-4) and np.allclose(ha["valid_loss"], hb["valid_loss"], rtol=2e-4)
    t, u0, lam = torch.rand(50), 0.5 + 1.5 * torch.rand(50), 0.5 + 1.5 * torch.rand(50)
    ua, ub = a.get_solution()(t, u0, lam).cpu(), b.get_solution()(t, u0, lam).cpu()
    assert torch.allclose(ua, ub, rtol=1e-3, atol=1e-4)
    assert torch.allclose(a.get_solution()(torch.zeros(50), u0, lam).cpu(), u0, atol=1e-6)        # u(0; u0, lam) = u0


def test_checkpoint_and_resume_of_a_fused_solver():
    """The reference's resume recipe (README: dill the internals, rebuild a solver around the loaded networks and
    optimiser; callbacks.py:129-155, solvers_utils.py:281-398): the fused path's flat parameters, device-side Adam
    moments and step counts must survive the round trip -- resumed training continues the uninterrupted trajectory."""
    import dill
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.generators import Generator1D
    from neurodiffeq_amd.solvers import Solver1D
    ode = lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)]
    conds = [IVP(0.0, 1.5), IVP(0.0, 1.0)]

    def make(nets=None, optimizer=None):
        s = Solver1D(ode, conds, t_min=0.1, t_max=4.0, nets=nets, optimizer=optimizer, n_batches_valid=1,
                     train_generator=Generator1D(64, 0.1, 4.0, "equally-spaced"), valid_generator=Generator1D(64, 0.1, 4.0, "equally-spaced"))
        s.fused = "require"
        return s
    torch.manual_seed(0)
    whole = make()
    whole.fit(20, tqdm_file=None)
    torch.manual_seed(0)
    first = make()
    first.fit(10, tqdm_file=None)
    blob = dill.dumps(first.get_internals(["nets", "optimizer", "global_epoch", "lowest_loss"], return_type="dict"))
    state = dill.loads(blob)
    resumed = make(nets=state["nets"], optimizer=state["optimizer"])
    resumed.fit(10, tqdm_file=None)
    assert resumed.fused_active
    a = np.array(whole.metrics_history["train_loss"][10:])
    b = np.array(resumed.metrics_history["train_loss"])
    assert np.allclose(a, b, rtol=1e-5), (a, b)
    pa, pb = R.get_flat(whole.nets).cpu().numpy(), R.get_flat(resumed.nets).cpu().numpy()
    assert rel_l2(pb, pa) < 1e-5
    # the other route: state_dict round trip into a live solver (torch idiom), then keep training natively
    torch.manual_seed(0)
    again = make()
    again.fit(10, tqdm_file=None)
    sd = dill.loads(dill.dumps(again.optimizer.state_dict()))
    again.optimizer.load_state_dict(sd)
    again.fit(10, tqdm_file=None)
    assert getattr(again._fused_sys, "_fast", None) is not None
    assert np.allclose(np.array(again.metrics_history["train_loss"][10:]), a, rtol=1e-5)


def test_batch_size_that_changes_every_epoch():
    """FilterGenerator (generators.py:904-952) keeps a different number of points each draw: the engine's per-size
    buffer sets are capped, and every epoch's loss is the loss of that epoch's points (checked through the composite path)."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import NoCondition
    from neurodiffeq_amd.generators import FilterGenerator, Generator2D
    from neurodiffeq_amd.solvers import Solver2D

    def run(mode):
        torch.manual_seed(0)
        gen = FilterGenerator(Generator2D((24, 24), (0, 0), (1, 1), "equally-spaced-noisy"),
                              lambda xs: (xs[0] - 0.5) ** 2 + (xs[1] - 0.5) ** 2 < 0.2)
        s = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2) - u], [NoCondition()], xy_min=(0, 0),
                     xy_max=(1, 1), train_generator=gen, valid_generator=gen, n_batches_valid=0)
        s.fused = mode
        torch.manual_seed(1)
        s.fit(30, tqdm_file=None)
        return s
    a, b = run("require"), run("off")
    assert a.fused_active and len(a._fused_sys._bufs) <= a._fused_sys.MAX_BUFFER_SETS
    assert np.allclose(a.metrics_history["train_loss"], b.metrics_history["train_loss"], rtol=3e-4)


def test_gradient_accumulation_and_validation_mode():
    """n_batches_train = 2 accumulates gradients before one step (solvers.py:360-419); a validation epoch leaves
    parameters and gradients untouched."""
    from tests import configs
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2", 16, n_batches_train=2, n_batches_valid=1)
    solver.fused = "require"
    torch.manual_seed(0)
    ocfg = R.build_config("c2", 16)
    loop = R.TrainLoop(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], ocfg["sampler"], n_batches=2)
    torch.manual_seed(7)
    loop.epoch()
    torch.manual_seed(7)
    solver.run_train_epoch()
    before = R.get_flat(cfg["nets"]).cpu().numpy().copy()
    solver.run_valid_epoch()
    assert np.array_equal(before, R.get_flat(cfg["nets"]).cpu().numpy())
    assert len(solver.metrics_history["valid_loss"]) == 1
    assert abs(solver.metrics_history["train_loss"][0] - loop.history[0]) <= 2e-5 * abs(loop.history[0])
    assert rel_l2(before, R.get_flat(ocfg["nets"]).numpy()) < 1e-5


def test_native_epoch_path_bookkeeping_matches_general_path():
    """The zero-sync native epoch (closure -> fused sums -> device-side tail) against the general path on the same
    seeds: loss history, lowest_loss, best_nets snapshot (pre-step parameters of the best epoch), final parameters,
    optimizer step count."""
    from tests import configs

    def run(native):
        torch.manual_seed(0)
        # a (no-op) metric forces the general path
        solver, cfg = configs.make_solver("c2", 16, metrics=None if native else {"zero": lambda u, x, y: (u * 0).mean()})
        solver.fused = "require"
        torch.manual_seed(5)
        for _ in range(6):
            solver.run_train_epoch()
        assert solver.fused_active
        hist = list(solver.metrics_history["train_loss"])
        best = R.get_flat(solver.best_nets).cpu().numpy()
        return solver, hist, solver.lowest_loss, best, R.get_flat(cfg["nets"]).cpu().numpy()

    s1, h1, l1, b1, p1 = run(True)
    s2, h2, l2, b2, p2 = run(False)
    assert s1._fused_sys._fast is not None and getattr(s2._fused_sys, "_fast", None) is None
    assert len(h1) == len(h2) == 6 and s1.global_epoch == 6
    assert np.allclose(h1, h2, rtol=2e-5)
    assert abs(l1 - min(h1)) <= 1e-7 * abs(l1) and abs(l1 - l2) <= 2e-5 * abs(l2)
    assert rel_l2(b1, b2) < 1e-5 and rel_l2(p1, p2) < 1e-5
    assert int(s1.optimizer.state_dict()["state"][0]["step"]) == 6


@pytest.mark.parametrize("name", ["c2", "c1"])
def test_fp64_epochs_stay_on_the_device_and_match_the_general_path(name):
    """fp64 networks (the reference's default precision, neurodiffeq/__init__.py:22): the three-kernel pipeline in double
    followed by the device-side epoch tail in double (ndq64_epoch_tail) -- no host synchronisation per epoch -- against the
    general path (host-side history, FusedAdam.step through ndq64_adam_step) on the same seeds.  Same kernels, same Adam
    arithmetic: histories, best snapshot, parameters and step counts agree to rounding of the loss mean."""
    from tests import configs

    def run(native):
        torch.manual_seed(0)
        solver, cfg = configs.make_solver(name, SIZES[name], n_batches_train=2, n_batches_valid=2,
                                          metrics=None if native else {"zero": lambda *a: (a[0] * 0).mean()})
        for net in cfg["nets"]:
            net.double()
        solver.fused = "require"
        torch.manual_seed(5)
        solver.run_train_epoch()
        solver.run_valid_epoch()
        solver.fit(3, tqdm_file=None)
        assert solver.fused_active and solver._fused_sys.f64
        h = solver.metrics_history
        return (solver, list(h["train_loss"]), list(h["valid_loss"]), solver.lowest_loss,
                R.get_flat(solver.best_nets).cpu().numpy(), R.get_flat(cfg["nets"]).cpu().numpy())

    s1, t1, v1, l1, b1, p1 = run(True)
    s2, t2, v2, l2, b2, p2 = run(False)
    assert getattr(s1._fused_sys, "_fast", None) is not None and getattr(s2._fused_sys, "_fast", None) is None
    assert s1._fused_sys._fast["loss_hist"].dtype == torch.float64
    assert len(t1) == len(v1) == 4 and s1.global_epoch == 4
    assert np.allclose(t1, t2, rtol=1e-13, atol=0) and np.allclose(v1, v2, rtol=1e-13, atol=0)
    assert abs(l1 - min(v1)) <= 1e-15 * abs(l1) and abs(l1 - l2) <= 1e-13 * abs(l2)
    assert b1.dtype == np.float64 and rel_l2(b1, b2) < 1e-13 and rel_l2(p1, p2) < 1e-13
    steps = {int(st["step"]) for st in s1.optimizer.state_dict()["state"].values()}
    assert steps == {4}


@pytest.mark.parametrize("name", ["c2", "c1"])
def test_native_fit_with_validation_matches_general_path(name):
    """fit() with validation epochs (best network chosen by the validation loss, solvers.py:414-415) on the zero-sync
    path -- single-launch (c2) and multi-network pipeline (c1) -- against the general host-synchronising path."""
    from tests import configs

    def run(native):
        torch.manual_seed(0)
        solver, cfg = configs.make_solver(name, SIZES[name], n_batches_train=2, n_batches_valid=2,
                                          metrics=None if native else {"zero": lambda *a: (a[0] * 0).mean()})
        solver.fused = "require"
        torch.manual_seed(5)
        solver.fit(4, tqdm_file=None)
        h = solver.metrics_history
        return (solver, list(h["train_loss"]), list(h["valid_loss"]), solver.lowest_loss,
                R.get_flat(solver.best_nets).cpu().numpy(), R.get_flat(cfg["nets"]).cpu().numpy())

    s1, t1, v1, l1, b1, p1 = run(True)
    s2, t2, v2, l2, b2, p2 = run(False)
    assert getattr(s1._fused_sys, "_fast", None) is not None and getattr(s2._fused_sys, "_fast", None) is None
    assert len(t1) == len(v1) == 4
    assert np.allclose(t1, t2, rtol=3e-5) and np.allclose(v1, v2, rtol=3e-5)
    assert abs(l1 - min(v1)) <= 1e-6 * abs(l1) and abs(l1 - l2) <= 3e-5 * abs(l2)
    assert rel_l2(b1, b2) < 2e-5 and rel_l2(p1, p2) < 2e-5


def test_default_configuration_fit_matches_general_path():
    """The reference's default set-up (Solver1D: 32 noisy training points, the SAME 'equally-spaced' validation grid
    served n_batches_valid = 4 times per epoch): the zero-sync path evaluates the static grid once per epoch and must
    report what the general path reports after evaluating it four times."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.solvers import Solver1D

    def run(native):
        torch.manual_seed(0)
        solver = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], t_min=0.0, t_max=2.0,
                          metrics=None if native else {"zero": lambda u, t: (u * 0).mean()})
        solver.fused = "require"
        solver.fit(30, tqdm_file=None)
        h = solver.metrics_history
        return solver, list(h["train_loss"]), list(h["valid_loss"]), solver.lowest_loss

    s1, t1, v1, l1 = run(True)
    s2, t2, v2, l2 = run(False)
    assert getattr(s1._fused_sys, "_fast", None) is not None and getattr(s2._fused_sys, "_fast", None) is None
    assert len(s1._fused_sys._static) == 1 and len(v1) == 30
    assert np.allclose(t1, t2, rtol=3e-5) and np.allclose(v1, v2, rtol=3e-5) and abs(l1 - l2) <= 3e-5 * abs(l2)


@pytest.mark.parametrize("name", ["c1", "c2", "c4"])
def test_solution_and_residuals_on_forward_kernels(name):
    """solver.get_solution()(coords) and solver.get_residuals(coords) (solvers.py:606-646, 682-720) run on the
    forward-only kernels + generated pointwise kernel and agree with the autograd oracle in fp64."""
    from tests import configs
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, SIZES[name])
    solver.fused = "require"
    torch.manual_seed(0)
    ocfg = R.build_config(name, SIZES[name], dtype=torch.float64)
    R.set_flat(ocfg["nets"], R.get_flat(cfg["nets"]).cpu().double())
    torch.manual_seed(9)
    ex = cfg["gen"].get_examples()
    coords = [ex.detach()] if isinstance(ex, torch.Tensor) else [c.detach() for c in ex]
    out = R.closure(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], [c.double() for c in coords], backward=False)
    kw = dict(harmonics_fn=configs.RealSphericalHarmonics(4)) if name == "c4" else {}
    sol = solver.get_solution(best=False, **kw)
    us = sol(*[c.cuda() for c in coords], to_numpy=True)
    us = us if isinstance(us, list) else [us]
    assert getattr(sol, "_eval_sys", None) is not None, "solution evaluation did not take the fused path"
    assert rel_l2(np.stack([u.reshape(-1) for u in us], axis=1), out["funcs"].numpy()) < TOL
    rs = solver.get_residuals(*coords, best=False, to_numpy=True)
    rs = rs if isinstance(rs, list) else [rs]
    assert getattr(solver, "_resid_sys", None) is not None
    assert rel_l2(np.stack([r.reshape(-1) for r in rs], axis=1), out["residuals"].numpy()) < TOL
    assert us[0].shape == tuple(coords[0].shape)


def test_solution_and_residuals_follow_python_state_changed_after_their_first_use():
    """The reference evaluates ``diff_eqs`` and ``cond.enforce`` afresh on every ``get_residuals`` / solution call
    (solvers.py:606-646, 682-720).  The fused evaluation paths cache a traced system: it is re-probed on every use (round 5 --
    the same silent-stale hazard as VERDICT r4 weak #1, on the evaluation side), numbers that moved become runtime constants
    of ONE rebuild.  Checked against the composite path (the reference's closure on torch autograd) of the same solver."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver1D
    torch.manual_seed(3)
    k = {"v": 1.0}
    cond = IVP(0.0, 1.0)
    solver = Solver1D(lambda u, t: [diff(u, t) + k["v"] * u], [cond], t_min=0.0, t_max=2.0, nets=[FCNN(1, 1, hidden_units=(32, 32))])
    solver.fused = "require"
    ts = torch.linspace(0.1, 1.9, 257).reshape(-1, 1)

    def both():
        solver.fused = "require"
        r = solver.get_residuals(ts, best=False, to_numpy=True)
        u = solver.get_solution(best=False)(ts.cuda(), to_numpy=True)
        sysm = solver._resid_sys
        solver.fused = "off"
        r_ref = solver.get_residuals(ts, best=False, to_numpy=True)
        solver.fused = "require"
        return r, u, r_ref, sysm
    r0, u0, ref0, s0 = both()
    assert s0 is not None and rel_l2(r0, ref0) < TOL
    k["v"] = 3.0                                        # a coefficient of the equation ...
    r1, u1, ref1, s1 = both()
    assert rel_l2(r1, ref1) < TOL and rel_l2(r1, r0) > 1e-2
    assert s1 is not s0 and s1.theta_frozen             # rebuilt once, the coefficient is a kernel argument now
    k["v"] = 0.25
    r2, u2, ref2, s2 = both()
    assert rel_l2(r2, ref2) < TOL and s2 is s1          # ... further values: argument updates of the same kernels
    sol = solver.get_solution(copy=False, best=False)   # (copy=False: the solution shares the solver's condition objects)
    ua = sol(ts.cuda(), to_numpy=True)
    cond.u_0 = 2.5                                      # ... and a boundary value stored on the condition object
    ub = sol(ts.cuda(), to_numpy=True)
    assert abs(float(sol(torch.zeros(1, 1).cuda(), to_numpy=True).reshape(-1)[0]) - 2.5) < 1e-6
    assert np.max(np.abs(ub - ua)) > 0.5


def test_pk_mfma_hazard_is_fixed_up_by_the_build():
    """neurodiffeq_amd/csrc/canary_pk_war.hip replays, with hard-coded registers, the instruction sequence that made one closure
    kernel's dW1 non-deterministic on gfx950 (packed-fp32 VALU op directly followed by a bf16 MFMA: lanes 48..63 of the
    packed op's low half come out wrong).  Built through the package's own pipeline (_hipcc.compile_shared: the
    assembly fix-up pass separates the pair) it must be exact; built with the pass switched off it documents whether
    this machine shows the hazard (recorded, not asserted)."""
    import ctypes, os, tempfile
    from neurodiffeq_amd import _hipcc
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neurodiffeq_amd", "csrc", "canary_pk_war.hip")
    counts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for label, off in (("fixed", "0"), ("raw", "1")):
            os.environ["NDQ_NO_PK_MFMA_FIX"] = off
            try:
                so = os.path.join(tmp, f"pk_war_{label}.so")
                sites = _hipcc.compile_shared(src, so, ["-DNDQ_PK_WAR_LIB=1", "-Wno-unused-value"])
            finally:
                os.environ.pop("NDQ_NO_PK_MFMA_FIX", None)
            lib = ctypes.CDLL(so)
            lib.ndq_pk_war_count.restype = ctypes.c_long
            counts[label] = int(lib.ndq_pk_war_count(200))
            counts[label + "_sites"] = sites
    diag("pk_mfma_hazard", counts)
    assert counts["fixed_sites"] > 0
    assert counts["fixed"] == 0, counts


def test_hazard_canary_detects_the_unfixed_kernel_and_passes_the_fixed_one():
    """the canary that runs at first use on every box (neurodiffeq_amd/_canary.py).  The
    reproducer built THROUGH the assembly fix-up pass counts zero wrong results; built from the compiler's unmodified output
    it shows the hazard (the committed signature, profiles/archive/r01/r01u_pk_mfma_hazard.txt); a spilling adjoint kernel with two
    waves per SIMD is bit-reproducible over 30 launches through the pass.  A canary that fails refuses the 8-wave builds."""
    import warnings
    from neurodiffeq_amd import _canary
    from neurodiffeq_amd.engine import FusedSystem
    _canary.STATUS.update(checked=False, fixed=None, raw=None, refuse_two_waves=False)
    st = _canary.check(iters=200)
    assert st["fixed"] == 0 and not st["refuse_two_waves"], st
    assert st["raw"] > 0, f"this box does not show the packed-fp32 -> MFMA hazard with the pass switched off: {st}"
    two = _canary.check_two_waves()
    diag("hazard_canary", dict(st, two_waves=two))
    assert two["fixed"] == 0 and two["fixed_waves"] == 8, two
    # a failing canary is loud and switches the two-waves-per-SIMD builds off
    _canary.STATUS.update(checked=False)
    fixed_so, raw_so = _canary._so("fixed"), _canary._so("raw")
    orig = _canary._so
    try:
        _canary._so = lambda label: raw_so            # pretend the pass had not fixed the kernel
        with pytest.warns(RuntimeWarning, match="does not cover"):
            bad = _canary.check(iters=200, rebuild=False)
        assert bad["refuse_two_waves"]
        from tests import configs
        torch.manual_seed(0)
        cfg = configs.make("c2", 16)
        for net in cfg["nets"]:
            net.to("cuda")
        fs = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cuda")
        assert fs.fusedk_wide is False
    finally:
        _canary._so = orig
        _canary.STATUS.update(checked=False, refuse_two_waves=False)
        _canary.check()


def test_closure_kernel_self_check_accepts_good_and_rejects_bad_kernels():
    """engine.verify_fused: the first training use of a system's single-launch closure kernel runs it twice plus the
    three-kernel pipeline; a healthy kernel is accepted (bit-reproducible, gradients equal to ~1e-6), one whose
    gradients change from launch to launch is rejected with a warning and the system carries on, correctly, on the
    pipeline."""
    from tests import zoo
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(11)
    system = zoo.build("helmholtz_xy")
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(3001, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.to("cuda")
    good = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=True)
    good.step([c.float() for c in coords], train=True, slot=0)
    torch.cuda.synchronize()
    assert good.fusedk is not None and good.fused_check["reproducible"] and good.fused_check["pipeline_reproducible"]
    assert good.fused_check["grad_rel_l2"] < 1e-5 and good.fused_check["loss_rel"] < 1e-5, good.fused_check

    bad = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=True)
    healthy = bad.fused_closure
    calls = [0]

    def flaky(b, n, stream, train, *a, **k):          # a kernel whose dW1[13][0] differs from launch to launch
        healthy(b, n, stream, train, *a, **k)
        calls[0] += 1
        if train:
            bad.flat[0].grad[39] += 1e-3 * calls[0]
    bad.fused_closure = flaky
    with pytest.warns(RuntimeWarning, match="self-check"):
        b, n = bad.step([c.float() for c in coords], train=True, slot=0)
    torch.cuda.synchronize()
    assert bad.fusedk is None and not bad.fused_check["reproducible"]
    errs = dict(loss=abs(bad.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(np.concatenate([fp.grad.cpu().numpy() for fp in bad.flat]), want_grad))
    diag("self_check", dict(good=good.fused_check, bad=bad.fused_check, after=errs))
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", ["poisson3d", "helmholtz_xy", "heat_wide", "coupled_sin", "swish_laplace"])
def test_closure_kernel_launches_are_bit_reproducible(name):
    """50 launches of a system's single-launch closure kernel on identical inputs: 50 bit-identical blocks of partial
    sums and loss partials (poisson3d is the kernel in which the packed-VALU -> MFMA hazard of DESIGN 4.6 surfaced as one
    non-deterministic dW1 entry)."""
    from tests import zoo
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    coords = [c.float() for c in system.sample(3001, seed=5)]
    for net in nets:
        net.to("cuda")
    fs = FusedSystem(nets, conds, pde, system.n_coords, "cuda", single_kernel=True)
    assert fs.fusedk is not None
    first = None
    for rep in range(50):
        b, n = fs.step(coords, train=True, slot=0)
        snap = torch.cat([p.reshape(-1) for p in b["fused_partials_all"]] + [b["fused_loss_partials"].reshape(-1)]).clone()
        if first is None:
            first = snap
        else:
            assert torch.equal(snap, first), f"launch {rep} differs in {(snap != first).sum().item()} entries"
    assert fs.fusedk is not None and fs.fused_check["reproducible"]


@pytest.mark.parametrize("kind", ["sgd_momentum", "rmsprop", "adamw", "adam_clipped"])
def test_torch_optimizers_and_step_overrides_on_the_fused_path(kind):
    """a10 of SURVEY 8(a): ``optimizer=`` may be any torch.optim optimiser over the networks' parameters and users
    override ``_do_optimizer_step`` (gradient clipping, solvers.py:331-341).  The fused closure leaves ordinary ``.grad``
    tensors behind (views of the flat gradient buffer), so all of that keeps working: five epochs against the autograd
    restatement driven by the same optimiser."""
    from itertools import chain
    from tests import configs
    from neurodiffeq_amd.solvers import Solver2D

    def make_opt(params):
        params = list(params)
        if kind == "sgd_momentum":
            return torch.optim.SGD(params, lr=1e-2, momentum=0.9, nesterov=True)
        if kind == "rmsprop":
            return torch.optim.RMSprop(params, lr=1e-3, alpha=0.9)
        if kind == "adamw":
            return torch.optim.AdamW(params, lr=2e-3, weight_decay=0.05)
        return torch.optim.Adam(params, lr=1e-3)

    class Clipped(Solver2D):
        def _do_optimizer_step(self, closure=None):
            torch.nn.utils.clip_grad_norm_(list(chain.from_iterable(n.parameters() for n in self.nets)), 0.05)
            self.optimizer.step()

    torch.manual_seed(0)
    cfg = configs.make("c2", 16)
    for net in cfg["nets"]:
        net.to("cuda")
    cls = Clipped if kind == "adam_clipped" else Solver2D
    solver = cls(cfg["pde"], cfg["conds"], xy_min=cfg["dom"][0], xy_max=cfg["dom"][1], nets=cfg["nets"],
                 train_generator=cfg["gen"], valid_generator=cfg["gen"], n_batches_valid=0,
                 optimizer=make_opt(chain.from_iterable(n.parameters() for n in cfg["nets"])))
    solver.fused = "require"
    torch.manual_seed(0)
    ocfg = R.build_config("c2", 16)
    loop = R.TrainLoop(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], ocfg["sampler"])
    loop.opt = make_opt(chain.from_iterable(n.parameters() for n in ocfg["nets"]))
    if kind == "adam_clipped":
        plain_step = loop.opt.step

        def clipped_step():
            torch.nn.utils.clip_grad_norm_(list(chain.from_iterable(n.parameters() for n in ocfg["nets"])), 0.05)
            plain_step()
        loop.opt.step = clipped_step
    torch.manual_seed(7)
    for _ in range(5):
        loop.epoch()
    torch.manual_seed(7)
    for _ in range(5):
        solver.run_train_epoch()
    assert solver.fused_active
    errs = dict(params=rel_l2(R.get_flat(cfg["nets"]).cpu().numpy(), R.get_flat(ocfg["nets"]).numpy()),
                loss=max(abs(a - b) / abs(b) for a, b in zip(solver.metrics_history["train_loss"], loop.history)))
    diag(f"optimizer_{kind}", errs)
    assert errs["params"] < 1e-5 and errs["loss"] < 2e-5, errs


def test_device_generator_prefetch_draws_the_same_batches_without_sampler_launches():
    """DeviceGenerator(prefetch=True): the next batch is drawn by extra workgroups of the epoch's sums + tail kernel.
    Same seed -> the same batches -> bit-identical training as with its own sampler launches; after the first epoch the
    generator launches nothing any more; whoever else asks for a batch still gets the right one."""
    from tests import configs
    from neurodiffeq_amd.generators import DeviceGenerator, SamplerGenerator

    def run(prefetch):
        torch.manual_seed(0)
        solver, cfg = configs.make_solver("c2", 32)
        solver.fused = "require"
        gen = DeviceGenerator(cfg["gen"], seed=11, prefetch=prefetch)
        solver.generator["train"] = SamplerGenerator(gen)
        for _ in range(8):
            solver.run_train_epoch()
        nxt = [c.clone() for c in gen.get_examples()]           # draw 8, asked for from outside the solver
        return solver, gen, list(solver.metrics_history["train_loss"]), R.get_flat(cfg["nets"]).cpu().numpy(), nxt

    s0, g0, h0, p0, n0 = run(False)
    s1, g1, h1, p1, n1 = run(True)
    assert s1._fused_sys._fast is not None
    assert g0.launches == 9 and g1.launches == 1, (g0.launches, g1.launches)
    assert h0 == h1 and np.array_equal(p0, p1)
    assert all(torch.equal(a, b) for a, b in zip(n0, n1))
    # the prefetched block is the generator's own: a second consumer in between simply re-draws what it needs
    ref = DeviceGenerator(configs.make("c2", 32)["gen"], seed=11)
    for _ in range(9):
        want = [c.clone() for c in ref.get_examples()]
    assert all(torch.equal(a, b) for a, b in zip(n1, want))


def test_opt_in_bf16_forward_mode_is_bounded_and_off_by_default(tmp_path):
    """BASELINE config 5 names "bf16 fwd / fp32 grad": the opt-in library built with -DNDQ_FWD_BF16X1=1 (hidden-layer GEMMs
    of the forward STREAM kernel on single bf16 operands; the adjoint kernels keep their bf16x3 forward pass) runs C5's
    closure within bf16-class distance of the default library -- and differs from it, i.e. the default is untouched
    (scripts/bf16_forward.py; measured at size: function values 4e-4, gradient 3e-4, closure 4.81 -> 4.26 ms)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "bf16_forward.py")
    outs = []
    for flags in ("", "-DNDQ_FWD_BF16X1=1"):
        env = dict(os.environ, NDQ_LIB_FLAGS=flags)
        env.pop("NDQ_JIT_FLAGS", None)
        out = str(tmp_path / f"c5{'x1' if flags else ''}.npz")
        r = subprocess.run([sys.executable, script, "run", out, "c5:48"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert str(a["lib"]) == "libndq.so" and str(b["lib"]).startswith("libndq_")
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y.astype(np.float64)) / np.linalg.norm(y.astype(np.float64))
    errs = dict(funcs=rel(b["funcs"], a["funcs"]), resid=rel(b["resid"], a["resid"]), grad=rel(b["grad"], a["grad"]))
    diag("bf16_forward_c5", errs)
    assert 1e-5 < errs["funcs"] < 5e-3 and errs["resid"] < 5e-3 and 1e-6 < errs["grad"] < 5e-3, errs
