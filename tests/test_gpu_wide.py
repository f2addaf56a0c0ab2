"""Networks WIDER than 64 hidden units on the HIP path (VERDICT r3 missing #1): the reference's own headline shapes --
README.md:125 ``FCNN(2, 1, hidden_units=(512,))``, the lid-driven-cavity notebooks' 256 / 512 wide three-output networks,
``tests/test_pde.py:377`` ``(100, 100)``.

One hidden layer (csrc/ndq_wide.h: units over lanes, weights in registers): stream kernels through the C-ABI against the
numpy jet oracle for many widths / stream sets / activations / output counts, closures against the goldens the UNMODIFIED
reference produced (w16, w17: tests/golden/make_golden.py) in both launch modes, solver trajectories, ``fit()``."""
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
TOL = 1e-5
DIAG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "diag")
ACT_ID = {"tanh": 0, "sin": 1, "sigmoid": 2, "swish": 3, "aptx": 4, "elu": 5, "softplus": 6, "gelu": 7}

# stream sets: name -> (multi-indices in the kernels' stream order, first, mask2, lap, mask3) for d inputs
def _streams(d, kind):
    pairs = [(a, b) for a in range(d) for b in range(a, d)]
    trips = [(a, b, c) for a in range(d) for b in range(a, d) for c in range(b, d)]
    first = [(a,) for a in range(d)]
    if kind == "value":
        return [()], 0, 0, 0, 0
    if kind == "first":
        return [()] + first, 1, 0, 0, 0
    if kind == "full2":
        return [()] + first + pairs, 1, (1 << len(pairs)) - 1, 0, 0
    if kind == "diag2":
        m = sum(1 << k for k, (a, b) in enumerate(pairs) if a == b)
        return [()] + first + [p for p in pairs if p[0] == p[1]], 1, m, 0, 0
    if kind == "lap":
        m = sum(1 << k for k, (a, b) in enumerate(pairs) if a == b)
        return [()] + first + ["lap"], 1, m, 1, 0
    if kind == "first+xx":      # value, gradient, d2/dx0^2
        return [()] + first + [(0, 0)], 1, 1, 0, 0
    if kind == "lap3":          # three-output network with a Laplacian stream is not offered by the tracer (lap needs n_out = 1): full2
        return [()] + first + pairs, 1, (1 << len(pairs)) - 1, 0, 0
    if kind == "full3":
        return [()] + first + pairs + trips, 1, (1 << len(pairs)) - 1, 0, (1 << len(trips)) - 1
    raise KeyError(kind)


# (d, width, n_out, activation, stream set)
SHAPES = [
    (2, 512, 1, "tanh", "lap"),          # README Laplace
    (2, 512, 3, "tanh", "full2"),        # one three-output network for (u, v, p)
    (2, 512, 1, "tanh", "full2"),
    (1, 512, 1, "sin", "full3"),
    (2, 300, 1, "sigmoid", "diag2"),     # 64 unit lanes, ragged (5 units per lane, the last row partly padding)
    (2, 256, 2, "tanh", "full2"),
    (3, 200, 1, "tanh", "lap"),          # 32 unit lanes x 2 point lanes
    (3, 129, 2, "swish", "full2"),
    (2, 128, 1, "aptx", "full2"),        # 16 unit lanes x 4 point lanes
    (2, 100, 1, "tanh", "full2"),        # tests/test_pde.py:377's width
    (1, 65, 1, "tanh", "full3"),
    (2, 96, 4, "sin", "first"),
    (4, 80, 1, "tanh", "value"),
    (2, 65, 8, "tanh", "full2"),         # 48 stream x output rows per point: four-point reduction passes
]
# deep networks wider than 64 units (csrc/ndq_deep.h: layer by layer through HBM): (d, width, n_out, activation, stream set, layers)
DEEP_SHAPES = [
    (2, 128, 1, "tanh", "first+xx", 3),  # w18's shape and stream set (Burgers: u_t, u_x, u_xx)
    (2, 256, 3, "tanh", "full2", 2),     # the RE100 notebook's network as FCNN builds it
    (2, 100, 1, "tanh", "lap", 2),       # tests/test_pde.py:377's (100, 100)
    (2, 512, 3, "tanh", "lap3", 2),      # the RE400 notebook's network
    (1, 65, 1, "sin", "full3", 2),
    (3, 80, 2, "sigmoid", "full2", 4),
    (2, 144, 1, "swish", "diag2", 2),
    (2, 96, 1, "aptx", "first", 5),
    (4, 72, 1, "tanh", "value", 2),
    # per-layer widths (hidden_units = (128, 64) ...: laid out for the widest layer, DeepCfg::WP): the width entry is the tuple
    (2, (128, 64), 1, "tanh", "first+xx", 2),
    (2, (64, 128), 1, "tanh", "lap", 2),
    (2, (96, 200, 40), 3, "sigmoid", "full2", 3),       # sigma(0) != 0 in the padding units
    (1, (256, 128, 72), 1, "sin", "full3", 3),
    (3, (100, 80), 2, "swish", "full2", 2),
]
# torch's own activation modules (nn.ELU / nn.Softplus / nn.GELU: generic derivative tables) on all three
# kernel families: fragment kernels (width <= 64), one wide layer, deep wide
ACT_SHAPES = [
    (2, 32, 1, "elu", "full2", 2), (2, 32, 1, "softplus", "full2", 2), (2, 32, 1, "gelu", "lap", 2), (1, 32, 1, "gelu", "full3", 2),
    (1, 48, 1, "softplus", "full3", 2), (2, 64, 1, "elu", "lap", 3), (2, 512, 1, "softplus", "lap", 1), (2, 200, 2, "gelu", "full2", 1),
    (1, 300, 1, "elu", "full3", 1), (2, 100, 1, "elu", "lap", 2), (2, 128, 1, "gelu", "first+xx", 2),
]
SHAPES = SHAPES + DEEP_SHAPES + ACT_SHAPES
IDS = ["-".join(map(str, sh)) for sh in SHAPES]


def shape_desc(shape):
    """(ndq_mlp_desc, widths of the hidden layers) of a SHAPES entry (the width entry is an int, or the tuple of per-layer widths)"""
    from neurodiffeq_amd import _lib
    d, w, n_out, act, kind = shape[:5]
    layers = shape[5] if len(shape) > 5 else 1
    ws = tuple(w) if isinstance(w, tuple) else (w,) * layers
    packed = sum(v << (10 * i) for i, v in enumerate(ws)) if isinstance(w, tuple) else 0
    _, first, mask2, lap, mask3 = _streams(d, kind)
    return _lib.MlpDesc(d, first, mask2, max(ws), layers, ACT_ID[act], n_out, lap, 0, mask3, 0, packed), ws


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _oracle_streams(flat64, dims, act, c64, streams):
    """jet oracle values per kernel stream; "lap" = sum of the diagonal second derivatives"""
    d = dims[0]
    need = [m for m in streams if m != "lap"] + ([(a, a) for a in range(d)] if "lap" in streams else [])
    need = list(dict.fromkeys(need))
    vals = J.mlp_jets(flat64, dims, act, c64, need)
    return {m: (sum(vals[(a, a)] for a in range(d)) if m == "lap" else vals[m]) for m in streams}


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4099])
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_wide_stream_kernels_match_jet_oracle(shape, n):
    from neurodiffeq_amd import _lib, codegen
    d, w, n_out, act, kind = shape[:5]
    desc, ws = shape_desc(shape)                                        # (ws: widths of the hidden layers)
    w = max(ws)
    if n not in (17, 1000) and shape not in SHAPES[:3] + SHAPES[6:7] + SHAPES[9:10] + DEEP_SHAPES[:3]:
        pytest.skip("edge sizes on a subset of the shapes")
    L = _lib.lib()
    streams, first, mask2, lap, mask3 = _streams(d, kind)
    assert codegen.ensure_mlp_kernels(desc) and L.ndq_mlp_supported(ctypes.byref(desc)) == 1
    dims = (d,) + ws + (n_out,)
    rng = np.random.default_rng(zlib.crc32(f"{shape}/{n}".encode()))
    parts = []
    for a, b in zip(dims[:-1], dims[1:]):
        k = 1.0 / np.sqrt(a)
        parts += [rng.uniform(-k, k, a * b), rng.uniform(-k, k, b)]
    flat = np.concatenate(parts).astype(np.float32)
    P = flat.size
    assert L.ndq_mlp_num_params(ctypes.byref(desc)) == P and L.ndq_mlp_num_streams(ctypes.byref(desc)) == len(streams)
    coords = rng.uniform(-1.0, 1.0, (d, n)).astype(np.float32)
    ld = (n + 63) // 64 * 64
    c = torch.zeros(d, ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
    p = torch.from_numpy(flat).cuda()
    jets = torch.full((len(streams), n_out, ld), float("nan"), device="cuda")
    assert L.ndq_mlp_jet_fwd(ctypes.byref(desc), c.data_ptr(), ld, n, p.data_ptr(), jets.data_ptr(), ld, _stream()) == 0
    torch.cuda.synchronize()
    got = jets[:, :, :n].cpu().numpy()
    f64, c64 = flat.astype(np.float64), list(coords.astype(np.float64))
    want = _oracle_streams(f64, dims, act, c64, streams)
    floor = (0.1 if n < 64 else 0.0) * np.sqrt(n * n_out) * max(np.sqrt(np.mean(want[m] ** 2)) for m in streams)
    errs = {str(m): float(np.linalg.norm(got[s].T - want[m]) / max(np.linalg.norm(want[m]), floor))
            for s, m in enumerate(streams)}
    # adjoint: random seeds on every stream; the Laplacian stream's seed goes to every diagonal pair
    gbar = rng.standard_normal((len(streams), n_out, n)).astype(np.float32)
    g = torch.zeros(len(streams), n_out, ld, device="cuda"); g[:, :, :n] = torch.from_numpy(gbar)
    nb = L.ndq_mlp_bwd_blocks(ctypes.byref(desc), n)
    part = torch.full((nb, P), float("nan"), device="cuda")
    out = torch.zeros(P, device="cuda")
    assert L.ndq_mlp_jet_bwd(ctypes.byref(desc), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, part.data_ptr(), _stream()) == 0
    assert L.ndq_reduce_partials(part.data_ptr(), nb, P, out.data_ptr(), 0, 1.0, _stream()) == 0
    torch.cuda.synchronize()
    grad = out.cpu().numpy()
    gb = {}
    for s, m in enumerate(streams):
        for mm in ([(a, a) for a in range(d)] if m == "lap" else [m]):
            gb[mm] = gb.get(mm, 0) + gbar[s].astype(np.float64).T
    want_grad = J.mlp_jets_vjp(f64, dims, act, c64, gb)[:P]
    errs["grad"] = rel_l2(grad, want_grad)
    w1, wL = ws[0], ws[-1]
    for name, lo, hi in (("dW1", 0, d * w1), ("db1", d * w1, d * w1 + w1), ("dWhidden", d * w1 + w1, P - n_out * wL - n_out),
                         ("dWout", P - n_out * wL - n_out, P - n_out), ("dbout", P - n_out, P)):
        errs[name] = float(np.linalg.norm(grad[lo:hi] - want_grad[lo:hi]) / max(np.linalg.norm(want_grad), 1e-300))
    # bit-reproducible: a second launch gives the same bits
    part2 = torch.full((nb, P), float("nan"), device="cuda")
    assert L.ndq_mlp_jet_bwd(ctypes.byref(desc), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, part2.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(part, part2)
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, f"wide_kernel_{'-'.join(map(str, shape))}_{n}.json"), "w") as fh:
        json.dump(errs, fh, indent=1)
    assert max(errs.values()) < TOL, errs


def _grad_in_torch_order(nets, flats):
    where = {}
    for fp in flats:
        for prm, off in zip(fp.params, fp._offsets):
            where[id(prm)] = fp.grad[off:off + prm.numel()]
    # (parameters outside the kernels' flat vectors -- a symbolic skip connection's weights -- carry their gradient in .grad)
    return torch.cat([(where[id(prm)] if id(prm) in where else prm.grad).reshape(-1).to("cuda")
                      for net in nets for prm in net.parameters()]).cpu().numpy()


GOLDEN_WIDE = ["w16", "w17", "w18", "w19", "w20", "w21", "w24", "w25", "w26", "w27", "w28"]      # w20: (100, 100) ELU; w21: Softplus + GELU networks (32 x 32);
#                                                 w24 / w25: Resnet 128 x 2 and 2 -> 512 -> 3 (skip connection above 64 units: symbolic, round 5)
GOLDEN_DEEP = ("w18", "w19", "w20", "w21", "w24", "w26", "w27", "w28")     # (w26 / w27: per-layer widths above 64 units, (128, 64) and (96, 200, 40) sigmoid)
#      # layer-by-layer kernels (w18 - w20) / two networks with different activations (w21):
#                                                 three-kernel pipeline, no single-launch closure


@pytest.mark.parametrize("mode", ["1k", "3k"])
@pytest.mark.parametrize("name", GOLDEN_WIDE)
def test_wide_closure_matches_reference_golden(golden_dir, name, mode):
    """funcs / residuals / loss / gradient of ONE closure against what the unmodified reference produced in fp64 on the same
    parameters and points -- single launch (wide_closure_kernel) and three-kernel pipeline (wide_jet_fwd / generated
    pointwise kernel / wide_jet_bwd)."""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    if name in GOLDEN_DEEP and mode == "1k":
        pytest.skip("no single-launch closure for this system (layer-by-layer kernels / networks of different shapes): pipeline mode only")
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(0)
    cfg = configs.make(name, None)
    for net in cfg["nets"]:
        net.to("cuda")
    fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                     compute_func_val=configs.func_val(cfg), single_kernel=(mode == "1k"))
    assert (fs.fusedk is not None) == (mode == "1k")
    R.set_flat(cfg["nets"], torch.from_numpy(gold["params0"]))
    b, n = fs.step([torch.from_numpy(c) for c in gold["coords"]], train=True, slot=0, want_funcs=True, want_resid=True)
    fs.attach_theta_grads()                    # (w24 / w25: the skip connection's weights are kernel arguments; their .grad)
    torch.cuda.synchronize()
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), gold["funcs_f64"]),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy()[:, :gold["residuals_f64"].shape[1]], gold["residuals_f64"]),
                loss=abs(fs.loss_buf[0].item() - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), gold["grad_f64"]))
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, f"wide_closure_{name}_{mode}.json"), "w") as fh:
        json.dump(errs, fh, indent=1)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", GOLDEN_WIDE)
def test_wide_solver_trajectory_matches_reference_golden(golden_dir, name):
    """Three epochs of Solver.run_train_epoch on the fused path against the reference solver's loss history and final
    parameters; no composite-path warning, describe() accepts the network."""
    import warnings
    from tests import configs
    from neurodiffeq_amd import networks
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    solver, cfg = configs.make_solver(name, None)
    assert networks.describe(cfg["nets"][0]) is not None
    solver.fused = "require"
    assert np.array_equal(R.get_flat(cfg["nets"]).cpu().numpy(), gold["params0"])
    torch.manual_seed(int(gold["seed"]) + 2)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        for _ in range(3):
            solver.run_train_epoch()
    assert solver.fused_active
    hist = np.array(solver.metrics_history["train_loss"])
    params = R.get_flat(cfg["nets"]).cpu().numpy()
    errs = dict(loss=float(np.max(np.abs(hist - gold["traj_loss"]) / np.abs(gold["traj_loss"]))),
                params=rel_l2(params, gold["traj_params"]))
    assert errs["loss"] < 2e-5 and errs["params"] < 1e-5, (errs, hist, gold["traj_loss"])


@pytest.mark.parametrize("name,size,mode", [("w17", 256, "1k"), ("w17", 256, "3k"), ("w18", 256, "3k"), ("w19", 256, "3k"),
                                            ("w18r", (251, 261), "3k"), ("w20", 256, "3k"), ("w26", 256, "3k")])
def test_wide_closure_matches_reference_golden_at_the_size_the_bench_times(golden_dir, name, size, mode):
    """VERDICT r4 weak #2 / next #2a: bench.py times the wide legs (cavity 512 x 1, Burgers 128 x 3, cavity 256 x 2) at 65 536
    points; the stripe / tile / reduction job tables of the layer-by-layer kernels depend on the batch size, so parity is
    pinned AT that size (and at a ragged 251 x 261 = 65 511 points) against numbers the unmodified reference produced in
    fp64 (tests/golden/make_golden.py make_full -> <name>_full.npz: loss, flat gradient, column sums of the function values
    and squared residuals).  The batch is regenerated from the seed (bit-exact generator contract) and checked against the
    head and the bit-pattern checksum of the reference's own draw."""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    gold = np.load(os.path.join(golden_dir, f"{name}_full.npz"))
    torch.manual_seed(0)
    cfg = configs.make(name, size)
    for net in cfg["nets"]:
        net.to("cuda")
    fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                     compute_func_val=configs.func_val(cfg), single_kernel=(mode == "1k"))
    assert (fs.fusedk is not None) == (mode == "1k")
    R.set_flat(cfg["nets"], torch.from_numpy(gold["params0"]))
    torch.manual_seed(int(gold["seed"]) + 1)
    coords = [c.detach() for c in cfg["gen"].get_examples()]
    assert coords[0].numel() == int(gold["n_points"])
    assert np.array_equal(np.stack([c[:8].numpy() for c in coords]), gold["coords_head"])
    bits = np.asarray([c.view(torch.int32).to(torch.int64).sum().item() for c in coords])
    assert np.array_equal(bits, gold["coords_bits_sum"]), (bits, gold["coords_bits_sum"])
    b, n = fs.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    n_eq = gold["resid_sq_sum"].shape[0]
    fsum = b["funcs"][:, :n].double().sum(dim=1).cpu().numpy()[:gold["funcs_sum"].shape[0]]
    r2sum = (b["resid"][:n_eq, :n].double() ** 2).sum(dim=1).cpu().numpy()
    loss = float(fs.loss_buf[0].item())
    errs = dict(loss=abs(loss - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), gold["grad_f64"]),
                funcs_sum=rel_l2(fsum, gold["funcs_sum"]), resid_sq_sum=rel_l2(r2sum, gold["resid_sq_sum"]))
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, f"wide_closure_full_{name}_{mode}.json"), "w") as fh:
        json.dump(errs, fh, indent=1)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name,base", [("w22", "w16"), ("w23", "w17")])
def test_wide_networks_in_the_references_default_precision(golden_dir, name, base):
    """VERDICT r4 missing #2 / next #3: ``import neurodiffeq`` means float64 (``__init__.py:22``), and the README's own network
    is ``FCNN(2, 1, hidden_units=(512,))`` (``README.md:125``) -- that pair used to leave the fused path.  Round 5 compiles the
    forward-stream / adjoint kernels of csrc/ndq_wide.h in double (three-kernel pipeline, epoch tail on the device).  Against
    what the unmodified reference produced under its float64 defaults (tests/golden/make_golden.py: make_default_precision --
    networks initialised in double, points drawn in double): one closure to 1e-9, three Adam epochs of the solver."""
    import warnings
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    from neurodiffeq_amd.utils import set_tensor_type
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    set_tensor_type(device="cpu", float_bits=64)          # (cpu default device: the host generator's numbers, bit for bit)
    try:
        torch.manual_seed(int(gold["seed"]))
        cfg = configs.make(base, None)
        assert R.get_flat(cfg["nets"]).dtype == torch.float64
        assert np.array_equal(R.get_flat(cfg["nets"]).numpy(), gold["params0"])          # same initialisation, in double
        for net in cfg["nets"]:
            net.to("cuda")
        fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                         compute_func_val=configs.func_val(cfg), dtype=torch.float64, single_kernel=False)
        assert fs.f64 and fs.fusedk is None
        b, n = fs.step([torch.from_numpy(c) for c in gold["coords"]], train=True, slot=0, want_funcs=True, want_resid=True)
        torch.cuda.synchronize()
        errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), gold["funcs_f64"]),
                    residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy()[:, :gold["residuals_f64"].shape[1]], gold["residuals_f64"]),
                    loss=abs(fs.loss_buf[0].item() - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                    grad=rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), gold["grad_f64"]))
        assert max(errs.values()) < 1e-9, errs
        # the solver, end to end: same seeds, same draws (float64 host generator), three epochs of the default Adam
        torch.manual_seed(int(gold["seed"]))
        solver, cfg = configs.make_solver(base, None)
        solver.fused = "require"
        torch.manual_seed(int(gold["seed"]) + 2)
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            for _ in range(3):
                solver.run_train_epoch()
        assert solver.fused_active and solver._fused_sys.f64
        hist = np.array(solver.metrics_history["train_loss"])
        params = R.get_flat(cfg["nets"]).cpu().numpy()
        tol = 1e-9 if name == "w22" else 1e-6          # (w23's loss grows 75-fold over the three epochs: rounding is amplified)
        errs = dict(loss=float(np.max(np.abs(hist - gold["traj_loss"]) / np.abs(gold["traj_loss"]))),
                    params=rel_l2(params, gold["traj_params"]))
        assert errs["loss"] < tol and errs["params"] < tol, (errs, hist, gold["traj_loss"])
    finally:
        set_tensor_type(device="cpu", float_bits=32)


def test_readme_laplace_512_at_the_headline_size_matches_oracle():
    """README.md:125's network on BASELINE C2's grid (256 x 256 = 65 536 points): single-launch closure against the fp64
    autograd oracle walked in chunks; and the two launch modes agree."""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(0)
    cfg = configs.make("w16", 256)
    flat0 = R.get_flat(cfg["nets"]).clone()
    torch.manual_seed(1)
    coords = [c.detach() for c in cfg["gen"].get_examples()]
    torch.manual_seed(0)
    ocfg = R.build_config("c2", 256, dtype=torch.float64)
    ocfg["nets"] = [R.make_fcnn(2, 1, (512,), "tanh", torch.float64)]
    R.set_flat(ocfg["nets"], flat0.double())
    want = R.closure_chunked(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], [c.double() for c in coords], chunk=16384)
    want_grad = R.get_flat_grad(ocfg["nets"]).numpy()
    for net in cfg["nets"]:
        net.to("cuda")
    out = {}
    for mode in ("1k", "3k"):
        fs = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cuda", single_kernel=(mode == "1k"))
        b, n = fs.step(coords, train=True, slot=0)
        torch.cuda.synchronize()
        out[mode] = (fs.loss_buf[0].item(), _grad_in_torch_order(cfg["nets"], fs.flat))
        errs = dict(loss=abs(out[mode][0] - want["loss"].item()) / abs(want["loss"].item()), grad=rel_l2(out[mode][1], want_grad))
        assert max(errs.values()) < TOL, (mode, errs)
    assert rel_l2(out["1k"][1], out["3k"][1]) < 2e-6


def test_wide_network_fit_is_bit_identical_to_epoch_by_epoch():
    """fit(n) (whole chunks of epochs per native call, training + validation workgroups in one launch) against single
    epochs, for the README network: histories, parameters and best network equal bit for bit."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import DirichletBVP2D
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver2D
    zero = lambda v: 0 * v
    runs = {}
    for how in ("fit", "single"):
        torch.manual_seed(0)
        solver = Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                          [DirichletBVP2D(0, lambda y: torch.sin(np.pi * y), 1, zero, 0, zero, 1, zero)],
                          xy_min=(0, 0), xy_max=(1, 1), nets=[FCNN(n_input_units=2, n_output_units=1, hidden_units=(512,))])
        solver.fused = "require"
        torch.manual_seed(1)
        if how == "fit":
            solver.fit(7)
        else:
            for _ in range(7):
                solver.run_train_epoch()
                solver.run_valid_epoch()
        assert solver.fused_active
        runs[how] = (np.array(solver.metrics_history["train_loss"]), np.array(solver.metrics_history["valid_loss"]),
                     R.get_flat(solver.nets).cpu().numpy(), R.get_flat(solver.best_nets).cpu().numpy())
    for a, b in zip(runs["fit"], runs["single"]):
        assert np.array_equal(a, b)
    assert runs["fit"][0][-1] < runs["fit"][0][0]


@pytest.mark.gpu
def test_deep_wide_network_at_c5_size_matches_its_shards_and_the_oracle():
    """the layer-by-layer kernels (csrc/ndq_deep.h) keep Z_l in an HBM workspace whose stripe /
    tile / reduction tables depend on the batch size, and nothing above 65 536 points pinned them.  The RE100 notebook's
    network (FCNN(n_hidden_units=256, n_hidden_layers=1) = 2 -> 256 -> 256 -> 3, lid-driven cavity on one network) at BASELINE C5's
    1 048 576 points: (1) size-independent property -- loss and gradient are sums over points, so the full batch must equal the
    sum of its sixteen 65 536-point shards evaluated one by one (each with the GLOBAL normalisation, accumulated in fp64 on the
    host); (2) the function values / residuals of the full launch, point by point, and (3) loss + gradient of 16 384 points of one shard
    against the fp64 autograd oracle on those points.  (The whole batch through the CPU oracle would cost ~10 minutes of host
    autograd; the property covers what the sample does not.)"""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    g = 1024
    torch.manual_seed(0)
    cfg = configs.make("w19", g)
    flat0 = R.get_flat(cfg["nets"]).clone()
    torch.manual_seed(1)
    coords = [c.detach() for c in cfg["gen"].get_examples()]
    n_all = coords[0].numel()
    assert n_all == 1 << 20
    for net in cfg["nets"]:
        net.to("cuda")
    fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), 2, "cuda", compute_func_val=configs.func_val(cfg),
                     single_kernel=False)
    dev = [c.cuda() for c in coords]
    b, n = fs.step(dev, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    assert n == n_all
    loss_full = float(fs.loss_buf[0].item())
    grad_full = _grad_in_torch_order(cfg["nets"], fs.flat).astype(np.float64)
    funcs_full = b["funcs"][:, :n].clone()
    resid_full = b["resid"][:, :n].clone()
    # (1) the sixteen shards, one launch sequence each, normalised by the global batch size
    shard = 1 << 16
    loss_sum, grad_sum = 0.0, np.zeros_like(grad_full)
    for k in range(n_all // shard):
        fs.step(dev, train=True, slot=0, n_global=n_all, lo=k * shard, hi=(k + 1) * shard)
        torch.cuda.synchronize()
        loss_sum += float(fs.loss_buf[0].item())
        grad_sum += _grad_in_torch_order(cfg["nets"], fs.flat).astype(np.float64)
    errs = dict(loss_vs_shards=abs(loss_full - loss_sum) / abs(loss_sum), grad_vs_shards=rel_l2(grad_full, grad_sum))
    # (2), (3) the fp64 autograd oracle on shard 11
    k = 11
    sl = slice(k * shard, k * shard + 16384)          # (a quarter of shard 11: ~25 s of host autograd instead of ~100 s)
    # the oracle's restatement of the single-network cavity (oracle/autograd_ref.py primitives: C5's conditions and equations,
    # experiments/lid-driven-cavity-RE400.ipynb cell 3, on the three output columns of ONE network; Re = 100 as in tests/configs.py w19)
    d, zero = R.ref_diff, (lambda s: 0)
    onet = R.make_fcnn(2, 3, (256, 256), "tanh", torch.float64)
    R.set_flat([onet], flat0.double())
    cu, cv = R.dirichlet_bvp2d(0, zero, 1, zero, 0, zero, 1, R.lid_profile), R.dirichlet_bvp2d(0, zero, 1, zero, 0, zero, 1, zero)

    class _Col(torch.nn.Module):          # column k of the shared network as a "network" of its own for the oracle's enforcers
        def __init__(self, k):
            super().__init__()
            self.k = k

        def forward(self, xy):
            return onet(xy)[:, self.k:self.k + 1]
    enforcers = [lambda net, x, y: torch.cat([cu(_Col(0), x, y), cv(_Col(1), x, y), _Col(2)(torch.cat([x, y], 1))], 1)]

    def opde(uvp, x, y):
        u, v, p = uvp[:, 0:1], uvp[:, 1:2], uvp[:, 2:3]
        mx = u * d(u, x) + v * d(u, y) + d(p, x) - 1 / 100.0 * (d(u, x, 2) + d(u, y, 2))
        my = u * d(v, x) + v * d(v, y) + d(p, y) - 1 / 100.0 * (d(v, x, 2) + d(v, y, 2))
        return [mx, my, d(u, x) + d(v, y)]
    want = R.closure_chunked([onet], enforcers, opde, [c[sl].double() for c in coords], chunk=16384, keep=True)
    want_grad = R.get_flat_grad([onet]).numpy()
    fs.step(dev, train=True, slot=0, lo=sl.start, hi=sl.stop)
    torch.cuda.synchronize()
    errs["loss_shard"] = abs(float(fs.loss_buf[0].item()) - want["loss"].item()) / abs(want["loss"].item())
    errs["grad_shard"] = rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), want_grad)
    errs["funcs_full_run"] = rel_l2(funcs_full[:, sl].T.cpu().numpy(), want["funcs"].numpy())
    errs["resid_full_run"] = rel_l2(resid_full[:, sl].T.cpu().numpy(), want["residuals"].numpy())
    os.makedirs(DIAG, exist_ok=True)
    with open(os.path.join(DIAG, "deep_wide_1m.json"), "w") as fh:
        json.dump(errs, fh, indent=1)
    assert max(errs.values()) < TOL, errs
