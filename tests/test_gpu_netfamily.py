"""Networks beyond the plain FCNN template on the HIP path (VERDICT r1 row f4 / missing #7): Resnet with several output
units (``ndq_mlp_desc.skip`` on the MFMA output layer) and trainable Swish / APTx parameters (``ndq_mlp_desc.actp``:
the scales live in the staged weights, the reverse pass carries the gradient back through that map and accumulates
APTx's alpha directly -- csrc/ndq_mlp.h).  Kernel level through the C-ABI against the numpy jet oracle, closure level
against the autograd oracle, and a few optimiser steps against torch on the composite path."""
import ctypes
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

FULL2 = {1: [(), (0,), (0, 0)], 2: [(), (0,), (1,), (0, 0), (0, 1), (1, 1)]}
# name: (dims, activation, skip, actp)
CASES = {
    "swish_tr": ((2, 32, 32, 1), "swish", 0, 1),
    "aptx_tr": ((2, 32, 32, 1), "aptx", 0, 1),
    "aptx_tr_wide": ((2, 64, 64, 64, 1), "aptx", 0, 1),
    "swish_tr_3out": ((1, 32, 32, 3), "swish", 0, 1),
    "aptx_tr_skip": ((2, 32, 32, 1), "aptx", 1, 1),
    "res_3out": ((2, 32, 32, 3), "tanh", 1, 0),
    "res_25out": ((1, 32, 32, 25), "tanh", 1, 0),
    "res_3out_wide": ((2, 64, 64, 64, 3), "tanh", 1, 0),
    "res_aptx_tr_3out": ((2, 32, 32, 3), "aptx", 1, 1),
    # hidden widths that are no multiple of 16 (Cfg::HR: padded in registers / LDS, real in the flat vectors)
    "w20": ((2, 20, 20, 1), "tanh", 0, 0),
    "w50x3": ((2, 50, 50, 50, 1), "tanh", 0, 0),
    "w40_sigmoid": ((1, 40, 40, 1), "sigmoid", 0, 0),
    "w24_3out_skip": ((2, 24, 24, 3), "tanh", 1, 0),
    "w50_25out": ((1, 50, 50, 25), "tanh", 0, 0),
    "w10x1_sin": ((2, 10, 1), "sin", 0, 0),
    "w50_swish_tr": ((2, 50, 50, 1), "swish", 0, 1),
    # fixed non-default activation scalars (actp = 2): behind the trainable entries of the parameter buffer, no gradients
    # hidden layers of different widths (ndq_mlp_desc.widths): all laid out for the widest one
    "funnel_64_32_16": ((2, 64, 32, 16, 1), "tanh", 0, 0),
    "funnel_50_30_3out_skip": ((2, 50, 30, 3), "tanh", 1, 0),
    "expand_16_48_sin": ((1, 16, 48, 1), "sin", 0, 0),
    "funnel_40_20_swish_tr": ((2, 40, 20, 1), "swish", 0, 1),
    # MonomialNN in front (ndq_mlp_desc.mono; 5th entry: the degrees): the first layer sees x_a^deg and its derivatives
    "mono123_tanh": ((2, 32, 32, 1), "tanh", 0, 0, (1, 2, 3)),
    "mono135_2out": ((1, 16, 16, 2), "sigmoid", 0, 0, (1, 3, 5)),
    "mono1234_sin_w20": ((1, 20, 20, 1), "sin", 0, 0, (1, 2, 3, 4)),
    "mono12_swish_tr": ((2, 32, 32, 1), "swish", 0, 1, (1, 2)),
    "swish_fixed": ((2, 32, 32, 1), "swish", 0, 2),
    "aptx_fixed_3out": ((2, 32, 32, 3), "aptx", 0, 2),
}
ACT_ID = {"tanh": 0, "sin": 1, "sigmoid": 2, "swish": 3, "aptx": 4}


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _desc(name):
    from neurodiffeq_amd import _lib
    dims, act, skip, actp = CASES[name][:4]
    mono = sum(1 << (k - 1) for k in CASES[name][4]) if len(CASES[name]) > 4 else 0
    d, ws = dims[0], dims[1:-1]
    widths = 0 if len(set(ws)) == 1 else sum(w << (8 * i) for i, w in enumerate(ws))
    return _lib.MlpDesc(d, 1, (1 << (d * (d + 1) // 2)) - 1, max(ws), len(ws), ACT_ID[act], dims[-1], 0, skip, 0, actp, widths, mono)


def _flat(name, rng):
    dims, act, skip, actp = CASES[name][:4]
    if len(CASES[name]) > 4:
        dims = (dims[0] * len(CASES[name][4]),) + tuple(dims[1:])
    parts = []
    for a, b in zip(dims[:-1], dims[1:]):
        k = 1.0 / np.sqrt(a)
        parts += [rng.uniform(-k, k, a * b), rng.uniform(-k, k, b)]
    if skip:
        parts.append(rng.uniform(-0.7, 0.7, dims[-1] * dims[0]))
    if actp:
        for _ in range(len(dims) - 2):
            parts.append(rng.uniform(0.6, 1.5, 1) if act == "swish"
                         else np.array([rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.5), rng.uniform(0.3, 0.8)]))
    return np.concatenate(parts).astype(np.float32)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n", [17, 1000])
@pytest.mark.parametrize("name", list(CASES))
def test_stream_kernels_match_jet_oracle(name, n):
    from neurodiffeq_amd import _lib, codegen
    L = _lib.lib()
    dims, act, skip, actp = CASES[name][:4]
    mono = list(CASES[name][4]) if len(CASES[name]) > 4 else None
    streams = FULL2[dims[0]]
    d = _desc(name)
    assert codegen.ensure_mlp_kernels(d) and L.ndq_mlp_supported(ctypes.byref(d)) == 1
    rng = np.random.default_rng(zlib.crc32(f"{name}/{n}".encode()))
    flat = _flat(name, rng)
    n_frozen = (len(dims) - 2) * (1 if act == "swish" else 3) if actp == 2 else 0
    P = flat.size - n_frozen
    assert L.ndq_mlp_num_params(ctypes.byref(d)) == P and L.ndq_mlp_num_streams(ctypes.byref(d)) == len(streams)
    coords = rng.uniform(-1.0, 1.0, (dims[0], n)).astype(np.float32)
    ld = (n + 63) // 64 * 64
    c = torch.zeros(dims[0], ld, device="cuda"); c[:, :n] = torch.from_numpy(coords)
    p = torch.from_numpy(flat).cuda()
    jets = torch.full((len(streams), dims[-1], ld), float("nan"), device="cuda")
    assert L.ndq_mlp_jet_fwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), jets.data_ptr(), ld, _stream()) == 0
    torch.cuda.synchronize()
    got = jets[:, :, :n].cpu().numpy()
    f64, c64 = flat.astype(np.float64), list(coords.astype(np.float64))
    if mono is not None and actp:       # the oracle states activation parameters and monomial features separately: theta by hand
        n_lin0 = J._n_fcnn_params(J._mono_dims(dims, mono))
        k_act = 1 if act == "swish" else 3
        thetas = [tuple(f64[n_lin0 + k_act * l: n_lin0 + k_act * (l + 1)]) for l in range(len(dims) - 2)]
        want = J._forward(f64[:n_lin0], dims, act, c64, J.close_streams(streams), thetas, mono)[0]
    else:
        want = J.mlp_jets(f64, dims, act, c64, streams, skip=bool(skip), actp=bool(actp), mono=mono)
    floor = (0.1 if n < 64 else 0.0) * np.sqrt(n * dims[-1]) * max(np.sqrt(np.mean(want[m] ** 2)) for m in streams)
    errs = {str(m): float(np.linalg.norm(got[s].T - want[m]) / max(np.linalg.norm(want[m]), floor))
            for s, m in enumerate(streams)}
    gbar = rng.standard_normal((len(streams), dims[-1], n)).astype(np.float32)
    g = torch.zeros(len(streams), dims[-1], ld, device="cuda"); g[:, :, :n] = torch.from_numpy(gbar)
    nb = L.ndq_mlp_bwd_blocks(ctypes.byref(d), n)
    part = torch.full((nb, P), float("nan"), device="cuda")
    out = torch.zeros(P, device="cuda")
    assert L.ndq_mlp_jet_bwd(ctypes.byref(d), c.data_ptr(), ld, n, p.data_ptr(), g.data_ptr(), ld, part.data_ptr(), _stream()) == 0
    assert L.ndq_reduce_partials(part.data_ptr(), nb, P, out.data_ptr(), 0, 1.0, _stream()) == 0
    torch.cuda.synchronize()
    grad = out.cpu().numpy()
    gb = {m: gbar[s].astype(np.float64).T for s, m in enumerate(streams)}
    if mono is not None and actp:
        want_grad = J.mlp_jets_vjp(f64[:n_lin0], dims, act, c64, gb, thetas=thetas, mono=mono)      # linear layers only
        grad = grad[:n_lin0]
        P = n_lin0
        actp = 0
    else:
        want_grad = J.mlp_jets_vjp(f64, dims, act, c64, gb, skip=bool(skip), actp=bool(actp), mono=mono)[:P]
    n_lin = J._n_fcnn_params(J._mono_dims(dims, mono))
    n_skip = dims[-1] * dims[0] if skip else 0
    errs["grad_linear"] = rel_l2(grad[:n_lin], want_grad[:n_lin])
    if skip:
        errs["grad_skip"] = rel_l2(grad[n_lin:n_lin + n_skip], want_grad[n_lin:n_lin + n_skip])
    if actp == 1:
        errs["grad_act"] = rel_l2(grad[n_lin + n_skip:], want_grad[n_lin + n_skip:])
    os.makedirs(DIAG, exist_ok=True)
    import json
    with open(os.path.join(DIAG, f"netfamily_kernel_{name}_{n}.json"), "w") as fh:
        json.dump(dict(errs, got_act=grad[n_lin + n_skip:].tolist(), want_act=want_grad[n_lin + n_skip:].tolist()), fh, indent=1)
    assert max(errs.values()) < TOL, errs


def _grad_in_torch_order(nets, flats):
    where = {}
    for fp in flats:
        for prm, off in zip(fp.params, fp._offsets):
            where[id(prm)] = fp.grad[off:off + prm.numel()]
    return torch.cat([where[id(prm)].reshape(-1) for net in nets for prm in net.parameters()]).cpu().numpy()


@pytest.mark.parametrize("mode", ["1k", "3k"])
@pytest.mark.parametrize("name", ["swish_tr_laplace", "aptx_tr_laplace", "aptx_tr_wide", "swish_tr_system", "aptx_tr_resnet",
                                  "shape_50x2", "shape_20x3", "shape_40x2_sigmoid", "shape_10x1", "swish_fixed_laplace",
                                  "aptx_fixed_laplace", "ensemble_lv", "shape_64_32", "shape_24_40_12_sigmoid",
                                  "mono_laplace", "mono_ode", "mono_poisson"])
def test_closure_of_networks_outside_the_template_matches_autograd_oracle(name, mode):
    """funcs / residuals / loss / gradient of one closure, the gradient compared parameter by parameter in torch order
    (activation scalars interleaved with the linear layers there, behind them in the kernels' flat vector)."""
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
    assert (fs.fusedk is not None) == (mode == "1k")
    b, n = fs.step([c.float() for c in coords], train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    grad = _grad_in_torch_order(nets, fs.flat)
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), want["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(grad, want_grad))
    off = 0
    for net, onet in zip(nets, onets):
        for (pname, prm), oprm in zip(net.named_parameters(), onet.parameters()):
            if prm.dim() == 0:
                errs[f"d_{pname}"] = abs(grad[off] - oprm.grad.item()) / max(abs(oprm.grad.item()), 1e-3 * np.linalg.norm(want_grad))
            off += prm.numel()
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("mode", ["1k", "3k"])
def test_multi_output_resnet_closure_matches_autograd_oracle(mode):
    """C4's spherical-harmonics coefficient problem with a Resnet(1 -> 25) instead of the FCNN: the skip matrix S (25 x 1)
    takes part in the value and the d/dr streams of all 25 outputs, and gets its gradient."""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    from neurodiffeq_amd.networks import Resnet
    size = 5000
    torch.manual_seed(0)
    cfg = configs.make("c4", size)
    cfg["nets"] = [Resnet(1, 25, hidden_units=(32, 32))]
    torch.manual_seed(0)
    ocfg = R.build_config("c4", size, dtype=torch.float64)
    ocfg["nets"] = [R.ResnetRef(1, 25, (32, 32), "tanh", dtype=torch.float64)]
    R.set_flat(ocfg["nets"], R.get_flat(cfg["nets"]).double())
    torch.manual_seed(3)
    coords = [c.detach() for c in cfg["gen"].get_examples()]
    out = R.closure(ocfg["nets"], ocfg["enforcers"], ocfg["pde"], [c.double() for c in coords])
    want_grad = R.get_flat_grad(ocfg["nets"]).numpy()
    for net in cfg["nets"]:
        net.to("cuda")
    for c in cfg["conds"]:
        c.R_0, c.R_1 = c.R_0.cuda(), c.R_1.cuda()
    fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                     compute_func_val=configs.func_val(cfg), single_kernel=(mode == "1k"))
    assert (fs.fusedk is not None) == (mode == "1k")
    b, n = fs.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    grad = _grad_in_torch_order(cfg["nets"], fs.flat)
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), out["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), out["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - out["loss"].item()) / abs(out["loss"].item()),
                grad=rel_l2(grad, want_grad), grad_skip=rel_l2(grad[-25:], want_grad[-25:]))
    assert max(errs.values()) < TOL, errs


def test_solver_trains_activation_parameters_like_torch():
    """Solver2D.fit on the fused path (device-side Adam over the flat vector, activation scalars included) against the
    same solver on the composite path (torch autograd + torch Adam): loss history, and every parameter -- the betas have
    moved and agree."""
    from functools import partial
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import DirichletBVP2D
    from neurodiffeq_amd.generators import Generator2D
    from neurodiffeq_amd.networks import FCNN, Swish
    from neurodiffeq_amd.solvers import Solver2D
    PI = np.pi
    zero = lambda s: 0 * s

    def run(fused):
        torch.manual_seed(0)
        net = FCNN(2, 1, hidden_units=(32, 32), actv=partial(Swish, beta=1.25, trainable=True))
        solver = Solver2D(pde_system=lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                          conditions=[DirichletBVP2D(0, lambda y: torch.sin(PI * y), 1, zero, 0, zero, 1, zero)],
                          xy_min=(0, 0), xy_max=(1, 1), nets=[net],
                          train_generator=Generator2D((32, 32), (0, 0), (1, 1), "equally-spaced-noisy"),
                          valid_generator=Generator2D((8, 8), (0, 0), (1, 1), "equally-spaced"), n_batches_valid=0)
        solver.fused = fused
        torch.manual_seed(1)
        solver.fit(max_epochs=25)
        assert solver.fused_active == (fused == "require")
        if fused == "require":      # best-network snapshot: a flat device copy in the kernels' parameter order, unpacked per module
            best = solver.best_nets[0]
            assert all(0.8 < best.NN[i].beta.item() < 1.8 for i in (1, 3)), [best.NN[i].beta.item() for i in (1, 3)]
            assert all(torch.isfinite(v).all() for v in best.parameters())
        return np.array(solver.metrics_history["train_loss"]), {k: v.detach().cpu().double().numpy().copy() for k, v in net.named_parameters()}

    hist_f, par_f = run("require")
    hist_c, par_c = run("off")
    assert np.allclose(hist_f, hist_c, rtol=2e-4), (hist_f, hist_c)
    betas = [k for k in par_f if k.endswith("beta")]
    assert len(betas) == 2 and all(abs(float(par_f[k]) - 1.25) > 1e-3 for k in betas)
    for k in par_f:
        assert np.linalg.norm(par_f[k] - par_c[k]) <= 2e-4 * max(np.linalg.norm(par_c[k]), 1e-2), k


def test_ensemble_condition_as_one_solver_function_trains_on_the_fused_path():
    """One two-output network under EnsembleCondition (conditions.py:157-202) handed to Solver1D as a SINGLE function
    whose columns the ODE system picks apart: fused path vs the composite path (torch autograd), and the solution object
    returns the (N, 2) function through the forward kernels."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import EnsembleCondition, IVP
    from neurodiffeq_amd.generators import Generator1D
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver1D

    def lv(uv, t):
        u, v = uv[:, 0:1], uv[:, 1:2]
        return [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)]

    def run(fused):
        torch.manual_seed(0)
        solver = Solver1D(lv, [EnsembleCondition(IVP(0.0, 1.5), IVP(0.0, 1.0))], t_min=0.1, t_max=4.0,
                          nets=[FCNN(1, 2, hidden_units=(32, 32))], train_generator=Generator1D(200, 0.1, 4.0, "equally-spaced-noisy"),
                          valid_generator=Generator1D(16, 0.1, 4.0), n_batches_valid=0)
        solver.fused = fused
        torch.manual_seed(1)
        solver.fit(max_epochs=20)
        assert solver.fused_active == (fused == "require")
        ts = torch.linspace(0.1, 4.0, 50, device="cuda").reshape(-1, 1)
        uv = solver.get_solution(best=False)(ts, to_numpy=False, no_reshape=True)
        return np.array(solver.metrics_history["train_loss"]), uv.detach().cpu().numpy()

    hist_f, uv_f = run("require")
    hist_c, uv_c = run("off")
    assert uv_f.shape == (50, 2) and uv_c.shape == (50, 2)
    assert np.allclose(hist_f, hist_c, rtol=2e-4), (hist_f, hist_c)
    assert np.linalg.norm(uv_f - uv_c) <= 2e-4 * np.linalg.norm(uv_c)


@pytest.mark.parametrize("name", ["c1", "c2", "c3x", "pendulum", "helmholtz_xy", "stokes_like", "sigmoid_mixed", "kdv",
                                  "resnet_laplace", "swish_tr_laplace", "aptx_tr_resnet", "shape_20x3", "mono_ode", "ensemble_lv",
                                  "shape_48x2", "shape_64x2"])
@pytest.mark.parametrize("mode", ["3k", "1k"])
def test_fp64_pipeline_matches_autograd_oracle(name, mode):
    """engine.FusedSystem(dtype=float64): forward streams (libndq64.so, f64 MFMA) -> the generated pointwise kernel
    compiled in double -> adjoint kernel -> fp64 sums, against the fp64 autograd oracle at 1e-9 (C1, C2, a 32-wide C3 and
    zoo systems: first order only, full Hessian, three networks, sigmoid, third-order streams, Laplacian-merged stream, skip
    connection, trainable activation parameters, a width that is no multiple of 16, monomial features, a two-column function)."""
    from tests import configs, zoo
    from neurodiffeq_amd.engine import FusedSystem
    torch.manual_seed(11)
    if name in ("c1", "c2", "c3x"):
        cfg = configs.make("c3" if name == "c3x" else name, 24 if name != "c1" else 500)
        if name == "c3x":
            from neurodiffeq_amd.networks import FCNN
            cfg["nets"] = [FCNN(2, 1, hidden_units=(32, 32, 32))]
        nets, conds, pde, n_coords = cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg)
        torch.manual_seed(11)
        ocfg = R.build_config("c3" if name == "c3x" else name, 24 if name != "c1" else 500, dtype=torch.float64)
        if name == "c3x":
            ocfg["nets"] = [R.make_fcnn(2, 1, (32, 32, 32), "tanh", torch.float64)]
        onets, enforcers, opde = ocfg["nets"], ocfg["enforcers"], ocfg["pde"]
        torch.manual_seed(3)
        ex = cfg["gen"].get_examples()
        coords = [c.detach().double() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
    else:
        system = zoo.build(name)
        nets, conds, pde = system.product()
        n_coords = system.n_coords
        coords = system.sample(3001, seed=5)
        onets, enforcers, opde = system.oracle(R.get_flat(nets))
    for net in nets:
        net.double()
    # fp64 initial values of their own (not fp32 roundings): the point is double precision end to end
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for net in nets:
            for prm in net.parameters():
                prm.add_(1e-9 * torch.randn(prm.shape, generator=gen, dtype=torch.float64))
    R.set_flat(onets, R.get_flat(nets).double())
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.to("cuda")
    # "1k": the single-launch closure kernel compiled in double (one network on the plain closure kernel); "3k": pipeline
    fs = FusedSystem(nets, conds, pde, n_coords, "cuda", dtype=torch.float64, single_kernel=(mode == "1k"))
    assert fs.f64 and (mode == "1k" or fs.fusedk is None)
    if mode == "1k" and fs.fusedk is None:
        pytest.skip("no fp64 closure kernel for this system (several networks / grouped kernel / LDS)")
    b, n = fs.step(coords, train=True, slot=0, want_funcs=True, want_resid=True)
    assert mode == "3k" or (fs.fusedk is not None and fs.fused_check["reproducible"] and fs.fused_check["grad_rel_l2"] < 1e-12)
    torch.cuda.synchronize()
    assert b["funcs"].dtype == torch.float64 and fs.flat[0].grad.dtype == torch.float64
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), want["funcs"].numpy()),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy(), want["residuals"].numpy()),
                loss=abs(fs.loss_buf[0].item() - want["loss"].item()) / abs(want["loss"].item()),
                grad=rel_l2(_grad_in_torch_order(nets, fs.flat), want_grad))
    assert max(errs.values()) < 1e-9, errs


def test_fused_solver_under_a_cuda_default_device():
    """A user script that calls ``set_tensor_type(device='cuda', float_bits=32)`` (utils.py:10-41: torch's default device
    becomes the GPU, generators sample there): the fused path -- native epochs, pinned staging, device-side Adam -- runs
    as under the CPU default and trains to the same losses on the same batches."""
    from tests import configs
    from neurodiffeq_amd.utils import set_tensor_type

    def run(cuda_default):
        try:
            if cuda_default:
                set_tensor_type(device="cuda", float_bits=32)
            torch.manual_seed(0)
            solver, cfg = configs.make_solver("c2", 16)
            solver.fused = "require"
            # the same initial parameters and points whatever device torch's RNG lives on
            init = np.random.default_rng(3).uniform(-0.3, 0.3, sum(p.numel() for p in cfg["nets"][0].parameters()))
            R.set_flat(cfg["nets"], torch.from_numpy(init.astype(np.float32)).to(next(cfg["nets"][0].parameters()).device))
            batch = [torch.linspace(0.05, 0.95, 256).reshape(-1, 1), torch.linspace(0.95, 0.05, 256).reshape(-1, 1) ** 2]
            solver.generator["train"].get_examples = lambda: batch          # the same points whatever the RNG device
            for _ in range(4):
                solver.run_train_epoch()
            assert solver.fused_active
            return np.array(solver.metrics_history["train_loss"])
        finally:
            set_tensor_type(device="cpu", float_bits=32)
    got, want = run(True), run(False)
    assert np.allclose(got, want, rtol=1e-6), (got, want)


# ------------------------------------------------------------------------------------------------ reference goldens
# w11 trainable Swish, w12 Resnet with hidden widths (50, 30), w13 MonomialNN front end, w14 EnsembleCondition on one
# two-output network, w15 trainable APTx: closures and 3-epoch trajectories produced by the UNMODIFIED reference
# (tests/golden/make_golden.py) -- the same pinning the BASELINE configs and w1 - w10 have.
# round 6: w29 fourth-order ODE (beam), w30 biharmonic equation -- diff(u, x, order=4) and the mixed quadruple xxyy
GOLDEN_FAMILY = ["w11", "w12", "w13", "w14", "w15", "w29", "w30"]


@pytest.mark.parametrize("mode", ["1k", "3k"])
@pytest.mark.parametrize("name", GOLDEN_FAMILY)
def test_network_family_closure_matches_reference_golden(golden_dir, name, mode):
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
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
    torch.cuda.synchronize()
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), gold["funcs_f64"]),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy()[:, :gold["residuals_f64"].shape[1]], gold["residuals_f64"]),
                loss=abs(fs.loss_buf[0].item() - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), gold["grad_f64"]))
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", GOLDEN_FAMILY)
def test_network_family_solver_trajectory_matches_reference_golden(golden_dir, name):
    """Three epochs of Solver.run_train_epoch (CPU RNG sampling, fused step, device-side Adam over the whole flat vector --
    activation scalars, skip weights included) against the reference solver's loss history and final parameters."""
    from tests import configs
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    solver, cfg = configs.make_solver(name, None)
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
    assert errs["loss"] < 2e-5 and errs["params"] < 1e-5, (errs, hist, gold["traj_loss"])


@pytest.mark.parametrize("name", ["c1", "c2", "c4", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "w8", "w9", "w10", "w11", "w12", "w13",
                                  "w14", "w15", "w16", "w17"])      # (w16 / w17: one hidden layer of 512 units, csrc/ndq_wide.h in double)
@pytest.mark.parametrize("mode", ["3k", "1k"])
def test_fp64_pipeline_matches_reference_golden(golden_dir, name, mode):
    """The fp64 pipeline against numbers the unmodified reference produced IN ITS DEFAULT PRECISION: every golden file
    holds the closure evaluated in fp64 on fp32-representable inputs (``funcs_f64`` ... ``grad_f64``).  1e-9 -- three
    orders below what fp32 arithmetic anywhere in the pipeline would leave.  (C3 / C5: 64 x 3 layers do not fit LDS in double.)"""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(0)
    cfg = configs.make(name, {"c1": 64, "c2": 16, "c4": 96}.get(name))
    for net in cfg["nets"]:
        net.double().to("cuda")
    R.set_flat(cfg["nets"], torch.from_numpy(gold["params0"]).double())
    fs = FusedSystem(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg), "cuda",
                     compute_func_val=configs.func_val(cfg), dtype=torch.float64, single_kernel=(mode == "1k"))
    if mode == "1k" and fs.fusedk is None:
        pytest.skip("no fp64 closure kernel for this system (several networks / grouped kernel / LDS)")
    assert (fs.fusedk is not None) == (mode == "1k")
    b, n = fs.step([torch.from_numpy(c).double() for c in gold["coords"]], train=True, slot=0, want_funcs=True, want_resid=True)
    torch.cuda.synchronize()
    errs = dict(funcs=rel_l2(b["funcs"][:, :n].T.cpu().numpy(), gold["funcs_f64"]),
                residuals=rel_l2(b["resid"][:, :n].T.cpu().numpy()[:, :gold["residuals_f64"].shape[1]], gold["residuals_f64"]),
                loss=abs(fs.loss_buf[0].item() - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])),
                grad=rel_l2(_grad_in_torch_order(cfg["nets"], fs.flat), gold["grad_f64"]))
    assert max(errs.values()) < 1e-9, errs
