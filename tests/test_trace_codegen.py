"""Tracer + symbolic diff + generated pointwise code (compiled for the host with gcc) + the jet oracle's VJP,
chained into a full closure and compared with the reference's golden vectors: everything on the fused path except
the HIP MLP kernels themselves (those are modelled lane-by-lane in test_wave_model.py)."""
import os

import numpy as np
import pytest
import torch

from neurodiffeq_amd import codegen
from neurodiffeq_amd.networks import describe
from neurodiffeq_amd.symbolic import Graph, Sym, trace_scope
from oracle import autograd_ref as R
from oracle import jet_ref as J
from tests import configs, zoo
from tests.pw_cpu import run_cpu

SIZES = {"c1": 64, "c2": 16, "c3": 12, "c5": 8, "c4": 96, "w1": None, "w2": None, "w3": None, "w4": None, "w5": None, "w6": None, "w7": None, "w8": None, "w9": None, "w10": None, "w11": None, "w12": None, "w13": None, "w14": None, "w15": None, "w16": None, "w17": None, "w18": None, "w19": None, "w20": None, "w21": None, "w24": None, "w25": None, "w26": None, "w27": None, "w28": None, "w29": None, "w30": None}


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def trace(nets, conds, pde, n_coords, lap=True, cfv=None, loss="l2", metrics=(), f64=False):
    from neurodiffeq_amd.symbolic import SymMat
    g = Graph(n_coords)
    g.f64 = bool(f64)                   # (as engine.trace_system does: the precision of the build the trace is for)
    g.register_nets(nets, [describe(n)["n_out"] for n in nets], skips=[describe(n).get("skip_sym") for n in nets])
    cfv = cfv or (lambda net, cond, *coords: cond.enforce(net, *coords))
    term, mterms = None, []
    with trace_scope(g):
        coords = [Sym(g, g.coord(i), leaf=True) for i in range(n_coords)]
        funcs = [cfv(n, c, *coords) for n, c in zip(nets, conds)]
        res = pde(*funcs, *coords)
        if callable(loss):              # custom loss_fn(residual, funcs, coords) -> batch mean (engine.trace_system)
            term, loss = loss(SymMat(res), list(funcs), list(coords)).term.i, "custom"
        mterms = [fn(*funcs, *coords).term.i for fn in metrics]
    for k, n in enumerate(nets):
        g.net_deps.setdefault(k, tuple(range(describe(n)["d"])))
        g.net_nout.setdefault(k, describe(n)["n_out"])
    cols = [c for f in funcs for c in (f.cols if isinstance(f, SymMat) else [f])]     # a multi-column function: one row per column
    return codegen.PointwiseProgram(g, [r.i for r in res], [c.i for c in cols] + mterms, len(nets),
                                    allow_lap=(lambda k, coords: True) if lap else None, loss=loss, loss_term=term)


def host_closure(nets, conds, pde, coords, params, lap=True, cfv=None, loss="l2", metrics=(), f64=False):
    """One training closure on the host: traced + generated pointwise code (gcc) around the jet oracle's network
    streams and VJP.  coords [d][n] fp32, params flat fp64.  Returns (program, funcs [n][nf], resid [n][neq], loss, grad)."""
    wdt = np.float64 if f64 else np.float32       # working precision of the generated pointwise code
    coords = np.ascontiguousarray(coords, wdt)
    n_coords, n = coords.shape
    prog = trace(nets, conds, pde, n_coords, lap, cfv, loss, metrics, f64=f64)
    if getattr(prog.g, "_nbatch_t", None) is not None:        # the batch size as a kernel argument (symbolic.Graph.nbatch)
        prog.g._nbatch_t.fill_(float(n))
    dims_act, flats, perms, off = [], [], [], 0
    for net in nets:
        info = describe(net)
        dims = (info["d"],) + tuple(l.out_features for l in info["linears"][:-1]) + (info["n_out"],)
        mono = [k + 1 for k in range(8) if (info["mono"] >> k) & 1] or None      # MonomialNN degrees in front of the network
        dims_act.append((dims, ("tanh", "sin", "sigmoid", "swish", "aptx", "elu", "softplus", "gelu")[info["act"]], bool(info["skip"]), bool(info["actp"]), mono))
        # ``params`` is in torch parameter order; the kernels' (and the jet oracle's) flat vector lists the linear layers,
        # then the skip weights, then the activation parameters (networks.describe): perm maps one onto the other
        start, at = {}, 0
        for prm in net.parameters():
            start[id(prm)] = at
            at += prm.numel()
        perm = np.concatenate([np.arange(start[id(prm)], start[id(prm)] + prm.numel()) for prm in info["params"]])
        # (a symbolic skip connection -- networks wider than 64 units -- keeps its weights outside the kernels' flat vector:
        # they are trainable scalars of the traced program, handled with the other theta entries below)
        assert perm.size == at or info.get("skip_sym") is not None
        perms.append((perm, at))
        # fixed non-default activation scalars (actp = 2) follow the trainable entries; the oracle takes them like trainable ones
        flats.append(np.concatenate([np.asarray(params[off:off + at], np.float64)[perm], np.asarray(info["frozen"], np.float64)]))
        off += at
    # a ("L", a, b, ..) symbol is the Laplacian stream = sum of the pure second derivatives (a,a), (b,b), ..
    parts = lambda mi: [(c, c) for c in mi[1:]] if (mi and mi[0] == "L") else [mi]
    # evaluation sites: (network, coordinate tuple) pairs; a virtual coordinate is a constant column
    site_net = prog.site_net
    column = lambda c: coords[c] if c < n_coords else np.full(n, prog.g.vcoords[c], wdt)
    needed = {k: set() for k in range(prog.n_sites)}
    for i in prog.symbols:
        _, k, o, mi = prog.g.nodes[i]
        needed[k].add(mi)
    jets = {}
    for k in range(prog.n_sites):
        dims, act, skip, actp, mono = dims_act[site_net[k]]
        deps = prog.streams[k].deps
        local = lambda mi: tuple(sorted(deps.index(c) for c in mi))
        want = sorted({local(m) for mi in needed[k] for m in parts(mi)})
        js = J.mlp_jets(flats[site_net[k]], dims, act, [column(c) for c in deps], want or [()], skip=skip, actp=actp, mono=mono)
        jets[k] = {mi: sum(js[local(m)] for m in parts(mi)) for mi in needed[k]}       # (N, n_out)
    syms = np.stack([jets[prog.g.nodes[i][1]][prog.g.nodes[i][3]][:, prog.g.nodes[i][2]]
                     for i in prog.symbols]).astype(wdt)
    n_eq = len(prog.residuals)
    seed = 1.0 / (n * prog.loss_norm)
    gtheta = None
    if prog.n_theta or prog.n_data:      # trainable scalars of the equations / per-point data columns (inverse problems)
        data = [np.asarray(t.detach().numpy(), wdt).reshape(-1) for t in prog.g.data]
        theta = [float(t[0].detach().reshape(-1)[t[1]]) if isinstance(t, tuple) else float(t.detach()) for t in prog.g.params]
        resid, funcs, gbar, lterm, gth = run_cpu(prog, coords, syms, seed, return_loss=True, f64=f64, data=data, theta=theta)
        gtheta = gth.astype(np.float64).sum(axis=1)[:prog.n_theta]
        host_closure.last_gtheta = gtheta
    else:
        resid, funcs, gbar, lterm = run_cpu(prog, coords, syms, seed, return_loss=True, f64=f64)
    r64 = resid.astype(np.float64)
    term = {"l2": lambda r: (r ** 2).sum(), "l1": lambda r: np.abs(r).sum(), "infinity": lambda r: np.abs(r).max(axis=0).sum()}
    loss = float(lterm.astype(np.float64).sum() * seed) if callable(loss) else float(term[loss](r64) * seed)
    # parameter gradient: adjoint streams through the jet oracle's VJP
    grads = [0.0] * len(nets)
    for k in range(prog.n_sites):                       # every site of a network adds into that network's gradient
        dims, act, skip, actp, mono = dims_act[site_net[k]]
        deps = prog.streams[k].deps
        gb = {}
        for idx, i in enumerate(prog.symbols):
            _, kk, o, mi = prog.g.nodes[i]
            if kk == k:
                for part in parts(mi):
                    m = gb.setdefault(tuple(sorted(deps.index(c) for c in part)), np.zeros((n, dims[-1])))
                    m[:, o] += gbar[idx].astype(np.float64)
        if not gb:
            gb = {(): np.zeros((n, dims[-1]))}
        grads[site_net[k]] = grads[site_net[k]] + J.mlp_jets_vjp(flats[site_net[k]], dims, act, [column(c) for c in deps], gb,
                                                               skip=skip, actp=actp, mono=mono)
    for j, (perm, at) in enumerate(perms):              # back to torch parameter order (frozen scalars: no gradient entry)
        g = np.zeros(at)
        g[perm] = grads[j][:perm.size]
        # entries of tensors that are trainable scalars of the program (symbolic skip weights): their per-point adjoint sums
        start, pos = {}, 0
        for prm in nets[j].parameters():
            start[id(prm)] = pos
            pos += prm.numel()
        for jt, t in enumerate(prog.g.params):
            if isinstance(t, tuple) and id(t[0]) in start:
                g[start[id(t[0])] + t[1]] = gtheta[jt]
        grads[j] = g
    return prog, funcs.T, resid.T, loss, np.concatenate(grads)


@pytest.mark.parametrize("name,lap", [("c1", True), ("c2", True), ("c2", False), ("c3", True), ("c5", True), ("c5", False),
                                      ("c4", True), ("w1", True), ("w2", True), ("w3", True), ("w4", True), ("w5", True), ("w6", True), ("w7", True), ("w8", True),
                                      ("w9", True), ("w10", True), ("w29", True), ("w30", True), ("w11", True), ("w12", True), ("w13", True), ("w14", True), ("w15", True), ("w16", True), ("w17", True), ("w18", True), ("w19", True), ("w20", True), ("w21", True),
                                      ("w24", True), ("w25", True),       # w24 / w25: Resnets above 64 units (symbolic skip connection)
                                      ("w26", True), ("w27", True), ("w28", True)])       # w26 / w27: per-layer widths above 64 units
def test_fused_pipeline_on_host_matches_reference(golden_dir, name, lap):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(0)
    cfg = configs.make(name, SIZES[name])
    prog, funcs, resid, loss, grad = host_closure(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), gold["coords"],
                                                  gold["params0"], lap, configs.func_val(cfg))
    resid = resid[:, :gold["residuals_f64"].shape[1]]      # Sobolev losses trace extra gradient columns behind the residuals
    if name in ("c2", "c5"):     # Laplace / Navier-Stokes use u_xx + u_yy only: the merge must be found when allowed
        assert any(prog.g.nodes[i][3][:1] == ("L",) for i in prog.symbols) == lap
    assert rel_l2(funcs, gold["funcs_f64"]) < 1e-5
    assert rel_l2(resid, gold["residuals_f64"]) < 1e-5
    assert abs(loss - float(gold["loss_f64"])) <= 1e-5 * abs(float(gold["loss_f64"]))
    assert rel_l2(grad, gold["grad_f64"]) < 1e-5


# stream set the tracer must arrive at for each zoo system: per network (first, mask2, lap)
ZOO_STREAMS = {"pendulum": [(1, 1, 0)], "coupled_sin": [(1, 0, 0)] * 2, "bvp_tanh": [(1, 1, 0)], "helmholtz_xy": [(1, 7, 0)],
               "advection": [(1, 0, 0)], "heat_wide": [(1, 1, 0)], "stokes_like": [(1, 5, 1), (1, 5, 1), (1, 0, 0)],
               "poisson3d": [(1, 41, 1)], "hessian3d": [(1, 63, 0)], "shell": [(1, 41, 0)],
               "swish_laplace": [(1, 5, 1)], "sigmoid_mixed": [(1, 7, 0)], "swish_ode": [(1, 1, 0), (1, 0, 0)],
               "bundle_decay": [(1, 0, 0)], "bundle_bvp": [(1, 1, 0)], "shape_64x2": [(1, 5, 1)], "shape_32x3": [(1, 5, 1)],
               "shape_48x2": [(1, 5, 1)], "shape_16x2_sin": [(1, 5, 1)], "shape_32x1": [(1, 5, 1)],
               "aptx_burgers": [(1, 1, 0)], "resnet_laplace": [(1, 5, 1)], "resnet_ode": [(1, 1, 0)],
               "swish_tr_laplace": [(1, 5, 1)], "aptx_tr_laplace": [(1, 5, 1)], "aptx_tr_wide": [(1, 5, 1)],
               "swish_tr_system": [(1, 1, 0), (1, 0, 0)], "aptx_tr_resnet": [(1, 1, 0)],
               "swish_fixed_laplace": [(1, 5, 1)], "aptx_fixed_laplace": [(1, 5, 1)], "ensemble_lv": [(1, 0, 0)],
               "mono_laplace": [(1, 7, 0)], "mono_ode": [(1, 1, 0)], "mono_poisson": [(1, 5, 1)],
               "shape_64_32": [(1, 5, 1)], "shape_24_40_12_sigmoid": [(1, 5, 1)],
               "shape_50x2": [(1, 5, 1)], "shape_20x3": [(1, 5, 1)], "shape_40x2_sigmoid": [(1, 5, 1)], "shape_10x1": [(1, 5, 1)],
               "shape_32x6": [(1, 5, 1)], "shape_16x8_sin": [(1, 5, 1)],
               # four / five inputs: 10 / 15 pair bits -- xx, yy, zz of (x, y, z, t) merged into one Laplacian stream; cc and ae
               "heat4d": [(1, 145, 1)], "mix5d": [(1, 528, 0)], "bundle_osc": [(1, 1, 0)],
               "piecewise_source": [(1, 5, 1)], "relu_ode": [(1, 0, 0)], "atan2_adv": [(1, 0, 0)],
               "rounding_ode": [(1, 0, 0)], "activations_ode": [(1, 0, 0)], "special_2d": [(1, 0, 0)], "autograd_grad_ode": [(1, 1, 0)],
               # third-order streams: (first, mask2, lap, mask3); the triple xxx brings its pair xx along
               "kdv": [(1, 1, 0, 1)], "ode3": [(1, 1, 0, 1)],
               # fourth-order streams: (first, mask2, lap, mask3, mask4); a quadruple brings its pairs and triples along --
               # biharmonic: xxxx, xxyy, yyyy (bits 0, 2, 4) need every pair and every triple of two coordinates
               "beam": [(1, 1, 0, 1, 1)], "beam_sigmoid": [(1, 1, 0, 1, 1)], "biharmonic": [(1, 7, 0, 15, 21)],
               "kuramoto": [(1, 1, 0, 1, 1)]}


@pytest.mark.parametrize("name", zoo.NAMES)
def test_zoo_on_host_matches_autograd_oracle(name):
    """Systems outside the BASELINE set: tracer + generated code + jet oracle against the fp64 autograd oracle built from
    independently written re-parameterisations (tests/zoo.py)."""
    from oracle import autograd_ref as R
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(40, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]).astype(np.float32),
                                                  flat.double().numpy())
    got_streams = [(int(st.first), int(st.mask2), int(st.lap)) + ((int(st.mask3),) if st.mask3 else ())
                   + ((int(st.mask4),) if st.mask4 else ()) for st in (prog.streams[k] for k in range(len(nets)))]
    assert got_streams == ZOO_STREAMS[name]
    # the oracle ran on the fp64 coordinates, the host pipeline on their fp32 rounding: tolerance 1e-5 covers it
    assert rel_l2(funcs, want["funcs"].numpy()) < 1e-5
    assert rel_l2(resid, want["residuals"].numpy()) < 1e-5
    assert abs(loss - want["loss"].item()) <= 1e-5 * abs(want["loss"].item())
    assert rel_l2(grad, want_grad) < 1e-5


@pytest.mark.parametrize("name", ["pendulum", "coupled_sin", "helmholtz_xy", "stokes_like", "sigmoid_mixed", "kdv", "shell",
                                  "bundle_bvp", "mono_laplace", "aptx_burgers", "piecewise_source", "relu_ode", "atan2_adv",
                                  "rounding_ode", "activations_ode", "special_2d", "autograd_grad_ode"])
def test_zoo_fp64_build_on_host_matches_autograd_oracle(name):
    """The fp64 build of the generated pointwise code (codegen.source_f64: types, math calls and literal suffixes of the
    same traced program rewritten for double -- what FusedSystem(dtype=float64) compiles for gfx950) between the fp64
    jet oracle's streams and VJP, against the fp64 autograd oracle: 1e-11, i.e. nothing in the program is left in fp32
    (a stray ``float`` or a 9-digit literal would show at 1e-7)."""
    from oracle import autograd_ref as R
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    for net in nets:
        net.double()
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for net in nets:
            for prm in net.parameters():
                prm.add_(1e-9 * torch.randn(prm.shape, generator=gen, dtype=torch.float64))
    flat = R.get_flat(nets)
    coords = system.sample(40, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    for net in nets:
        net.float()                                   # describe() / the tracer look at fp32 modules; values come from ``flat``
    prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]), flat.numpy(), f64=True)
    assert rel_l2(funcs, want["funcs"].numpy()) < 1e-11
    assert rel_l2(resid, want["residuals"].numpy()) < 1e-11
    assert abs(loss - want["loss"].item()) <= 1e-11 * abs(want["loss"].item())
    assert rel_l2(grad, want_grad) < 1e-11


@pytest.mark.parametrize("name,kind", [("helmholtz_xy", "l1"), ("stokes_like", "l1"), ("stokes_like", "infinity"),
                                       ("pendulum", "infinity")])
def test_l1_and_infinity_losses_on_host_match_autograd_oracle(name, kind):
    """losses.py:4-12: mean |r| and mean over points of max_e |r_e| as per-point terms of the generated code."""
    from oracle import autograd_ref as R
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(40, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords, loss=kind)
    want_grad = R.get_flat_grad(onets).numpy()
    prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]).astype(np.float32),
                                                  flat.double().numpy(), loss=kind)
    assert abs(loss - want["loss"].item()) <= 1e-5 * abs(want["loss"].item())
    assert rel_l2(grad, want_grad) < 1e-5


@pytest.mark.parametrize("name,kind", [("coupled_sin", "h1"), ("advection", "h1"), ("advection", "h1 semi"),
                                       # second-order systems: the extra d r/dx needs THIRD-order network streams
                                       ("helmholtz_xy", "h1"), ("pendulum", "h1"), ("pendulum", "h1 semi"),
                                       ("bvp_tanh", "h1"), ("poisson3d", "h1 semi")])
def test_sobolev_losses_on_host_match_autograd_oracle(name, kind):
    """losses.py:17-26: the h1 norms are the l2 loss of the residual list extended by d(sum_e r_e)/dx_a (what
    solvers.BaseSolver._fused_system traces); on second-order systems that asks for third-order streams."""
    from neurodiffeq_amd import diff
    from oracle import autograd_ref as R
    torch.manual_seed(11)
    system = zoo.build(name)
    nets, conds, pde = system.product()
    n_funcs = len(nets)

    def extended(*variables):
        res = list(pde(*variables))
        total = res[0]
        for r in res[1:]:
            total = total + r
        grads = [diff(total, x) for x in variables[n_funcs:]]
        return grads if kind == "h1 semi" else res + grads
    flat = R.get_flat(nets)
    coords = system.sample(40, seed=5)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords, loss=kind)
    want_grad = R.get_flat_grad(onets).numpy()
    prog, funcs, resid, loss, grad = host_closure(nets, conds, extended, np.stack([c.numpy() for c in coords]).astype(np.float32),
                                                  flat.double().numpy())
    assert abs(loss - want["loss"].item()) <= 1e-5 * abs(want["loss"].item())
    assert rel_l2(grad, want_grad) < 1e-5


CUSTOM_LOSSES = {
    # weighted residual + a term on the function values + a constant (criterion callable, solvers.py:216-226)
    "weighted": lambda r, f, x: ((1.0 + x[0] ** 2) * r ** 2).mean() + 0.1 * (f[0] ** 2).mean() + 0.25,
    # torch.nn loss modules are wrapped by the solver as criterion(r, zeros_like(r))
    "mse_module": lambda r, f, x: torch.nn.MSELoss()(r, torch.zeros_like(r)),
    "l1_module": lambda r, f, x: torch.nn.L1Loss()(r, torch.zeros_like(r)),
    # column-wise means with per-equation weights
    "per_equation": lambda r, f, x: ((r ** 2).mean(dim=0) * torch.tensor([1.0, 3.0, 0.5][:r.shape[1]])).sum(),
    # a loss that differentiates again: needs streams the residual alone does not
    "with_gradient": lambda r, f, x: (r ** 2).mean() + 0.05 * (configs.diff(f[0], x[0]) ** 2).mean(),
}


@pytest.mark.parametrize("kind", sorted(CUSTOM_LOSSES))
@pytest.mark.parametrize("name", ["c2", "c1", "c5"])
def test_custom_loss_callables_are_traced_and_match_autograd_oracle(name, kind):
    """VERDICT r1 missing #3: user loss_fn / additional_loss are pointwise expressions under a batch mean; traced to a
    per-point term + its adjoint, they give the loss and parameter gradient torch autograd gives for the same callable."""
    from oracle import autograd_ref as R
    torch.manual_seed(0)
    cfg = configs.make(name, SIZES[name])
    params = R.get_flat(cfg["nets"]).double().numpy()
    torch.manual_seed(5)
    ex = cfg["gen"].get_examples()
    coords = [c.detach() for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
    fn = CUSTOM_LOSSES[kind]
    metric = lambda *a: (a[0] * a[0]).mean() + configs.diff(a[0], a[len(cfg["nets"])]).mean()
    prog, funcs, resid, loss, grad = host_closure(cfg["nets"], cfg["conds"], cfg["pde"], np.stack([c.numpy() for c in coords]),
                                                  params, True, None, fn, metrics=[metric])
    assert prog.loss == "custom"
    ocfg = R.build_config(name, SIZES[name], dtype=torch.float64)
    R.set_flat(ocfg["nets"], torch.from_numpy(params))
    batch = [c.double().reshape(-1, 1).requires_grad_(True) for c in coords]
    of = [e(n, *batch) for n, e in zip(ocfg["nets"], ocfg["enforcers"])]
    ores = torch.cat(ocfg["pde"](*of, *batch), dim=1)
    import neurodiffeq_amd
    want = fn(ores, of, batch)
    want_metric = metric(*of, *batch).item()
    want.backward()
    want_grad = R.get_flat_grad(ocfg["nets"]).numpy()
    assert abs(loss - want.item()) <= 1e-5 * abs(want.item()), (loss, want.item())
    assert rel_l2(grad, want_grad) < 1e-5
    nf = len(cfg["nets"])
    assert abs(float(funcs[:, nf].astype(np.float64).mean()) - want_metric) <= 1e-5 * abs(want_metric)


def test_losses_outside_the_traced_family_raise_trace_unsupported():
    from neurodiffeq_amd.symbolic import TraceUnsupported
    torch.manual_seed(0)
    cfg = configs.make("c2", 8)
    for bad in (lambda r, f, x: (r ** 2).sum(),                         # batch sum: needs N
                lambda r, f, x: (r ** 2).mean() * (f[0] ** 2).mean(),   # product of two means
                lambda r, f, x: (r ** 2).max(),                         # not a mean
                lambda r, f, x: (r ** 2).mean().item()):                # leaves the graph
        with pytest.raises(TraceUnsupported):
            trace(cfg["nets"], cfg["conds"], cfg["pde"], 2, True, None, bad)


def test_unsupported_constructs_raise_trace_unsupported():
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.symbolic import TraceUnsupported
    g = Graph(1)
    with trace_scope(g):
        t = Sym(g, g.coord(0), leaf=True)
        with pytest.raises(TraceUnsupported):
            torch.cumsum(t, 0)
        with pytest.raises(TraceUnsupported):
            bool(t)
        assert g.cval(diff(3.0 * t * t, t, order=3).i) == 0.0
        assert g.cval(diff(t ** 2, t, order=2).i) == 2.0
        # round 5: what stays outside the traced family says so -- operations across points, shapes the column semantics do
        # not carry, in-place methods, multi-element constants made from a column
        for bad in (lambda: t - t.mean(), lambda: torch.roll(t, 1, 0), lambda: t.flatten(), lambda: t.squeeze(1), lambda: diff(t.detach(), t),
                    lambda: torch.stack([t, t]), lambda: t.floor_(), lambda: t.new_ones(5), lambda: t[:, 0], lambda: t[3],
                    lambda: torch.lgamma(t), lambda: torch.einsum("ij->i", t), lambda: diff(t ** 5, t, order=1) @ t):
            with pytest.raises(TraceUnsupported):
                bad()
        # ... and the spellings of "the whole column" / one-element constants that are inside it
        # (views are NEW tensors in torch: the same node, but not the coordinate itself as a diff() target -- round 6)
        assert t[:, 0:1].i == t.i and t[...].i == t.i and t[:, :].i == t.i and t.expand_as(t).i == t.i and t.contiguous() is t
        assert t.leaf and not t[...].leaf and not t.clone().leaf and not (t + 0.0).leaf and not t.view(-1, 1).leaf
        for derived in (t + 0.0, t * 1.0, t.clone(), t.view(-1, 1), t[:, 0:1], t.reshape(t.shape[0], 1)):
            assert derived.i == t.i
            with pytest.raises(TraceUnsupported, match="derived from a batch coordinate"):
                diff(t * t, derived)
        assert float(t.new_ones(1)) == 1.0 and float(t.new_zeros(1, 1)) == 0.0 and float(t.new_tensor(2.5)) == 2.5
        assert t.dtype == torch.get_default_dtype() and t.floor().i != t.i and torch.frac(t).i == (t - torch.trunc(t)).i
        assert g.cval(diff(torch.floor(t) * 2.0, t).i) == 0.0        # piecewise constant: zero derivative
        assert g.cval(diff(t * t.detach(), t).i) is None and diff(t * t.detach(), t).i == t.detach().i      # d/dt [t sg(t)] = sg(t)


def test_trainable_and_per_point_tensors_are_not_baked_into_the_kernel():
    """ADVICE r1 / VERDICT r2 #7: an nn.Parameter coefficient (inverse problems) or an (N, 1) data column inside
    ``diff_eqs`` is never frozen into / mis-broadcast by the generated kernel: since round 3 a trainable SCALAR is a
    kernel argument with a gradient (leaf 'param'), an (N, 1) column an input row (leaf 'data'), torch expressions of
    the two get a symbolic twin; trainable vectors and other shapes still send the system to the composite path.  A plain
    scalar tensor is a constant of the trace and its in-place modification is visible through the version counter."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.engine import trace_system
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.symbolic import TraceUnsupported
    net, cond = FCNN(1, 1, hidden_units=(32, 32)), IVP(0.0, 1.0)
    k = torch.nn.Parameter(torch.tensor(2.0))
    data = torch.linspace(0, 1, 7).reshape(-1, 1)
    prog, _ = trace_system([net], [cond], lambda u, t: [diff(u, t) + k * u - torch.exp(-k) * data], 1)
    assert prog.n_theta == 1 and prog.g.params[0] is k and prog.n_data == 1 and prog.g.data[0] is data
    assert not prog.g.captured                      # nothing of it was baked in as a constant
    kv = torch.nn.Parameter(torch.ones(3))
    with pytest.raises(TraceUnsupported, match="trainable"):
        trace_system([net], [cond], lambda u, t: [diff(u, t) + (kv * u).sum(dim=1, keepdim=True)], 1)
    with pytest.raises(TraceUnsupported, match="shape"):
        trace_system([net], [cond], lambda u, t: [diff(u, t) - torch.ones(4, 4, 2)], 1)
    with pytest.raises(TraceUnsupported):
        trace_system([net], [cond], lambda u, t: [torch.cat([u, u], dim=1)], 1)
    c = torch.tensor(3.0)
    prog, _ = trace_system([net], [cond], lambda u, t: [diff(u, t) + c * u], 1)
    assert [(t is c, v, val) for t, v, val in prog.g.captured] == [(True, c._version, c.item())]
    from neurodiffeq_amd.symbolic import captured_unchanged
    assert captured_unchanged(prog.g)
    c.data.mul_(0.5)                                 # (no version bump: seen by value)
    assert not captured_unchanged(prog.g)
    c.data.mul_(2.0)
    assert captured_unchanged(prog.g)
    c.mul_(2.0)
    assert prog.g.captured[0][1] != c._version


def test_assembly_fixup_separates_packed_valu_from_mfma():
    """_hipcc.fix_pk_mfma: one wait state between a packed-fp32 VALU instruction and a directly following MFMA (labels
    and comments in between do not count), nothing anywhere else, idempotent."""
    from neurodiffeq_amd import _hipcc
    asm = "\n".join([
        "\tv_pk_mul_f32 v[2:3], v[10:11], v[58:59] op_sel:[0,1]",
        "\tv_mfma_f32_16x16x32_bf16 a[52:55], v[116:119], v[82:85], a[52:55]",
        "\tv_pk_fma_f32 v[4:5], v[6:7], v[8:9], v[4:5]",
        "; a comment",
        ".LBB0_3:",
        "\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]",
        "\tv_pk_add_f32 v[4:5], v[6:7], v[8:9]",
        "\ts_waitcnt lgkmcnt(0)",
        "\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]",
        "\tv_mul_f32_e32 v1, v2, v3",
        "\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]",
    ])
    out, sites = _hipcc.fix_pk_mfma(asm)
    assert sites == 2
    lines = out.split("\n")
    assert lines[1].strip() == "s_nop 0" and lines[2].startswith("\tv_mfma")
    assert lines[6].strip() == "s_nop 0" and lines[5] == ".LBB0_3:" and lines[7].startswith("\tv_mfma_f32_16x16x4")
    assert out.count("s_nop 0") == 2
    again, more = _hipcc.fix_pk_mfma(out)
    assert more == 0 and again == out


def test_build_pipeline_produces_a_loadable_library(tmp_path):
    """The split hipcc pipeline (device assembly -> fix-up -> assembler -> bundle -> host compile) yields a shared library
    that loads and exports its host symbols (no GPU needed: hipcc cross-compiles)."""
    import ctypes
    from neurodiffeq_amd import _hipcc
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\n__global__ void k(float* x) { x[threadIdx.x] += 1.f; }\n'
                   'extern "C" int answer() { return 42; }\n')
    so = str(tmp_path / "k.so")
    assert _hipcc.compile_shared(str(src), so) == 0
    assert ctypes.CDLL(so).answer() == 42


def test_assembly_fixup_scalarizes_packed_ops_that_cross_halves():
    """_hipcc.scalarize_pk: packed-fp32 instructions whose LOW half reads a HIGH source half (op_sel) -- the ones that
    misbehave on gfx950, DESIGN 4.6 -- become two scalar VOP3 instructions with the same operands, modifiers and order
    hazards respected; everything else is left alone."""
    from neurodiffeq_amd import _hipcc
    asm = "\n".join([
        "\tv_pk_mul_f32 v[2:3], v[10:11], v[58:59] op_sel:[0,1]",
        "\tv_pk_fma_f32 v[18:19], v[18:19], v[140:141], v[20:21] op_sel:[1,0,0]",
        "\tv_pk_fma_f32 v[4:5], v[8:9], v[46:47], v[4:5] op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]",
        "\tv_pk_mul_f32 v[60:61], v[202:203], 0 op_sel_hi:[1,0]",
        "\tv_pk_add_f32 v[6:7], v[6:7], v[8:9]",
        "\tv_pk_mul_f32 v[70:71], v[32:33], s[58:59] op_sel:[1,0] op_sel_hi:[0,0]",
        "\tv_pk_mul_f32 v[2:3], v[2:3], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]",
    ])
    out, done, skipped = _hipcc.scalarize_pk(asm, "opsel")
    lines = [l.strip() for l in out.split("\n")]
    assert done == 5 and skipped == 0
    assert lines[0:2] == ["v_mul_f32_e64 v2, v10, v59", "v_mul_f32_e64 v3, v11, v59"]
    assert lines[2:4] == ["v_fma_f32 v18, v19, v140, v20", "v_fma_f32 v19, v19, v141, v21"]
    assert lines[4:6] == ["v_fma_f32 v4, -v8, v47, v4", "v_fma_f32 v5, -v9, v47, v5"]
    assert lines[6].startswith("v_pk_mul_f32 v[60:61]") and lines[7].startswith("v_pk_add_f32")
    assert lines[8:10] == ["v_mul_f32_e64 v70, v33, s58", "v_mul_f32_e64 v71, v32, s58"]
    # halves that would clobber each other's sources: the pair product x.lo * x.hi in both halves -> once, then a copy
    assert lines[10:12] == ["v_mul_f32_e64 v2, v3, v2", "v_mov_b32_e32 v3, v2"]
    # ... the general case: exchange the destination registers first
    crossed, n_c, n_s = _hipcc.scalarize_pk("\tv_pk_mul_f32 v[2:3], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]", "opsel")
    assert (n_c, n_s) == (1, 0) and [l.strip() for l in crossed.split("\n")] == \
        ["v_swap_b32 v2, v3", "v_mul_f32_e64 v2, v2, v4", "v_mul_f32_e64 v3, v3, v5"]
    every, n_all, _ = _hipcc.scalarize_pk(asm, "all")
    assert n_all == 7 and every.count("v_pk_") == 0
    # what is left over fails the build (the pass fails closed) -- with ONE exception: a genuine two-in / two-out update
    # (each half reads both destination registers: no two-instruction rewrite exists) may stay in a module that does not
    # spill; the only unexplained failure of op_sel forms was in spilling kernels, the direct hazard is covered by the nop
    two = "\tv_pk_fma_f32 v[2:3], v[2:3], v[2:3], v[6:7] op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
    _hipcc.verify_fixup(two + "; ScratchSize: 0\n")
    with pytest.raises(RuntimeError, match="op_sel left in the output"):
        _hipcc.verify_fixup(two + "; ScratchSize: 0\n; ScratchSize: 224\n")
    with pytest.raises(RuntimeError, match="op_sel left in the output"):
        _hipcc.verify_fixup("\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]\n; ScratchSize: 0\n")
    with pytest.raises(RuntimeError, match="directly followed by an MFMA"):
        _hipcc.verify_fixup("\tv_pk_add_f32 v[2:3], v[4:5], v[6:7]\n\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]")
    # operands the rewrite does not understand (output modifiers, special registers) are never touched
    odd = "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] clamp\n\tv_pk_add_f32 v[0:1], v[2:3], vcc op_sel:[0,1]"
    same, n_odd, n_skip = _hipcc.scalarize_pk(odd, "opsel")
    assert same == odd and n_odd == 0 and n_skip == 2


def test_closure_kernel_build_is_chosen_from_the_tile_round_count():
    """engine.FusedSystem.prefers_wide: 8-wave workgroups (2 048 waves per round at 6.7 us) against 4-wave ones (1 024 at
    4.1 us), measured on MI355X: the BASELINE headline size and every large batch go to the 8-wave build,
    small batches and the sizes where it would need as many rounds for more money do not."""
    from neurodiffeq_amd.engine import FusedSystem
    wide = FusedSystem.prefers_wide
    assert not any(wide(n) for n in (32, 1024, 16384, 32768, 33124, 40000, 81920))
    assert all(wide(n) for n in (49284, 57600, 65536, 98304, 131072, 262144, 1 << 20, 1 << 22))


def test_scalarized_packed_ops_compute_what_the_packed_ones_do():
    """Property test of _hipcc.scalarize_pk on random instructions: a small interpreter executes the packed form (ISA
    semantics: per half, source i is the low or high register of its pair as op_sel / op_sel_hi say, negated as neg_lo /
    neg_hi say, all sources read before the destination pair is written) and the rewritten scalar sequence (executed in
    order) on the same random register file -- the register files must end up identical, including when the destination
    overlaps the sources."""
    import random
    import re
    import numpy as np
    from neurodiffeq_amd import _hipcc
    rng = random.Random(7)
    f32 = np.float32

    def fma(op, a, b, c=None):
        return f32(a * b) if op == "mul" else f32(a + b) if op == "add" else f32(np.float32(np.float64(a) * np.float64(b) + np.float64(c)))

    def run_packed(regs, op, dst, srcs, sel, sel_hi, neg_lo, neg_hi):
        def val(s, which, neg):
            v = regs[s + which] if isinstance(s, int) else f32(s)
            return f32(-v) if neg else v
        lo = fma(op, *[val(s, sel[i] if isinstance(s, int) else 0, neg_lo[i]) for i, s in enumerate(srcs)])
        hi = fma(op, *[val(s, sel_hi[i] if isinstance(s, int) else 0, neg_hi[i]) for i, s in enumerate(srcs)])
        regs[dst], regs[dst + 1] = lo, hi

    def run_scalar(regs, line):
        mv = re.match(r"\s*v_(mov_b32_e32|swap_b32)\s+v(\d+),\s*v(\d+)$", line)
        if mv:
            a_, b_ = int(mv.group(2)), int(mv.group(3))
            if mv.group(1) == "swap_b32":
                regs[a_], regs[b_] = regs[b_], regs[a_]
            else:
                regs[a_] = regs[b_]
            return
        m = re.match(r"\s*v_(mul|add|fma)_f32(?:_e64)?\s+v(\d+),\s*(.*)$", line)
        op, d, rest = m.group(1), int(m.group(2)), [t.strip() for t in m.group(3).split(",")]
        vals = []
        for t in rest:
            neg = t.startswith("-") and t[1:].startswith("v")
            t2 = t[1:] if neg else t
            v = regs[int(t2[1:])] if t2.startswith("v") else f32(float(t2))
            vals.append(f32(-v) if neg else v)
        regs[d] = fma(op, *vals)

    checked = 0
    for _ in range(3000):
        op = rng.choice(["mul", "add", "fma"])
        nsrc = 3 if op == "fma" else 2
        dst = rng.randrange(0, 12, 1)
        srcs = [rng.choice([rng.randrange(0, 12), rng.choice([0.5, 2.0, -1.0, 4.0])]) if rng.random() < 0.15
                else rng.randrange(0, 12) for _ in range(nsrc)]
        sel = [rng.randint(0, 1) for _ in range(nsrc)]
        sel_hi = [rng.randint(0, 1) for _ in range(nsrc)]
        neg_lo = [int(rng.random() < 0.2) for _ in range(nsrc)]
        neg_hi = [int(rng.random() < 0.2) for _ in range(nsrc)]
        for i, s in enumerate(srcs):                     # constants: encoded the way the compiler does (no selects, no neg)
            if not isinstance(s, int):
                sel[i], sel_hi[i], neg_lo[i], neg_hi[i] = 0, 0, 0, 0
        text = f"\tv_pk_{op}_f32 v[{dst}:{dst + 1}], " + ", ".join(f"v[{s}:{s + 1}]" if isinstance(s, int) else repr(s) for s in srcs)
        text += f" op_sel:[{','.join(map(str, sel))}] op_sel_hi:[{','.join(map(str, sel_hi))}]"
        text += f" neg_lo:[{','.join(map(str, neg_lo))}] neg_hi:[{','.join(map(str, neg_hi))}]"
        out, done, skipped = _hipcc.scalarize_pk(text, "all")
        if not done:
            assert skipped == 1 and out == text
            continue
        base = [f32(rng.uniform(-3, 3)) for _ in range(14)]
        a, b = list(base), list(base)
        run_packed(a, op, dst, srcs, sel, sel_hi, neg_lo, neg_hi)
        for line in out.split("\n"):
            run_scalar(b, line)
        assert all(x == y or (np.isnan(x) and np.isnan(y)) for x, y in zip(a, b)), (text, out)
        checked += 1
    assert checked > 2000


@pytest.mark.parametrize("name", ["inv1", "inv2"])
def test_inverse_problem_closure_on_host_matches_reference(golden_dir, name):
    """nn.Parameter coefficients inside the equations become kernel arguments whose gradient is one more
    sum of per-point adjoints; (N, 1) data tensors become input rows.  Tracer + generated code (gcc) + jet oracle against
    what the unmodified reference computed (tests/golden/inv1.npz: Burgers with trainable viscosity and advection
    amplitude; inv2.npz: Poisson with a measured source column and two trainable scalars)."""
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    cfg = configs.make_inverse(name)
    assert np.array_equal(R.get_flat(cfg["nets"]).numpy(), gold["params0"])
    if cfg["data"]:
        assert np.array_equal(np.stack([d.numpy().reshape(-1) for d in cfg["data"]]), gold["data"])
    prog, funcs, resid, loss, grad = host_closure(cfg["nets"], cfg["conds"], cfg["pde"], gold["coords"], gold["params0"])
    assert prog.n_theta == 2 and prog.n_data == len(cfg["data"])
    order = [next(j for j, t in enumerate(prog.g.params) if t is p) for p in cfg["theta"]]    # leaves are numbered by first use
    assert rel_l2(funcs, gold["funcs_f64"]) < 1e-5 and rel_l2(resid, gold["residuals_f64"]) < 1e-5
    assert abs(loss - float(gold["loss_f64"])) <= 1e-5 * abs(float(gold["loss_f64"]))
    assert rel_l2(grad, gold["grad_f64"]) < 1e-5
    assert rel_l2(host_closure.last_gtheta[order], gold["grad_theta_f64"]) < 1e-5


@pytest.mark.parametrize("name", ["inv1", "inv2"])
def test_inverse_problem_trajectory_on_the_composite_path(golden_dir, name):
    """The same problems through the Solver on a host without a GPU (composite path = the reference's closure): three epochs
    of Adam over the network AND the coefficients reproduce the reference's trajectory."""
    import warnings
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    solver, cfg = configs.make_inverse_solver(name)
    solver.fused = "off"
    torch.manual_seed(int(gold["seed"]) + 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):
            solver.run_train_epoch()
    assert np.allclose(solver.metrics_history["train_loss"], gold["traj_loss"], rtol=2e-5)
    assert rel_l2(R.get_flat(cfg["nets"]).numpy(), gold["traj_params"]) < 1e-5
    assert np.allclose([p.item() for p in cfg["theta"]], gold["traj_theta"], rtol=1e-5)


def test_twin_of_a_tensor_updated_in_place_is_dropped():
    """``a = k * f; a += 1.0`` -- the twin recorded for ``a`` describes the old value; the tensor must be refused
    (composite path), not traced as k * f.  Slices of a twinned column are not the column."""
    from neurodiffeq_amd.symbolic import Graph, trace_scope, TraceUnsupported, _as_node
    g = Graph(1)
    k = torch.nn.Parameter(torch.tensor(2.0))
    f = torch.linspace(0, 1, 8).reshape(-1, 1)
    with trace_scope(g):
        a = k * f
        assert isinstance(_as_node(g, a), int)
        a += 1.0
        with pytest.raises(TraceUnsupported):
            _as_node(g, a)
        b = k * f
        b.mul_(3.0)
        with pytest.raises(TraceUnsupported):
            _as_node(g, b)
        c = k * f
        with pytest.raises(TraceUnsupported):
            _as_node(g, c[0:4])
        assert _as_node(g, c[:, 0:1]) == _as_node(g, c)
        assert _as_node(g, (k * f).reshape(-1, 1)) == _as_node(g, k * f)


def test_resample_generator_larger_than_its_source_is_not_fixed_size():
    from neurodiffeq_amd.generators import Generator1D, ResampleGenerator, draws_have_fixed_size
    g = Generator1D(16, 0.0, 1.0)
    assert draws_have_fixed_size(ResampleGenerator(g, size=8))
    assert not draws_have_fixed_size(ResampleGenerator(g, size=32, replacement=False))
    assert draws_have_fixed_size(ResampleGenerator(g, size=32, replacement=True))


def test_equation_probe_sees_python_state_changed_between_epochs():
    """``lambda u, t: [diff(u, t) + nu['v'] * u]`` -- the float is a literal of the generated kernel.  The
    state watch notices that nu['v'] moved, the re-trace (program.eq_probe) that the equations now compute something else;
    state that moved without changing the equations leaves the probe true."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd._pystate import StateWatch
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.engine import trace_system
    from neurodiffeq_amd.networks import FCNN
    nu = {"v": 1.0, "unused": 3.0}
    eqs = lambda u, t: [diff(u, t) + nu["v"] * u]
    cond = IVP(0.0, 1.0)
    program, _ = trace_system([FCNN(1, 1)], [cond], eqs, 1)
    watch = StateWatch([eqs, cond])
    assert program.g.captured == [] and program.eq_probe() and not watch.dirty()
    src = program.point_fn_source()
    nu["v"] = 5.0
    assert watch.dirty() and not program.eq_probe()
    program2, _ = trace_system([FCNN(1, 1)], [cond], eqs, 1)
    assert program2.point_fn_source() != src
    nu["v"] = 1.0
    assert not watch.dirty() and program.eq_probe()
    nu["unused"] = 4.0                      # state the equations do not read: dirty watch, same trace
    assert watch.dirty() and program.eq_probe()
    cond.u_0 = 2.0                          # an attribute of the condition object: other function values
    assert StateWatch([eqs, cond]).entries and not program.eq_probe()
    # a stateless lambda has (almost) nothing to watch
    assert len(StateWatch([lambda u, t: [diff(u, t) + u]])) <= 2


def test_outside_numbers_that_move_become_runtime_constants_of_the_generated_kernel():
    """symbolic.Graph.external / engine.trace_system(volatile=...): a number the equations read from Python state is a literal of
    the first build; once a re-trace differs in such numbers only, ``suggest_volatile()`` names their positions and the next
    trace takes them as frozen 'param' leaves -- same per-point values as the literal program, no adjoint, and every later
    value is absorbed by ``eq_probe()`` (it refills the frozen scalars) without another build."""
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.engine import trace_system
    from neurodiffeq_amd.networks import FCNN
    nu = {"v": 0.05}
    k = torch.tensor(2.0)                                       # a captured 1-element tensor is an outside number as well
    cond = IVP(0.0, 1.5)
    eqs = lambda u, t: [diff(u, t) + nu["v"] * u ** 2 - k * torch.sin(t) + 3.0]
    net = FCNN(1, 1)
    lit, _ = trace_system([net], [cond], eqs, 1)
    assert lit.n_theta == 0 and lit.eq_probe() and lit.suggest_volatile() == frozenset()
    nu["v"] = 0.035
    k.mul_(1.5)
    assert not lit.eq_probe()
    vol = lit.suggest_volatile()
    assert len(vol) == 2                                        # nu['v'] and k; the IVP's 1.5, the literal 3.0 stay literals
    run, _ = trace_system([net], [cond], eqs, 1, volatile=vol)
    assert run.n_theta == 2 and run.g.frozen == {0, 1} and run.eq_probe()
    lit2, _ = trace_system([net], [cond], eqs, 1)              # the literal program of the CURRENT values
    rng = np.random.default_rng(0)
    n = 257
    coords = rng.uniform(0.1, 2.0, (1, n)).astype(np.float32)
    syms = rng.normal(size=(len(lit2.symbols), n)).astype(np.float32)
    assert [lit2.g.nodes[i] for i in lit2.symbols] == [run.g.nodes[i] for i in run.symbols]
    theta = [float(p) for p in run.g.params]
    assert theta == [0.035, 3.0]
    r0, f0, g0 = run_cpu(lit2, coords, syms, 1.0 / n)
    r1, f1, g1, gth = run_cpu(run, coords, syms, 1.0 / n, theta=theta)
    assert np.allclose(r0, r1, rtol=2e-6, atol=1e-6) and np.array_equal(f0, f1) and np.allclose(g0, g1, rtol=2e-6, atol=1e-7)
    assert not gth.any()                                        # runtime constants carry no adjoint
    # a further change: absorbed by the probe, no new program
    nu["v"] = 0.0245
    assert run.eq_probe() and [float(p) for p in run.g.params] == [0.0245, 3.0]
    # a change of a number that is NOT volatile is still a different program ... whose suggestion adds that position
    cond.u_0 = 1.75
    assert not run.eq_probe() and len(run.suggest_volatile()) == 3
    cond.u_0 = 1.5
    # exponents stay compile-time constants: never promoted, a new value is a new program
    p = {"e": 2.0}
    eqs2 = lambda u, t: [diff(u, t) + u ** p["e"]]
    a, _ = trace_system([net], [cond], eqs2, 1)
    p["e"] = 3.0
    assert not a.eq_probe() and a.suggest_volatile() == frozenset()
    # fp64 programs have no scalar arguments: the hint is ignored, the numbers stay literals
    d, _ = trace_system([FCNN(1, 1).double()], [cond], eqs, 1, f64=True, volatile=vol)
    assert d.n_theta == 0


def test_fp64_closure_source_is_the_fp32_module_rewritten_for_double():
    """codegen.can_fuse_f64 / fused_source(f64=True): single-network systems on the plain closure kernel get the SAME generated
    module under NDQ_F64 -- types, math calls and literal suffixes rewritten, no loop / pull launcher (fp32 only) --, other
    systems keep the fp64 three-kernel pipeline."""
    import re
    from neurodiffeq_amd import codegen, engine
    from tests import configs

    def traced(name):
        torch.manual_seed(0)
        cfg = configs.make(name, None)
        for net in cfg["nets"]:
            net.double()
        return engine.trace_system(cfg["nets"], cfg["conds"], configs.fused_equations(cfg), configs.n_coords(cfg),
                                   compute_func_val=configs.func_val(cfg), f64=True)
    program, descs = traced("c2")
    assert codegen.can_fuse_f64(program, descs)
    src = program.fused_source(descs[0], f64=True)
    assert src.startswith("#define NDQ_F64 1\n")
    assert not re.search(r"\bfloat\b", src) and "double" in src
    assert not re.search(r"\d\.\d+f\b", src) and "expf(" not in src and "tanhf(" not in src
    assert "fused_closure_loop_kernel" not in src and "ndq_fused_launch_tv" in src
    assert src != program.fused_source(descs[0])                      # (and the fp32 module is untouched by the flag)
    assert "#define NDQ_F64" not in program.fused_source(descs[0])
    for name in ("c1", "c4"):                                         # two networks / the grouped closure: pipeline in double
        program, descs = traced(name)
        assert not codegen.can_fuse_f64(program, descs)
