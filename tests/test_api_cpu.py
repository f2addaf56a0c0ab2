"""CPU tests of the API mirror (composite path): the behaviours the reference pins in its own unit tests
(tests/test_neurodiffeq.py, test_operators_*.py, test_conditions.py, test_generators.py, test_networks.py,
test_losses.py, test_solvers.py), re-stated against neurodiffeq_amd.  fp64 where the reference uses fp64."""
import math
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

from neurodiffeq_amd import diff, safe_diff, unsafe_diff
from neurodiffeq_amd import operators as ops
from neurodiffeq_amd.conditions import (IVP, DirichletBVP, DirichletBVP2D, IBVP1D, NoCondition, EnsembleCondition,
                                        DirichletBVPSpherical, DirichletBVPSphericalBasis, BundleIVP, BundleDirichletBVP)
from neurodiffeq_amd.function_basis import RealSphericalHarmonics, HarmonicsLaplacian
from neurodiffeq_amd.generators import (Generator1D, Generator2D, Generator3D, GeneratorSpherical, ConcatGenerator,
                                        EnsembleGenerator, StaticGenerator, PredefinedGenerator, SamplerGenerator)
from neurodiffeq_amd.losses import _losses
from neurodiffeq_amd.networks import FCNN, SinActv, Swish, APTx, Resnet, MonomialNN
from neurodiffeq_amd.solvers import Solver1D, Solver2D, SolverSpherical, BundleSolver1D

F64 = torch.float64


def col(n=16, lo=0.1, hi=1.0):
    return (torch.rand(n, 1, dtype=F64) * (hi - lo) + lo).requires_grad_(True)


# ----------------------------------------------------------------------------------------------- diff
def test_diff_shape_contract_and_alias():
    u, t = torch.rand(10, 1, dtype=F64, requires_grad=True), None
    t = torch.rand(10, 1, dtype=F64, requires_grad=True)
    u = t ** 2
    for bad_u, bad_t in [(u.reshape(-1), t), (u, t.reshape(-1)), (u.reshape(5, 2), t.reshape(5, 2)), (u[:5], t)]:
        with pytest.raises(ValueError):
            safe_diff(bad_u, bad_t)
        with pytest.raises(ValueError):
            diff(bad_u, bad_t)
    assert torch.allclose(unsafe_diff(u.reshape(-1), t)[:, 0], 2 * t[:, 0])   # no shape check (legacy behaviour)
    assert torch.allclose(diff(u, t, shape_check=False), 2 * t)
    with pytest.warns(FutureWarning):
        assert torch.allclose(diff(x=u, t=t), 2 * t)


def test_diff_higher_orders_and_unused_variable():
    t = torch.linspace(-1, 1, 20, dtype=F64).reshape(-1, 1).requires_grad_(True)
    sq, ex = t ** 2, torch.exp(t)
    assert torch.allclose(diff(sq, t, order=1), 2 * t) and torch.allclose(diff(sq, t, order=2), 2 * torch.ones_like(t))
    for k in range(3, 8):
        d = diff(sq, t, order=k)
        assert torch.equal(d, torch.zeros_like(t)) and d.requires_grad
        assert torch.allclose(diff(ex, t, order=k), ex)
    s = torch.rand(20, 1, dtype=F64, requires_grad=True)
    z = diff(sq, s)
    assert torch.equal(z, torch.zeros_like(s)) and z.requires_grad


# ----------------------------------------------------------------------------------------------- operators
def _fields(n_in=3):
    torch.manual_seed(0)
    xs = [col(50) for _ in range(n_in)]
    net = FCNN(n_in, 3, hidden_units=(16, 16)).double()
    out = net(torch.cat(xs, dim=1))
    return [out[:, i:i + 1] for i in range(3)], xs


def test_cartesian_operators_equal_compositions_of_diff():
    (u, v, w), (x, y, z) = _fields()
    assert torch.equal(ops.div(u, v, w, x, y, z), diff(u, x) + diff(v, y) + diff(w, z))
    cx, cy, cz = ops.curl(u, v, w, x, y, z)
    assert torch.allclose(cx, diff(w, y) - diff(v, z)) and torch.allclose(cz, diff(v, x) - diff(u, y))
    assert torch.allclose(ops.laplacian(u, x, y, z), diff(u, x, order=2) + diff(u, y, order=2) + diff(u, z, order=2))
    gx, gy, gz = ops.grad(u * 0 + x ** 2, x, y, z)
    assert torch.allclose(gx, 2 * x) and torch.equal(gy, torch.zeros_like(y)) and gy.requires_grad
    with pytest.raises(RuntimeError):
        ops.div(u, v, x)


@pytest.mark.parametrize("system", ["cartesian", "spherical", "cylindrical"])
def test_vector_calculus_identities(system):
    (u, v, w), (a, b, c) = _fields()
    if system == "cartesian":
        G, D, C, L, VL = ops.grad, ops.div, ops.curl, ops.laplacian, ops.vector_laplacian
    elif system == "spherical":
        G, D, C, L, VL = (ops.spherical_grad, ops.spherical_div, ops.spherical_curl, ops.spherical_laplacian,
                          ops.spherical_vector_laplacian)
    else:
        G, D, C, L, VL = (ops.cylindrical_grad, ops.cylindrical_div, ops.cylindrical_curl, ops.cylindrical_laplacian,
                          ops.cylindrical_vector_laplacian)
    tol = 1e-8
    assert D(*C(u, v, w, a, b, c), a, b, c).abs().max() < tol                      # div curl = 0
    assert max(t.abs().max() for t in C(*G(u, a, b, c), a, b, c)) < tol            # curl grad = 0
    assert (D(*G(u, a, b, c), a, b, c) - L(u, a, b, c)).abs().max() < tol          # div grad = laplacian
    cc = C(*C(u, v, w, a, b, c), a, b, c)                                          # curl curl = grad div - vec lap
    gd = G(D(u, v, w, a, b, c), a, b, c)
    vl = VL(u, v, w, a, b, c)
    assert max((p - (q - r)).abs().max() for p, q, r in zip(cc, gd, vl)) < 1e-7


# ----------------------------------------------------------------------------------------------- conditions
def test_conditions_hold_exactly_on_the_boundary():
    torch.manual_seed(1)
    net = FCNN(1, 1).double()
    t0 = torch.full((8, 1), 0.3, dtype=F64, requires_grad=True)
    assert torch.allclose(IVP(0.3, 1.5).enforce(net, t0), torch.full_like(t0, 1.5))
    u = IVP(0.3, 1.5, u_0_prime=-0.7).enforce(net, t0)
    assert torch.allclose(u, torch.full_like(t0, 1.5)) and torch.allclose(diff(u, t0), torch.full_like(t0, -0.7))
    bvp = DirichletBVP(0.0, 1.0, 2.0, -3.0)
    assert torch.allclose(bvp.enforce(net, torch.zeros(4, 1, dtype=F64)), torch.ones(4, 1, dtype=F64))
    assert torch.allclose(bvp.enforce(net, torch.full((4, 1), 2.0, dtype=F64)), torch.full((4, 1), -3.0, dtype=F64))
    assert torch.equal(NoCondition().enforce(net, t0), net(t0))

    net2 = FCNN(2, 1).double()
    f0, f1 = (lambda y: torch.sin(math.pi * y)), (lambda y: 0 * y)
    g0, g1 = (lambda x: 0 * x), (lambda x: x * (1 - x))
    c = DirichletBVP2D(0, f0, 1, f1, 0, g0, 1, g1)
    s = torch.rand(16, 1, dtype=F64)
    zero, one = torch.zeros_like(s), torch.ones_like(s)
    assert torch.allclose(c.enforce(net2, zero, s), f0(s)) and torch.allclose(c.enforce(net2, one, s), f1(s))
    assert torch.allclose(c.enforce(net2, s, zero), g0(s)) and torch.allclose(c.enforce(net2, s, one), g1(s), atol=1e-12)


@pytest.mark.parametrize("kind", ["dd", "dn", "nd", "nn"])
def test_ibvp1d_variants(kind):
    torch.manual_seed(2)
    net = FCNN(2, 1).double()
    u0 = lambda x: torch.sin(math.pi * x)
    g, h = (lambda t: 0 * t), (lambda t: 0 * t)
    p, q = (lambda t: math.pi * torch.ones_like(t)), (lambda t: -math.pi * torch.ones_like(t))
    kw = dict(x_min_val=g) if kind[0] == "d" else dict(x_min_prime=p)
    kw.update(dict(x_max_val=h) if kind[1] == "d" else dict(x_max_prime=q))
    c = IBVP1D(0.0, 1.0, 0.0, u0, **kw)
    x = torch.rand(12, 1, dtype=F64, requires_grad=True)
    t = torch.rand(12, 1, dtype=F64, requires_grad=True)
    zero, one = torch.zeros_like(x).requires_grad_(True), torch.ones_like(x).requires_grad_(True)
    assert torch.allclose(c.enforce(net, x, torch.zeros_like(t)), u0(x), atol=1e-12)          # initial condition
    for edge, is_dirichlet, val, slope in ((zero, kind[0] == "d", g, p), (one, kind[1] == "d", h, q)):
        u = c.enforce(net, edge, t)
        if is_dirichlet:
            assert torch.allclose(u, val(t), atol=1e-10)
        else:
            assert torch.allclose(diff(u, edge), slope(t), atol=1e-8)
    with pytest.raises(NotImplementedError):
        IBVP1D(0, 1, 0, u0, x_min_val=g, x_min_prime=p)


def test_ensemble_and_spherical_conditions():
    torch.manual_seed(3)
    net = FCNN(1, 2).double()
    t0 = torch.zeros(5, 1, dtype=F64)
    out = EnsembleCondition(IVP(0.0, 1.0), IVP(0.0, -2.0)).enforce(net, t0)
    assert out.shape == (5, 2) and torch.allclose(out[:, 0], torch.ones(5, dtype=F64)) and torch.allclose(out[:, 1], -2 * torch.ones(5, dtype=F64))
    with pytest.raises(ValueError):
        EnsembleCondition(IBVP1D(0, 1, 0, lambda x: x, x_min_prime=lambda t: t, x_max_val=lambda t: t))
    net3 = FCNN(3, 1).double()
    f, g = (lambda th, ph: torch.cos(th)), (lambda th, ph: torch.sin(ph))
    c = DirichletBVPSpherical(0.5, f, 2.0, g)
    th, ph = torch.rand(6, 1, dtype=F64), torch.rand(6, 1, dtype=F64)
    assert torch.allclose(c.enforce(net3, torch.full_like(th, 0.5), th, ph), f(th, ph))
    assert torch.allclose(c.enforce(net3, torch.full_like(th, 2.0), th, ph), g(th, ph))
    netr = FCNN(1, 9).double()
    R0, R1 = torch.arange(9, dtype=F64), -torch.arange(9, dtype=F64)
    cb = DirichletBVPSphericalBasis(0.5, R0, 2.0, R1)
    assert torch.allclose(cb.enforce(netr, torch.full((4, 1), 0.5, dtype=F64)), R0.expand(4, 9))
    assert torch.allclose(cb.enforce(netr, torch.full((4, 1), 2.0, dtype=F64)), R1.expand(4, 9))
    with pytest.raises(ValueError):
        DirichletBVPSphericalBasis(0.5, R0, r_1=2.0)


def test_bundle_conditions_and_solver():
    """tests/test_conditions.py:165-260 and test_solvers.py (bundle) of the reference, re-stated."""
    torch.manual_seed(5)
    net = FCNN(3, 1).double()
    t0 = torch.zeros(9, 1, dtype=F64, requires_grad=True)
    u0, v0 = torch.rand(9, 1, dtype=F64), torch.rand(9, 1, dtype=F64)
    c = BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0, "u_0_prime": 1})
    u = c.enforce(net, t0, u0, v0)
    assert torch.allclose(u, u0) and torch.allclose(diff(u, t0), v0)
    c = BundleIVP(u_0=1.5, bundle_param_lookup={"t_0": 1})             # t_0 itself sampled (second bundle input)
    ts = torch.rand(9, 1, dtype=F64)
    assert torch.allclose(c.enforce(net, ts, u0, ts), torch.full_like(ts, 1.5))
    b = BundleDirichletBVP(0.0, None, 2.0, -1.0, bundle_param_lookup={"u_0": 0})
    net2 = FCNN(2, 1).double()
    assert torch.allclose(b.enforce(net2, torch.zeros_like(u0), u0), u0)
    assert torch.allclose(b.enforce(net2, torch.full_like(u0, 2.0), u0), torch.full_like(u0, -1.0))
    with pytest.raises(ValueError):
        BundleIVP(0.0, 1.0, bundle_param_lookup={"bogus": 0})
    with pytest.warns(FutureWarning):
        BundleIVP(0.0, x_0=1.0)
    mesh = Generator1D(4) ^ Generator1D(5) ^ Generator1D(6)
    assert mesh.size == 120 and len(mesh.generators) == 3 and all(x.shape == (120,) for x in mesh.get_examples())
    s = BundleSolver1D(lambda u, t, lam: [diff(u, t) + lam * u], [BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0})],
                       t_min=0.0, t_max=1.0, theta_min=(0.5, 0.5), theta_max=(2.0, 2.0), eq_param_index=(1,), n_batches_valid=1)
    assert s.generator["train"].size == 32 ** 3 and s.nets[0].NN[0].in_features == 3
    s.generator["train"] = SamplerGenerator(Generator1D(8) ^ Generator1D(4, 0.5, 2.0) ^ Generator1D(4, 0.5, 2.0))
    s.generator["valid"] = SamplerGenerator(Generator1D(8) ^ Generator1D(4, 0.5, 2.0) ^ Generator1D(4, 0.5, 2.0))
    s.fit(2, tqdm_file=None)
    assert len(s.metrics_history["train_loss"]) == 2
    sol = s.get_solution()(torch.zeros(5), torch.linspace(0.5, 2, 5), torch.ones(5))
    assert torch.allclose(sol.cpu(), torch.linspace(0.5, 2, 5), atol=1e-6)
    assert set(s.get_internals(["r_min", "eq_param_index"], return_type="dict")) == {"r_min", "eq_param_index"}
    with pytest.raises(ValueError):
        BundleSolver1D(lambda u, t: [u], [IVP(0, 1)], t_min=0, t_max=1, theta_min=(0,), theta_max=())


# ----------------------------------------------------------------------------------------------- harmonics
def test_real_spherical_harmonics_against_cartesian_forms():
    th, ph = torch.rand(40, 1, dtype=F64) * math.pi, torch.rand(40, 1, dtype=F64) * 2 * math.pi
    x, y, z = ops.spherical_to_cartesian(torch.ones_like(th), th, ph)
    Y = RealSphericalHarmonics(4)(th, ph)
    assert Y.shape == (40, 25)
    k1, k2 = math.sqrt(3) / 2, math.sqrt(15) / 2
    want = {0: 0.5 * torch.ones_like(x), 1: k1 * y, 2: k1 * z, 3: k1 * x, 4: k2 * x * y, 5: k2 * y * z,
            6: math.sqrt(5) / 4 * (2 * z ** 2 - x ** 2 - y ** 2), 7: k2 * z * x, 8: k2 / 2 * (x ** 2 - y ** 2)}
    for k, w in want.items():
        assert torch.allclose(Y[:, k:k + 1], w, atol=1e-6), k
    with pytest.raises(NotImplementedError):
        RealSphericalHarmonics(5)
    with pytest.raises(ValueError):
        RealSphericalHarmonics(2)(th.reshape(-1), ph.reshape(-1))


def test_harmonics_laplacian_matches_brute_force():
    torch.manual_seed(4)
    r = col(30, 0.5, 2.0)
    th, ph = col(30, 0.3, 2.8), col(30, 0.1, 6.0)
    net = FCNN(1, 9, hidden_units=(16,)).double()
    R = net(r)
    u = (R * RealSphericalHarmonics(2)(th, ph)).sum(dim=1, keepdim=True)
    assert torch.allclose(HarmonicsLaplacian(2)(R, r, th, ph), ops.spherical_laplacian(u, r, th, ph), atol=1e-6)


# ----------------------------------------------------------------------------------------------- generators
def test_generators_shapes_ranges_and_combinators():
    for method in ("uniform", "equally-spaced", "equally-spaced-noisy", "log-spaced", "log-spaced-noisy", "chebyshev",
                   "chebyshev2", "chebyshev2-noisy", "latin-hypercube"):
        g = Generator1D(32, 0.1, 2.0, method)
        t = g.get_examples()
        assert t.shape == (32,) and t.requires_grad and g.size == 32
    with pytest.raises(ValueError):
        Generator1D(8, method="bogus")
    with pytest.raises(ValueError):
        Generator1D(8, -1.0, 1.0, "log-spaced")
    for method in ("equally-spaced", "equally-spaced-noisy", "chebyshev", "chebyshev2", "chebyshev2-noisy", "latin-hypercube"):
        x, y = Generator2D((4, 5), (0, 0), (1, 2), method).get_examples()
        assert x.shape == y.shape == (20,)
    x, y, z = Generator3D((2, 3, 4)).get_examples()
    assert x.shape == (24,)
    r, th, ph = GeneratorSpherical(64, 0.5, 1.5).get_examples()
    assert (r >= 0.5).all() and (r <= 1.5).all() and (th >= 0).all() and (th <= math.pi).all() and (ph >= 0).all() and (ph <= 2 * math.pi).all()
    with pytest.raises(ValueError):
        GeneratorSpherical(8, 1.0, 0.5)
    cat = Generator1D(8) + Generator1D(4)
    assert isinstance(cat, ConcatGenerator) and cat.size == 12 and cat.get_examples().shape == (12,)
    ens = Generator1D(8) * Generator2D((2, 4))
    assert isinstance(ens, EnsembleGenerator) and len(ens.get_examples()) == 3
    with pytest.raises(ValueError):
        EnsembleGenerator(Generator1D(8), Generator1D(9))
    st = StaticGenerator(Generator1D(8, method="uniform"))
    assert torch.equal(st.get_examples(), st.get_examples())
    pre = PredefinedGenerator([0.0, 1.0, 2.0], [3.0, 4.0, 5.0])
    assert pre.size == 3 and len(pre.get_examples()) == 2
    cols = SamplerGenerator(Generator2D((3, 3))).get_examples()
    assert all(c.shape == (9, 1) and c.requires_grad for c in cols)


# ----------------------------------------------------------------------------------------------- networks / losses
def test_networks_and_losses():
    assert FCNN(2, 3, hidden_units=(8, 9, 10))(torch.rand(7, 2)).shape == (7, 3)
    with pytest.warns(FutureWarning):
        net = FCNN(1, 1, n_hidden_units=16, n_hidden_layers=2)
    assert [m.out_features for m in net.NN if isinstance(m, nn.Linear)] == [16, 16, 16, 1]
    assert len(list(FCNN().NN)) == 5 and isinstance(FCNN().NN[1], nn.Tanh)
    x = torch.linspace(-2, 2, 9)
    assert torch.allclose(SinActv()(x), torch.sin(x)) and torch.allclose(Swish(2.0)(x), x * torch.sigmoid(2 * x))
    assert torch.allclose(APTx(1.0, 1.0, 0.5)(x), (1 + torch.tanh(x)) * 0.5 * x)
    assert len(list(Swish(trainable=True).parameters())) == 1 and len(list(APTx(trainable=True).parameters())) == 3
    torch.manual_seed(0)
    rn = Resnet(2, 3, hidden_units=(8, 8))
    xin = torch.rand(5, 2)
    assert torch.allclose(rn(xin), rn.skip_connection(xin) + rn.residual(xin)) and rn.skip_connection.bias is None
    assert torch.equal(MonomialNN(3)(xin), torch.cat([xin, xin ** 2, xin ** 3], dim=1)) and MonomialNN([2, 4]).degrees == (2, 4)
    with pytest.raises(ValueError):
        MonomialNN([])
    xs = [col(10) for _ in range(2)]
    res = torch.cat([xs[0] * xs[1], xs[0] ** 2], dim=1)
    for name, fn in _losses.items():
        val = fn(res, None, xs)
        assert val.shape == () and val.requires_grad, name


# ----------------------------------------------------------------------------------------------- solvers (composite path)
def test_solver_plumbing_on_cpu():
    torch.manual_seed(0)
    ode = lambda u, t: [diff(u, t) + u]
    with pytest.raises(ValueError):
        Solver1D(ode, [IVP(0.0, 1.0)])
    hits = []
    s = Solver1D(ode, [IVP(0.0, 1.0)], t_min=0.0, t_max=1.0, n_batches_valid=1,
                 metrics={"mse": lambda u, t: ((u - torch.exp(-t)) ** 2).mean()})
    s.fit(3, callbacks=[lambda solver: hits.append(solver.global_epoch)], tqdm_file=None)
    h = s.metrics_history
    assert hits == [1, 2, 3] and all(len(h[k]) == 3 for k in ("train_loss", "valid_loss", "train__mse", "valid__mse"))
    assert s.lowest_loss == min(h["valid_loss"]) and s.best_nets is not None and s.global_epoch == 3
    sol = s.get_solution()
    ts = np.linspace(0, 1, 7, dtype=np.float32)
    assert sol(ts, to_numpy=True).shape == (7,) and isinstance(sol(torch.tensor(ts)), torch.Tensor)
    assert s.get_residuals(ts, to_numpy=True).shape == (7,)
    assert set(s.get_internals(["nets", "t_min"], return_type="dict")) == {"nets", "t_min"}
    with pytest.raises(TypeError):
        Solver1D(ode, [IVP(0.0, 1.0)], t_min=0, t_max=1, loss_fn=3)
    s2 = Solver2D(lambda u, x, y: [ops.laplacian(u, x, y)], [NoCondition()], xy_min=(0, 0), xy_max=(1, 1), loss_fn="l1",
                  optimizer=None, n_batches_valid=0)
    s2.fit(2, tqdm_file=None)
    assert len(s2.metrics_history["train_loss"]) == 2 and s2.metrics_history["valid_loss"] == []
    s3 = SolverSpherical(lambda u, r, th, ph: [ops.spherical_laplacian(u, r, th, ph)],
                         [DirichletBVPSpherical(0.5, lambda th, ph: 0 * th, 1.0, lambda th, ph: 0 * th + 1)], 0.5, 1.0,
                         n_batches_valid=0)
    s3.fit(1, tqdm_file=None)
    assert s3.get_solution()(torch.rand(4), torch.rand(4), torch.rand(4)).shape == (4,)
    lb = Solver1D(ode, [IVP(0.0, 1.0)], t_min=0.0, t_max=1.0, n_batches_valid=1)
    lb.optimizer = torch.optim.LBFGS(lb.nets[0].parameters(), max_iter=2)
    lb.fit(1, tqdm_file=None)                                # closure-based optimiser steps per batch
    assert len(lb.metrics_history["train_loss"]) == 1


def test_describe_recognises_the_network_family():
    """networks.describe(): what the gfx950 kernels are asked to run for each member of the reference's network family
    (networks.py:26-209) -- widths, skip, activation parameters, monomial front end -- and what they turn down."""
    from functools import partial
    import torch.nn as nn
    from neurodiffeq_amd.networks import FCNN, Resnet, MonomialNN, Swish, APTx, SinActv, describe, FlatParams
    d = describe(FCNN(2, 1))
    assert (d["d"], d["hidden"], d["layers"], d["n_out"], d["skip"], d["actp"], d["widths"], d["mono"]) == (2, 32, 2, 1, 0, 0, 0, 0)
    d = describe(FCNN(2, 3, hidden_units=(50, 50, 50), actv=SinActv))
    assert (d["hidden"], d["layers"], d["n_out"], d["widths"], d["act"]) == (50, 3, 3, 0, 1)
    d = describe(FCNN(1, 1, hidden_units=(64, 32, 16)))
    assert d["hidden"] == 64 and d["widths"] == 64 | (32 << 8) | (16 << 16)
    d = describe(Resnet(2, 5, hidden_units=(32, 32)))
    assert d["skip"] == 1 and d["n_out"] == 5 and d["params"][-1].shape == (5, 2)
    net = FCNN(2, 1, actv=partial(Swish, trainable=True))
    d = describe(net)
    assert d["actp"] == 1 and len(d["params"]) == 6 + 2 and all(p.dim() == 0 for p in d["params"][-2:])
    assert [id(p) for p in d["params"][-2:]] == [id(net.NN[1].beta), id(net.NN[3].beta)]       # behind the linear layers
    d = describe(FCNN(2, 1, actv=partial(APTx, alpha=0.8, beta=1.3, gamma=0.6)))
    assert d["actp"] == 2 and d["frozen"] == [0.8, 1.3, 0.6] * 2 and len(d["params"]) == 6
    assert describe(FCNN(2, 1, actv=APTx))["actp"] == 0
    d = describe(nn.Sequential(MonomialNN(3), FCNN(6, 1)))
    assert (d["d"], d["mono"], d["hidden"]) == (2, 0b111, 32)
    d = describe(nn.Sequential(MonomialNN([1, 3]), nn.Linear(2, 16), nn.Tanh(), nn.Linear(16, 2)))
    assert (d["d"], d["mono"], d["hidden"], d["layers"], d["n_out"]) == (1, 0b101, 16, 1, 2)
    # turned down: unsorted degrees, a feature count that is no multiple of the degree count, nine hidden layers (five
    # of DIFFERENT widths), mixed activations, a bias-free layer, fp64 parameters under an fp32 description
    assert describe(nn.Sequential(MonomialNN([2, 1]), FCNN(2, 1))) is None
    assert describe(nn.Sequential(MonomialNN(3), FCNN(5, 1))) is None
    d = describe(FCNN(1, 1, hidden_units=(16,) * 5))           # since round 3: up to 8 hidden layers of one width
    assert (d["hidden"], d["layers"], d["widths"]) == (16, 5, 0)
    assert describe(FCNN(1, 1, hidden_units=(16,) * 9)) is None
    assert describe(FCNN(1, 1, hidden_units=(16, 32, 16, 32, 16))) is None
    assert describe(nn.Sequential(nn.Linear(1, 16), nn.Tanh(), nn.Linear(16, 16), nn.Sigmoid(), nn.Linear(16, 1))) is None
    assert describe(nn.Sequential(nn.Linear(1, 16, bias=False), nn.Tanh(), nn.Linear(16, 1))) is None
    assert describe(FCNN(1, 1).double()) is None and describe(FCNN(1, 1).double(), dtype=torch.float64) is not None
    # FlatParams: parameters become views of ONE flat buffer in the kernels' order, fixed activation scalars behind them
    net = FCNN(2, 1, actv=partial(Swish, beta=1.7))
    fp = FlatParams(net, "cpu")
    assert fp.numel == sum(p.numel() for p in net.parameters()) and fp.flat.numel() == fp.numel + 2
    assert fp.flat[-2:].tolist() == [1.7000000476837158, 1.7000000476837158]
    assert all(p.data_ptr() == fp.flat.data_ptr() + 4 * off for p, off in zip(fp.params, fp._offsets))


def test_library_code_suspends_global_torch_modes_and_user_code_restores_them():
    """engine.library_code: package host code runs without torch's global TorchFunctionMode (a default device installs one:
    ~1 us per tensor method call); user callables inside such a block -- further batches of a multi-batch epoch -- run with
    the modes back in force (``user_code``), i.e. with the default device the user set."""
    import torch
    from neurodiffeq_amd.engine import library_code
    assert torch.empty(1).device.type == "cpu"
    with library_code() as lc:                       # nothing installed: a no-op
        assert lc.ctx is None
        with lc.user_code():
            assert torch.empty(1).device.type == "cpu"
    torch.set_default_device("meta")
    try:
        assert torch.empty(1).device.type == "meta"
        with library_code() as lc:
            assert torch.empty(1).device.type == "cpu"        # the global mode is off for package code
            with lc.user_code():
                assert torch.empty(1).device.type == "meta"   # ... and back for the user's generator
            assert torch.empty(1).device.type == "cpu"
        assert torch.empty(1).device.type == "meta"
    finally:
        torch.set_default_device(None)
    assert torch.empty(1).device.type == "cpu"
