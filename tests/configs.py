"""The BASELINE.json configs written against the neurodiffeq_amd API -- exactly what a user script
for the reference looks like with the import changed.  Shared by the tests, bench.py and __graft_entry__.py."""
import math

import torch

from neurodiffeq_amd import diff
from functools import partial
from neurodiffeq_amd.conditions import (IVP, DirichletBVP2D, IBVP1D, NoCondition, DirichletBVPSphericalBasis, BundleIVP,
                                        DirichletBVPSpherical, DoubleEndedBVP1D, EnsembleCondition)
from neurodiffeq_amd.function_basis import RealSphericalHarmonics
from neurodiffeq_amd.generators import Generator1D, Generator2D, GeneratorSpherical
from neurodiffeq_amd.operators import spherical_laplacian
from neurodiffeq_amd.networks import FCNN, SinActv, Swish, APTx, Resnet, MonomialNN

PI = math.pi
DEFAULT_SIZE = {"c1": 1024, "c2": 256, "c3": 512, "c5": 1024, "c4": 131072}


def lid(x):
    return (1 - torch.exp(-50.0 * x)) * (1 - torch.exp(50.0 * (x - 1)))


def make(name, size=None):
    """Returns dict(kind, pde, nets, conds, gen, n_points); nets use torch's default init (consumes the global RNG
    like the reference's constructors)."""
    size = size or DEFAULT_SIZE.get(name)
    zero = lambda s: 0
    if name == "c1":
        pde = lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)]
        nets = [FCNN(1, 1, hidden_units=(32, 32), actv=SinActv) for _ in range(2)]
        conds = [IVP(0.0, 1.5), IVP(0.0, 1.0)]
        gen = Generator1D(size, 0.1, 12.0, "equally-spaced-noisy")
        return dict(kind="1d", pde=pde, nets=nets, conds=conds, gen=gen, n_points=size, dom=(0.1, 12.0))
    if name == "c2":
        pde = lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)]
        nets = [FCNN(2, 1, hidden_units=(32, 32))]
        conds = [DirichletBVP2D(x_min=0, x_min_val=lambda y: torch.sin(PI * y), x_max=1, x_max_val=zero,
                                y_min=0, y_min_val=zero, y_max=1, y_max_val=zero)]
        gen = Generator2D((size, size), (0, 0), (1, 1), "equally-spaced-noisy")
        return dict(kind="2d", pde=pde, nets=nets, conds=conds, gen=gen, n_points=size * size, dom=((0, 0), (1, 1)))
    if name == "c3":
        nu = 0.01 / PI
        pde = lambda u, x, t: [diff(u, t) + u * diff(u, x) - nu * diff(u, x, order=2)]
        nets = [FCNN(2, 1, hidden_units=(64, 64, 64))]
        conds = [IBVP1D(x_min=-1, x_max=1, t_min=0, t_min_val=lambda x: -torch.sin(PI * x), x_min_val=zero,
                        x_max_val=zero)]
        gen = Generator2D((size, size), (-1, 0), (1, 1), "equally-spaced-noisy")
        return dict(kind="2d", pde=pde, nets=nets, conds=conds, gen=gen, n_points=size * size, dom=((-1, 0), (1, 1)))
    if name == "c5":
        re = 400.0

        def pde(u, v, p, x, y):
            mx = u * diff(u, x) + v * diff(u, y) + diff(p, x) - 1 / re * (diff(u, x, order=2) + diff(u, y, order=2))
            my = u * diff(v, x) + v * diff(v, y) + diff(p, y) - 1 / re * (diff(v, x, order=2) + diff(v, y, order=2))
            return [mx, my, diff(u, x) + diff(v, y)]

        nets = [FCNN(2, 1, hidden_units=(64, 64, 64)) for _ in range(3)]
        conds = [DirichletBVP2D(0, zero, 1, zero, 0, zero, 1, lid), DirichletBVP2D(0, zero, 1, zero, 0, zero, 1, zero),
                 NoCondition()]
        gen = Generator2D((size, size), (0, 0), (1, 1), "equally-spaced-noisy")
        return dict(kind="2d", pde=pde, nets=nets, conds=conds, gen=gen, n_points=size * size, dom=((0, 0), (1, 1)))
    if name == "c4":      # Poisson with a Gaussian charge density in a spherical shell, harmonic expansion l <= 4
        r0, r1 = 0.1, 3.0
        gauss = 1.0 / (2 * PI) ** 1.5
        kq = 1.0 / (4 * PI)
        v0 = kq / r0 * math.erf(r0 / math.sqrt(2.0))
        v1 = kq / r1 * math.erf(r1 / math.sqrt(2.0))
        R0 = torch.zeros(25, dtype=torch.float32); R0[0] = 2 * v0
        R1 = torch.zeros(25, dtype=torch.float32); R1[0] = 2 * v1
        Y = RealSphericalHarmonics(max_degree=4)
        pde = lambda u, r, th, ph: [spherical_laplacian(u, r, th, ph) + gauss * torch.exp(-r ** 2 / 2)]
        nets = [FCNN(1, 25, hidden_units=(32, 32))]
        conds = [DirichletBVPSphericalBasis(r_0=r0, R_0=R0, r_1=r1, R_1=R1)]
        gen = GeneratorSpherical(size, r0, r1)
        enforcer = lambda net, cond, coords: (cond.enforce(net, coords[0]) * Y(*coords[1:])).sum(dim=1, keepdim=True)
        return dict(kind="sph", pde=pde, nets=nets, conds=conds, gen=gen, n_points=size, dom=(r0, r1), enforcer=enforcer)
    # ---- rows widened into after the BASELINE configs; fixed small sizes, goldens in tests/golden/w*.npz
    if name == "w1":      # BundleSolver1D: u' + lam u = 0, u(0) = u0, bundle inputs (u0, lam)
        ode = lambda u, t, lam: [diff(u, t) + lam * u]
        nets = [FCNN(3, 1, hidden_units=(32, 32))]
        conds = [BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0})]
        gen = Generator1D(8, 0.0, 1.0, "equally-spaced-noisy") ^ Generator1D(4, 0.5, 2.0, "equally-spaced-noisy") \
            ^ Generator1D(4, 0.5, 2.0, "equally-spaced-noisy")
        return dict(kind="bundle", ode=ode, pde=lambda u, t, u0, lam: ode(u, t, lam), nets=nets, conds=conds, gen=gen,
                    n_points=128, dom=(0.0, 1.0), theta=((0.5, 0.5), (2.0, 2.0)), eq_param_index=(1,))
    if name == "w2":      # SolverSpherical with its default FCNN(3, 1) and DirichletBVPSpherical
        pde = lambda u, r, th, ph: [spherical_laplacian(u, r, th, ph)]
        nets = [FCNN(3, 1, hidden_units=(32, 32))]
        conds = [DirichletBVPSpherical(0.5, lambda th, ph: torch.cos(th), 2.0, lambda th, ph: 0.25 * torch.cos(th))]
        return dict(kind="sph", pde=pde, nets=nets, conds=conds, gen=GeneratorSpherical(96, 0.5, 2.0), n_points=96,
                    dom=(0.5, 2.0))
    if name == "w3":      # Swish network on the C2 problem
        c = make("c2", 12)
        c["nets"] = [FCNN(2, 1, hidden_units=(32, 32), actv=Swish)]
        return c
    if name in ("w6", "w7"):      # IBVP1D with Neumann ends: the network is evaluated on the boundary as well
        if name == "w6":
            pde = lambda u, x, t: [diff(u, t) - 0.1 * diff(u, x, order=2)]
            cond = IBVP1D(x_min=0.0, x_max=1.0, t_min=0.0, t_min_val=lambda x: torch.cos(PI * x),
                          x_min_prime=lambda t: 0.0 * t, x_max_prime=lambda t: 0.2 * t)
        else:
            pde = lambda u, x, t: [diff(u, t) - 0.1 * diff(u, x, order=2) + u ** 2]
            cond = IBVP1D(x_min=0.0, x_max=1.0, t_min=0.0, t_min_val=lambda x: torch.sin(PI * x / 2),
                          x_min_val=lambda t: 0.0 * t, x_max_prime=lambda t: torch.sin(t))
        return dict(kind="2d", pde=pde, nets=[FCNN(2, 1, hidden_units=(32, 32))], conds=[cond],
                    gen=Generator2D((10, 10), (0, 0), (1, 1), "equally-spaced-noisy"), n_points=100, dom=((0, 0), (1, 1)))
    if name == "w8":      # DoubleEndedBVP1D: Neumann-Dirichlet and Dirichlet-Neumann, one network each
        pde = lambda u, v, x: [diff(u, x, order=2) + u - v, diff(v, x, order=2) - v + torch.sin(x)]
        nets = [FCNN(1, 1, hidden_units=(32, 32)) for _ in range(2)]
        conds = [DoubleEndedBVP1D(0.0, 1.0, x_min_prime=-0.5, x_max_val=2.0),
                 DoubleEndedBVP1D(0.0, 1.0, x_min_val=1.0, x_max_prime=0.5)]
        return dict(kind="1d", pde=pde, nets=nets, conds=conds, gen=Generator1D(48, 0.0, 1.0, "equally-spaced-noisy"),
                    n_points=48, dom=(0.0, 1.0))
    if name in ("w11", "w12", "w13"):      # network family on the C2 problem (goldens w11 - w13)
        c = make("c2", 12)
        c["nets"] = [{"w11": lambda: FCNN(2, 1, hidden_units=(32, 32), actv=partial(Swish, beta=1.25, trainable=True)),
                      "w12": lambda: Resnet(2, 1, hidden_units=(50, 30)),
                      "w13": lambda: torch.nn.Sequential(MonomialNN(3), FCNN(6, 1, hidden_units=(32, 32)))}[name]()]
        return c
    if name == "w14":     # Lotka-Volterra on ONE two-output network under EnsembleCondition: a single two-column function
        def ode(uv, t):
            u, v = uv[:, 0:1], uv[:, 1:2]
            return [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)]
        return dict(kind="1d", pde=ode, nets=[FCNN(1, 2, hidden_units=(32, 32), actv=SinActv)],
                    conds=[EnsembleCondition(IVP(0.0, 1.5), IVP(0.0, 1.0))],
                    gen=Generator1D(64, 0.1, 4.0, "equally-spaced-noisy"), n_points=64, dom=(0.1, 4.0))
    if name == "w15":     # APTx with trainable alpha, beta, gamma on a second-order ODE
        pde = lambda u, t: [diff(u, t, order=2) + 0.5 * diff(u, t) + u - torch.cos(t)]
        return dict(kind="1d", pde=pde, nets=[FCNN(1, 1, hidden_units=(32, 32), actv=partial(APTx, trainable=True))],
                    conds=[IVP(0.0, 1.0, u_0_prime=0.5)], gen=Generator1D(48, 0.0, 2.0, "equally-spaced-noisy"),
                    n_points=48, dom=(0.0, 2.0))
    if name == "w20":     # tests/test_pde.py:370-377: (100, 100) ELU network on a nonlinear Poisson problem
        c = make("c2", size or 12)
        c["pde"] = lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2) + torch.exp(u) - 1.0 - x ** 2 - y ** 2
                                    - 4.0 / (1.0 + x ** 2 + y ** 2) ** 2]
        c["nets"] = [FCNN(n_input_units=2, hidden_units=(100, 100), actv=torch.nn.ELU)]
        return c
    if name == "w21":     # Softplus and GELU networks in one system
        nets = [FCNN(2, 1, hidden_units=(32, 32), actv=torch.nn.Softplus), FCNN(2, 1, hidden_units=(32, 32), actv=torch.nn.GELU)]
        c = make("c2", size or 10)          # (after the networks: the golden script draws its initial weights in this order)
        c["pde"] = lambda u, v, x, y: [diff(u, x, order=2) + diff(u, y, order=2) - v, diff(v, x) + diff(v, y) - u * v]
        c["nets"] = nets
        c["conds"] = c["conds"] + [NoCondition()]
        return c
    if name == "w16":     # README.md:125 -- FCNN(2, 1, hidden_units=(512,)) on the C2 problem
        c = make("c2", size or 12)
        c["nets"] = [FCNN(n_input_units=2, n_output_units=1, hidden_units=(512,))]
        return c
    if name == "w18":     # 128 x 3 on the Burgers problem of C3
        c = make("c3", size or 12)
        c["nets"] = [FCNN(2, 1, hidden_units=(128, 128, 128))]
        return c
    if name == "w26":     # per-layer widths above 64 units: (128, 64) on the Burgers problem of C3
        c = make("c3", size or 12)
        c["nets"] = [FCNN(2, 1, hidden_units=(128, 64))]
        return c
    if name == "w27":     # (96, 200, 40) with nn.Sigmoid on the C2 problem (sigma(0) != 0 in the padding units)
        c = make("c2", size or 12)
        c["nets"] = [FCNN(2, 1, hidden_units=(96, 200, 40), actv=torch.nn.Sigmoid)]
        return c
    if name == "w28":     # Resnet (128, 64) on the C2 problem: symbolic skip connection + per-layer widths
        c = make("c2", size or 12)
        c["nets"] = [Resnet(2, 1, hidden_units=(128, 64))]
        return c
    if name == "w24":     # Resnet 128 x 2 on the C2 problem: skip connection above 64 units (handled by the tracer)
        c = make("c2", size or 12)
        c["nets"] = [Resnet(2, 1, hidden_units=(128, 128))]
        return c
    if name == "w25":     # Resnet 2 -> 512 -> 3 on the single-network cavity problem
        net = Resnet(n_input_units=2, n_output_units=3, hidden_units=(512,))     # (first: the golden script's RNG order)
        c = make("w17", size)
        c["nets"] = [net]
        return c
    if name == "w18r":    # w18 on a ragged batch: 251 x 261 = 65 511 points (tests/golden/make_golden.py: cfg_w18r)
        c = make("w18", 12)
        g = size or (251, 261)
        c["gen"] = Generator2D(tuple(g), (-1, 0), (1, 1), "equally-spaced-noisy")
        c["n_points"] = g[0] * g[1]
        return c
    if name in ("w17", "w19"):     # lid-driven cavity on ONE three-output network (EnsembleCondition): 2 -> 512 -> 3 and the
        #                            RE100 notebook's FCNN(n_hidden_units=256, n_hidden_layers=1) = 2 -> 256 -> 256 -> 3
        re = 400.0 if name == "w17" else 100.0

        def pde(uvp, x, y):
            u, v, p = uvp[:, 0:1], uvp[:, 1:2], uvp[:, 2:3]
            mx = u * diff(u, x) + v * diff(u, y) + diff(p, x) - 1 / re * (diff(u, x, order=2) + diff(u, y, order=2))
            my = u * diff(v, x) + v * diff(v, y) + diff(p, y) - 1 / re * (diff(v, x, order=2) + diff(v, y, order=2))
            return [mx, my, diff(u, x) + diff(v, y)]
        if name == "w17":
            net = FCNN(n_input_units=2, n_output_units=3, hidden_units=(512,))
        else:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", FutureWarning)
                net = FCNN(n_input_units=2, n_hidden_units=256, n_hidden_layers=1, n_output_units=3, actv=torch.nn.Tanh)
        conds = [EnsembleCondition(DirichletBVP2D(0, zero, 1, zero, 0, zero, 1, lid),
                                   DirichletBVP2D(0, zero, 1, zero, 0, zero, 1, zero), NoCondition())]
        g = size or 8
        return dict(kind="2d", pde=pde, nets=[net], conds=conds, gen=Generator2D((g, g), (0, 0), (1, 1), "equally-spaced-noisy"),
                    n_points=g * g, dom=((0, 0), (1, 1)))
    if name == "w5":      # Resnet on the C2 problem
        c = make("c2", 12)
        c["nets"] = [Resnet(2, 1, hidden_units=(32, 32))]
        return c
    if name == "w9":      # Sobolev loss on a second-order PDE: third-order network streams (losses.py:17-26)
        c = make("c2", 10)
        c["pde"] = lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2) + u * diff(u, x) - torch.sin(PI * x)]
        c["loss"] = "h1"
        return c
    if name == "w10":     # third-order ODE, sin network
        pde = lambda u, t: [diff(u, t, order=3) + diff(u, t, order=2) * diff(u, t) + u - torch.sin(t)]
        return dict(kind="1d", pde=pde, nets=[FCNN(1, 1, hidden_units=(32, 32), actv=SinActv)], conds=[IVP(0.0, 1.0)],
                    gen=Generator1D(48, 0.0, 2.0, "equally-spaced-noisy"), n_points=48, dom=(0.0, 2.0))
    if name == "w29":     # fourth-order ODE (beam on an elastic foundation), round 6
        pde = lambda u, t: [diff(u, t, order=4) + u - torch.cos(t)]
        return dict(kind="1d", pde=pde, nets=[FCNN(1, 1, hidden_units=(32, 32))], conds=[IVP(0.0, 1.0, u_0_prime=0.5)],
                    gen=Generator1D(48, 0.0, 2.0, "equally-spaced-noisy"), n_points=48, dom=(0.0, 2.0))
    if name == "w30":     # biharmonic equation on the C2 domain: pure and mixed fourth derivatives
        c = make("c2", 10)
        c["pde"] = lambda u, x, y: [diff(u, x, order=4) + 2.0 * diff(diff(u, x, order=2), y, order=2) + diff(u, y, order=4)
                                    - torch.sin(PI * x) * torch.sin(PI * y)]
        return c
    if name == "w4":      # APTx networks: second-order ODE with a Neumann-form IVP coupled to a first-order one
        pde = lambda u, v, t: [diff(u, t, order=2) + v * diff(u, t) + u, diff(v, t) - u * v + torch.sin(t)]
        nets = [FCNN(1, 1, hidden_units=(32, 32), actv=APTx) for _ in range(2)]
        conds = [IVP(0.0, 1.0, u_0_prime=0.0), IVP(0.0, 0.5)]
        return dict(kind="1d", pde=pde, nets=nets, conds=conds, gen=Generator1D(64, 0.0, 2.0, "equally-spaced-noisy"),
                    n_points=64, dom=(0.0, 2.0))
    raise KeyError(name)


def make_inverse(name, size=None):
    """Inverse problems (tests/golden/make_golden.py: cfg_inv1 / cfg_inv2): trainable scalars inside the equations, a
    per-point data column.  Same construction order as the golden script (network init, then the data draw)."""
    from neurodiffeq_amd.conditions import DirichletBVP
    from neurodiffeq_amd.generators import PredefinedGenerator
    if name == "inv1":
        g = size or 12
        nets = [FCNN(2, 1, hidden_units=(32, 32))]
        nu, amp = torch.nn.Parameter(torch.tensor(0.05)), torch.nn.Parameter(torch.tensor(0.8))
        pde = lambda u, x, t: [diff(u, t) + amp * u * diff(u, x) - nu * diff(u, x, order=2)]
        conds = [IBVP1D(x_min=-1, x_max=1, t_min=0, t_min_val=lambda x: -torch.sin(PI * x), x_min_val=lambda t: 0,
                        x_max_val=lambda t: 0)]
        gen = Generator2D((g, g), (-1, 0), (1, 1), "equally-spaced-noisy")
        return dict(kind="2d", pde=pde, nets=nets, conds=conds, gen=gen, theta=[nu, amp], data=[], n_points=g * g,
                    dom=((-1, 0), (1, 1)))
    if name == "inv2":
        n = size or 48
        nets = [FCNN(1, 1, hidden_units=(32, 32), actv=SinActv)]
        xs = torch.linspace(0.0, 1.0, n)
        f = (-(PI ** 2) * torch.sin(PI * xs) + 0.1 * torch.rand(n)).reshape(-1, 1)
        a, b = torch.nn.Parameter(torch.tensor(0.6)), torch.nn.Parameter(torch.tensor(0.1))
        pde = lambda u, x: [diff(u, x, order=2) - a * f - b]
        conds = [DirichletBVP(0.0, 0.0, 1.0, 0.0)]
        return dict(kind="1d", pde=pde, nets=nets, conds=conds, gen=PredefinedGenerator(xs), theta=[a, b], data=[f],
                    n_points=n, dom=(0.0, 1.0))
    raise KeyError(name)


def make_inverse_solver(name, size=None, **kw):
    """The solver of an inverse problem: the coefficients sit in the optimiser next to the network's parameters (the
    reference's default optimiser knows the networks only, solvers.py:182)."""
    from neurodiffeq_amd.solvers import Solver1D, Solver2D
    cfg = make_inverse(name, size)
    opt = torch.optim.Adam([p for n in cfg["nets"] for p in n.parameters()] + cfg["theta"], lr=1e-3)
    kw.setdefault("n_batches_valid", 0)
    if cfg["kind"] == "2d":
        s = Solver2D(cfg["pde"], cfg["conds"], xy_min=cfg["dom"][0], xy_max=cfg["dom"][1], nets=cfg["nets"],
                     train_generator=cfg["gen"], valid_generator=cfg["gen"], optimizer=opt, **kw)
    else:
        s = Solver1D(cfg["pde"], cfg["conds"], t_min=cfg["dom"][0], t_max=cfg["dom"][1], nets=cfg["nets"],
                     train_generator=cfg["gen"], valid_generator=cfg["gen"], optimizer=opt, **kw)
    return s, cfg


def n_coords(cfg):
    return {"1d": 1, "2d": 2, "sph": 3, "bundle": 3}[cfg["kind"]]


def func_val(cfg):
    """compute_func_val of the config's solver class (SolverSpherical routes through the enforcer)."""
    if "enforcer" in cfg:
        return lambda net, cond, *coords: cfg["enforcer"](net, cond, coords)
    return None


def make_solver(name, size=None, **kw):
    from neurodiffeq_amd.solvers import Solver1D, Solver2D, SolverSpherical, BundleSolver1D
    cfg = make(name, size)
    kw.setdefault("n_batches_valid", 0)
    if cfg.get("loss"):
        kw.setdefault("loss_fn", cfg["loss"])
    if cfg["kind"] == "sph":
        s = SolverSpherical(cfg["pde"], cfg["conds"], r_min=cfg["dom"][0], r_max=cfg["dom"][1], nets=cfg["nets"],
                            train_generator=cfg["gen"], valid_generator=cfg["gen"], enforcer=cfg.get("enforcer"), **kw)
    elif cfg["kind"] == "bundle":
        s = BundleSolver1D(cfg["ode"], cfg["conds"], t_min=cfg["dom"][0], t_max=cfg["dom"][1], theta_min=cfg["theta"][0],
                           theta_max=cfg["theta"][1], eq_param_index=cfg["eq_param_index"], nets=cfg["nets"],
                           train_generator=cfg["gen"], valid_generator=cfg["gen"], **kw)
    elif cfg["kind"] == "1d":
        s = Solver1D(cfg["pde"], cfg["conds"], t_min=cfg["dom"][0], t_max=cfg["dom"][1], nets=cfg["nets"],
                     train_generator=cfg["gen"], valid_generator=cfg["gen"], **kw)
    else:
        s = Solver2D(cfg["pde"], cfg["conds"], xy_min=cfg["dom"][0], xy_max=cfg["dom"][1], nets=cfg["nets"],
                     train_generator=cfg["gen"], valid_generator=cfg["gen"], **kw)
    return s, cfg


# ---- a solver with a user loss_fn, an additional_loss override and metrics (all traced onto the fused path)
def custom_loss_fn(r, f, x):
    return ((1.0 + x[0] ** 2) * r ** 2).mean() + 0.1 * (f[0] ** 2).mean()


CUSTOM_METRICS = {"energy": lambda u, x, y: (diff(u, x) ** 2 + diff(u, y) ** 2).mean(),
                  "mean_u": lambda u, x, y: u.mean()}


def make_custom_loss_solver(size=16, **kw):
    from neurodiffeq_amd.solvers import Solver2D

    class PenalisedSolver(Solver2D):
        def additional_loss(self, residual, funcs, coords):          # solvers.py:587-604
            return 0.5 * (diff(funcs[0], coords[0]) ** 2).mean()

    cfg = make("c2", size)
    kw.setdefault("n_batches_valid", 0)
    solver = PenalisedSolver(cfg["pde"], cfg["conds"], xy_min=cfg["dom"][0], xy_max=cfg["dom"][1], nets=cfg["nets"],
                             train_generator=cfg["gen"], valid_generator=cfg["gen"], loss_fn=custom_loss_fn,
                             metrics=dict(CUSTOM_METRICS), **kw)
    return solver, cfg


def fused_equations(cfg):
    """The residual list the fused path traces for a config: the PDE itself, or -- Sobolev losses -- the PDE extended by
    the gradient of the summed residual (solvers.sobolev_equations; its l2 loss is the h1 loss)."""
    if cfg.get("loss") in ("h1", "h1 semi"):
        from neurodiffeq_amd.solvers import sobolev_equations
        return sobolev_equations(cfg["pde"], len(cfg["nets"]), semi=(cfg["loss"] == "h1 semi"))
    return cfg["pde"]
