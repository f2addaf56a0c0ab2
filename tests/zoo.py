"""A zoo of small ODE / PDE systems beyond the five BASELINE configs, used to exercise the tracer, the symbolic
differentiation, the generated pointwise code and every stream-set kernel (first order only, mixed second derivatives,
three coordinates, multi-network, sin activations ...).  Each system is stated twice from the same maths:

* ``product``: against the neurodiffeq_amd API (conditions classes + ``diff``), i.e. what a user writes;
* ``oracle``:  plain closures over ``oracle.autograd_ref.ref_diff`` with hand-written re-parameterisations
  (conditions.py of the reference: IVP 247-267, DirichletBVP 373-394, DirichletBVP2D 473-509, IBVP1D 669-681,
  DirichletBVPSpherical 920-958), so the checker shares no code with the thing checked.
"""
import math

import torch

PI = math.pi


class System:
    def __init__(self, name, n_coords, nets, box, pde, conds, enforcers):
        self.name, self.n_coords, self.net_specs, self.box = name, n_coords, nets, box
        self.pde, self.conds, self.enforcers = pde, conds, enforcers

    def sample(self, n, seed=0):
        g = torch.Generator().manual_seed(seed)
        return [lo + (hi - lo) * torch.rand(n, generator=g, dtype=torch.float64) for lo, hi in self.box]

    def product(self):
        """(nets fp32, conditions, diff_eqs) on the neurodiffeq_amd API"""
        from neurodiffeq_amd import diff
        from neurodiffeq_amd.networks import FCNN, SinActv, Swish, APTx, Resnet
        from functools import partial
        actv = {"tanh": torch.nn.Tanh, "sin": SinActv, "sigmoid": torch.nn.Sigmoid, "swish": Swish, "aptx": APTx,
                "swish-tr": partial(Swish, trainable=True), "aptx-tr": partial(APTx, trainable=True),
                "swish-fixed": partial(Swish, beta=1.7), "aptx-fixed": partial(APTx, alpha=0.8, beta=1.3, gamma=0.6)}
        from neurodiffeq_amd.networks import MonomialNN

        def make(i, o, h, a):
            if a.startswith("mono"):            # "mono3-tanh": MonomialNN(3) in front of an FCNN with 3 i inputs
                k = int(a[4])
                return torch.nn.Sequential(MonomialNN(k), FCNN(i * k, o, hidden_units=h, actv=actv[a[6:]]))
            return (Resnet if a.startswith("resnet-") else FCNN)(i, o, hidden_units=h, actv=actv[a.replace("resnet-", "")])
        nets = [make(*spec) for spec in self.net_specs]
        # trainable activation parameters: move them off their defaults (every layer its own values), deterministically
        k = 0
        for net in nets:
            for m in net.modules():
                if isinstance(m, (Swish, APTx)) and m.trainable:
                    for p in m.parameters():
                        p.data.mul_(1.0 + 0.11 * ((k % 5) - 2))
                        k += 1
        return nets, self.conds(), self.pde(diff)

    def oracle(self, flat):
        """(nets fp64 carrying ``flat``, enforcers, pde) on the oracle"""
        from oracle import autograd_ref as R
        def make(i, o, h, a):
            if a.startswith("mono"):
                k = int(a[4])
                return torch.nn.Sequential(R.MonomialRef(range(1, k + 1)), R.make_fcnn(i * k, o, h, a[6:], dtype=torch.float64))
            if a.startswith("resnet-"):
                return R.ResnetRef(i, o, h, a[7:], dtype=torch.float64)
            return R.make_fcnn(i, o, h, a, dtype=torch.float64)
        nets = [make(*spec) for spec in self.net_specs]
        R.set_flat(nets, flat.double())
        return nets, self.enforcers(R.ref_diff), self.pde(R.ref_diff)


def _R():
    from oracle import autograd_ref
    return autograd_ref


def _cat(*cols):
    return torch.cat(cols, dim=1)


def build(name):
    from neurodiffeq_amd import conditions as C
    zero = lambda s: 0 * s
    if name == "pendulum":            # u'' + sin u = 0, u(0) = 1, u'(0) = 0.5  -> second-order IVP, 1-D mask 0b1
        pde = lambda D: (lambda u, t: [D(u, t, order=2) + torch.sin(u)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
        enf = lambda D: [lambda net, t: 1.0 + t * 0.5 + (1 - torch.exp(-t)) ** 2 * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "coupled_sin":         # two sin networks, non-polynomial coefficients (division, exp, sqrt)
        pde = lambda D: (lambda u, v, t: [D(u, t) - v / (1 + t ** 2), D(v, t) + u * torch.exp(-t) - torch.sqrt(t + 1)])
        conds = lambda: [C.IVP(0.0, 0.0), C.IVP(0.0, 1.0)]
        enf = lambda D: [lambda net, t: 0.0 + (1 - torch.exp(-t)) * net(t), lambda net, t: 1.0 + (1 - torch.exp(-t)) * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "sin")] * 2, [(0.0, 3.0)], pde, conds, enf)
    if name == "bvp_tanh":            # u'' - tanh(u) u' = cos t between two Dirichlet ends
        pde = lambda D: (lambda u, t: [D(u, t, order=2) - torch.tanh(u) * D(u, t) - torch.cos(t)])
        conds = lambda: [C.DirichletBVP(0.0, 1.0, 2.0, -1.0)]

        def e(net, t):
            s = t / 2.0
            return 1.0 * (1 - s) - 1.0 * s + (1 - torch.exp((1 - s) * s)) * net(t)
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, lambda D: [e])
    if name == "helmholtz_xy":        # all three second derivatives -> 2-D mask 0b111
        f0 = lambda y: torch.sin(PI * y)
        g1 = lambda x: x * (1 - x)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + 0.5 * D(D(u, x), y) + 4.0 * u
                                          - torch.sin(PI * x) * torch.cos(PI * y)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, g1)]

        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, g1)])
    if name == "advection":           # first order only (2-D, mask 0), raw network
        pde = lambda D: (lambda u, x, y: [D(u, x) + 2.0 * D(u, y) - u ** 2 + torch.exp(-x * y)])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name == "heat_wide":           # three hidden layers of 64, IBVP1D Dirichlet-Dirichlet
        u0 = lambda x: torch.sin(PI * x)
        pde = lambda D: (lambda u, x, t: [D(u, t) - 0.1 * D(u, x, order=2) + u ** 3])
        conds = lambda: [C.IBVP1D(0.0, 1.0, 0.0, u0, x_min_val=zero, x_max_val=zero)]

        return System(name, 2, [(2, 1, (64, 64, 64), "tanh")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().ibvp1d_dd(0.0, 1.0, 0.0, u0, zero, zero)])
    if name == "stokes_like":         # three networks sharing two coordinates, mixed stream sets (Laplacian + first order)
        def pde(D):
            def f(u, v, p, x, y):
                return [D(u, x, order=2) + D(u, y, order=2) - D(p, x) + torch.sin(PI * y),
                        D(v, x, order=2) + D(v, y, order=2) - D(p, y), D(u, x) + D(v, y)]
            return f
        conds = lambda: [C.NoCondition()] * 3
        raw = lambda net, x, y: net(_cat(x, y))
        return System(name, 2, [(2, 1, (32, 32), "tanh")] * 3, [(0.0, 1.0), (0.0, 1.0)], pde, conds, lambda D: [raw] * 3)
    if name == "swish_laplace":       # Swish network (beta = 1) on the C2 problem: Laplacian stream + DirichletBVP2D
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, (32, 32), "swish")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name == "sigmoid_mixed":       # sigmoid network, full 2-D Hessian, nonlinear in u
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) - 2.0 * D(D(u, x), y) + 3.0 * D(u, y, order=2) + u * D(u, x)
                                          - torch.cos(x + y)])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "sigmoid")], [(-1.0, 1.0), (-1.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name == "swish_ode":           # second-order ODE on a Swish network coupled to a first-order one on a sigmoid network
        pde = lambda D: (lambda u, v, t: [D(u, t, order=2) + v * D(u, t) + u, D(v, t) - u * v + torch.sin(t)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.0), C.IVP(0.0, 0.5)]
        enf = lambda D: [lambda net, t: 1.0 + t * 0.0 + (1 - torch.exp(-t)) ** 2 * net(t), _R().ivp(0.0, 0.5)]
        return System(name, 1, [(1, 1, (32, 32), "swish"), (1, 1, (32, 32), "sigmoid")], [(0.0, 2.0)], pde, conds, enf)
    if name == "bundle_decay":        # bundle of IVPs: u' + lam u = 0, u(0) = u0 with (u0, lam) as extra network inputs
        pde = lambda D: (lambda u, t, u0, lam: [D(u, t) + lam * u])
        conds = lambda: [C.BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0})]
        enf = lambda D: [lambda net, t, u0, lam: u0 + (1 - torch.exp(-t)) * net(_cat(t, u0, lam))]
        return System(name, 3, [(3, 1, (32, 32), "tanh")], [(0.0, 1.0), (0.5, 2.0), (0.5, 2.0)], pde, conds, enf)
    if name == "bundle_bvp":          # bundle of two-point problems: u'' + u = 0, u(0) = 0, u(1) = u1 sampled
        pde = lambda D: (lambda u, t, u1: [D(u, t, order=2) + u])
        conds = lambda: [C.BundleDirichletBVP(0.0, 0.0, 1.0, None, bundle_param_lookup={"u_1": 0})]

        def e(net, t, u1):
            s = (t - 0.0) / (1.0 - 0.0)
            return 0.0 * (1 - s) + u1 * s + (1 - torch.exp((1 - s) * s)) * net(_cat(t, u1))
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(0.0, 1.0), (-1.0, 1.0)], pde, conds, lambda D: [e])
    # ---- network shapes outside libndq.so's table: compiled on first use as extension modules (codegen.ensure_mlp_kernels)
    if name in ("shape_64x2", "shape_32x3", "shape_48x2", "shape_16x2_sin", "shape_32x1", "shape_50x2", "shape_20x3",
                "shape_40x2_sigmoid", "shape_10x1", "shape_64_32", "shape_24_40_12_sigmoid", "shape_32x6", "shape_16x8_sin"):
        # widths that are no multiple of 16 run padded (csrc/ndq_mlp.h: Cfg::HR); sigmoid: the padding units output 1/2
        hidden, act = {"shape_64x2": ((64, 64), "tanh"), "shape_32x3": ((32, 32, 32), "tanh"),
                       "shape_48x2": ((48, 48), "tanh"), "shape_16x2_sin": ((16, 16), "sin"),
                       "shape_32x1": ((32,), "tanh"), "shape_50x2": ((50, 50), "tanh"), "shape_20x3": ((20, 20, 20), "tanh"),
                       "shape_40x2_sigmoid": ((40, 40), "sigmoid"), "shape_10x1": ((10,), "sin"),
                       # more than four hidden layers of one width (the reference's FCNN takes any depth, networks.py:26-66)
                       "shape_32x6": ((32,) * 6, "tanh"), "shape_16x8_sin": ((16,) * 8, "sin"),
                       # hidden layers of different widths: laid out for the widest one (ndq_mlp_desc.widths)
                       "shape_64_32": ((64, 32), "tanh"), "shape_24_40_12_sigmoid": ((24, 40, 12), "sigmoid")}[name]
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + u * D(u, x) - torch.exp(-x * y)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, hidden, act)], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    # ---- more than three network inputs (round 3: d <= 6)
    if name == "heat4d":              # heat equation in three space dimensions + time: a Laplacian stream over 3 of 4 inputs
        pde = lambda D: (lambda u, x, y, z, t: [D(u, t) - 0.3 * (D(u, x, order=2) + D(u, y, order=2) + D(u, z, order=2))
                                                + u ** 2 - torch.exp(-t) * torch.sin(x + y * z)])
        conds = lambda: [C.NoCondition()]
        return System(name, 4, [(4, 1, (32, 32), "tanh")], [(-1.0, 1.0)] * 3 + [(0.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y, z, t: net(_cat(x, y, z, t))])
    if name == "mix5d":               # five inputs: first derivatives of all, one pure and one mixed second derivative
        pde = lambda D: (lambda u, a, b, c, d, e: [D(u, a) + u * D(u, b) - D(u, c, order=2) - D(u, d) * D(u, e)
                                                   + 0.5 * D(D(u, a), e) - torch.cos(a * b + c) * d])
        conds = lambda: [C.NoCondition()]
        return System(name, 5, [(5, 1, (32, 32), "sin")], [(-1.0, 1.0)] * 5, pde, conds,
                      lambda D: [lambda net, a, b, c, d, e: net(_cat(a, b, c, d, e))])
    if name == "bundle_osc":          # bundle of IVPs with THREE parameters: u'' + w^2 u = 0, u(0) = u0, u'(0) = v0 (BundleSolver1D)
        pde = lambda D: (lambda u, t, u0, v0, w: [D(u, t, order=2) + w ** 2 * u])
        conds = lambda: [C.BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0, "u_0_prime": 1})]
        enf = lambda D: [lambda net, t, u0, v0, w: u0 + t * v0 + (1 - torch.exp(-t)) ** 2 * net(_cat(t, u0, v0, w))]
        return System(name, 4, [(4, 1, (32, 32), "tanh")], [(0.0, 1.0), (0.5, 2.0), (-1.0, 1.0), (0.5, 2.0)], pde, conds, enf)
    if name == "resnet_laplace":      # Resnet (networks.py:73-106): FCNN + trainable linear skip, on the C2 problem
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, (32, 32), "resnet-tanh")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name == "resnet_ode":          # Resnet on a second-order ODE with a Neumann-form IVP (the skip feeds value and u')
        pde = lambda D: (lambda u, t: [D(u, t, order=2) + 0.5 * D(u, t) + u - torch.cos(t)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
        enf = lambda D: [lambda net, t: 1.0 + t * 0.5 + (1 - torch.exp(-t)) ** 2 * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "resnet-tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "aptx_burgers":        # APTx network (default parameters) on a Burgers-type problem; kernels built on first use
        u0 = lambda x: -torch.sin(PI * x)
        pde = lambda D: (lambda u, x, t: [D(u, t) + u * D(u, x) - 0.05 * D(u, x, order=2)])
        conds = lambda: [C.IBVP1D(-1.0, 1.0, 0.0, u0, x_min_val=zero, x_max_val=zero)]
        return System(name, 2, [(2, 1, (32, 32), "aptx")], [(-1.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().ibvp1d_dd(-1.0, 1.0, 0.0, u0, zero, zero)])
    if name == "mono_poisson":        # pure Laplacian -> the merged second-order stream on a monomial first layer
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) - torch.sin(PI * x) * y])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "mono3-tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name in ("mono_laplace", "mono_ode"):      # MonomialNN feature map in front of the network (networks.py:109-139)
        if name == "mono_ode":
            pde = lambda D: (lambda u, t: [D(u, t, order=2) + 0.5 * D(u, t) + u - torch.cos(t)])
            conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
            enf = lambda D: [lambda net, t: 1.0 + t * 0.5 + (1 - torch.exp(-t)) ** 2 * net(t)]
            return System(name, 1, [(1, 1, (32, 32), "mono4-sin")], [(0.0, 1.5)], pde, conds, enf)
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + 0.5 * D(D(u, x), y) + u * D(u, x) - torch.exp(-x * y)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, (32, 32), "mono3-tanh")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name == "ensemble_lv":         # ONE two-output network, EnsembleCondition (conditions.py:157-202) as ONE solver function
        def pde(D):
            def f(uv, t):             # the equations pick the columns apart themselves
                u, v = uv[:, 0:1], uv[:, 1:2]
                return [D(u, t) - (u - u * v), D(v, t) - (u * v - v)]
            return f
        conds = lambda: [C.EnsembleCondition(C.IVP(0.0, 1.5), C.IVP(0.0, 1.0))]

        def e(net, t):
            out, decay = net(t), 1 - torch.exp(-t)
            return _cat(1.5 + decay * out[:, 0:1], 1.0 + decay * out[:, 1:2])
        return System(name, 1, [(1, 2, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, lambda D: [e])
    if name in ("swish_fixed_laplace", "aptx_fixed_laplace"):      # fixed non-default activation parameters
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + u * D(u, x) - torch.exp(-x * y)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, (32, 32), name.split("_")[0] + "-fixed")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name in ("swish_tr_laplace", "aptx_tr_laplace", "aptx_tr_wide"):   # trainable activation parameters (networks.py:155-209)
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + u * D(u, x) - torch.exp(-x * y)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        hidden = (64, 64, 64) if name == "aptx_tr_wide" else (32, 32)
        return System(name, 2, [(2, 1, hidden, name.split("_")[0] + "-tr")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name == "swish_tr_system":     # two trainable-Swish networks on one coordinate (multi-network closure kernel)
        pde = lambda D: (lambda u, v, t: [D(u, t, order=2) + v * D(u, t) + u, D(v, t) - u * v + torch.sin(t)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.0), C.IVP(0.0, 0.5)]
        enf = lambda D: [lambda net, t: 1.0 + t * 0.0 + (1 - torch.exp(-t)) ** 2 * net(t), _R().ivp(0.0, 0.5)]
        return System(name, 1, [(1, 1, (32, 32), "swish-tr")] * 2, [(0.0, 2.0)], pde, conds, enf)
    if name == "aptx_tr_resnet":      # Resnet with trainable APTx parameters: skip weights and activation scalars together
        pde = lambda D: (lambda u, t: [D(u, t, order=2) + 0.5 * D(u, t) + u - torch.cos(t)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
        enf = lambda D: [lambda net, t: 1.0 + t * 0.5 + (1 - torch.exp(-t)) ** 2 * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "resnet-aptx-tr")], [(0.0, 2.0)], pde, conds, enf)
    if name == "kdv":                 # Korteweg-de Vries: a third-order derivative in x (diff(u, x, order=3), neurodiffeq.py:21-34)
        u0 = lambda x: 0.5 / torch.cosh(0.5 * x) ** 2
        pde = lambda D: (lambda u, x, t: [D(u, t) + 6.0 * u * D(u, x) + D(u, x, order=3)])
        conds = lambda: [C.IBVP1D(-1.0, 1.0, 0.0, u0, x_min_val=zero, x_max_val=zero)]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().ibvp1d_dd(-1.0, 1.0, 0.0, u0, zero, zero)])
    if name == "ode3":                # third-order ODE with a sin network and a mixed product of derivatives
        pde = lambda D: (lambda u, t: [D(u, t, order=3) + D(u, t, order=2) * D(u, t) + u - torch.sin(t)])
        conds = lambda: [C.IVP(0.0, 1.0)]
        return System(name, 1, [(1, 1, (32, 32), "sin")], [(0.0, 2.0)], pde, conds, lambda D: [_R().ivp(0.0, 1.0)])
    # fourth-order streams (round 6: diff(u, x, order=4), neurodiffeq.py:21-34 has no order limit)
    if name == "beam":                # static beam on an elastic foundation: u_tttt + u = q(t)  (pure fourth derivative, one input)
        pde = lambda D: (lambda u, t: [D(u, t, order=4) + u - torch.cos(t)])
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
        enf = lambda D: [lambda net, t: 1.0 + 0.5 * t + (1 - torch.exp(-t)) ** 2 * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "beam_sigmoid":        # ... with a sigmoid network, three layers, and the lower derivatives in the equation as well
        pde = lambda D: (lambda u, t: [D(u, t, order=4) + 0.3 * D(u, t, order=3) * D(u, t) - D(u, t, order=2) + u - torch.sin(t)])
        conds = lambda: [C.IVP(0.0, 1.0)]
        return System(name, 1, [(1, 1, (32, 32, 32), "sigmoid")], [(0.0, 2.0)], pde, conds, lambda D: [_R().ivp(0.0, 1.0)])
    if name == "biharmonic":          # plate equation: u_xxxx + 2 u_xxyy + u_yyyy = f  (the mixed quadruple xxyy by two diff calls)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=4) + 2.0 * D(D(u, x, order=2), y, order=2) + D(u, y, order=4)
                                          - torch.sin(x) * torch.cos(y)])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0)] * 2, pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name == "kuramoto":            # Kuramoto-Sivashinsky: u_t + u u_x + u_xx + u_xxxx, sin network, Dirichlet IBVP
        u0 = lambda x: torch.cos(3.0 * x) * (1.0 - x ** 2)
        pde = lambda D: (lambda u, x, t: [D(u, t) + u * D(u, x) + D(u, x, order=2) + 0.1 * D(u, x, order=4)])
        conds = lambda: [C.IBVP1D(-1.0, 1.0, 0.0, u0, x_min_val=zero, x_max_val=zero)]
        return System(name, 2, [(2, 1, (32, 32), "sin")], [(-1.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().ibvp1d_dd(-1.0, 1.0, 0.0, u0, zero, zero)])
    if name == "poisson3d":           # three coordinates, Laplacian -> one merged second-order stream
        pde = lambda D: (lambda u, x, y, z: [D(u, x, order=2) + D(u, y, order=2) + D(u, z, order=2)
                                             + torch.exp(-(x ** 2 + y ** 2 + z ** 2))])
        conds = lambda: [C.NoCondition()]
        return System(name, 3, [(3, 1, (32, 32), "tanh")], [(-1.0, 1.0)] * 3, pde, conds,
                      lambda D: [lambda net, x, y, z: net(_cat(x, y, z))])
    if name == "hessian3d":           # every entry of the 3-D Hessian with different weights -> mask 0b111111
        pde = lambda D: (lambda u, x, y, z: [D(u, x, order=2) + 2.0 * D(u, y, order=2) + 3.0 * D(u, z, order=2)
                                             + D(D(u, x), y) - D(D(u, y), z) + 0.5 * D(D(u, z), x) + u * D(u, z)])
        conds = lambda: [C.NoCondition()]
        return System(name, 3, [(3, 1, (32, 32), "tanh")], [(-1.0, 1.0)] * 3, pde, conds,
                      lambda D: [lambda net, x, y, z: net(_cat(x, y, z))])
    if name == "shell":               # Laplace in a spherical shell, SolverSpherical's setting: diagonal second derivatives
        f = lambda th, ph: torch.cos(th)
        g = lambda th, ph: torch.sin(th) * torch.cos(ph)
        r0, r1 = 0.5, 2.0

        def pde(D):
            def lap(u, r, th, ph):             # operators.py spherical_laplacian, expanded form
                return [D(u, r, order=2) + 2.0 / r * D(u, r) + (D(u, th, order=2) + D(u, th) / torch.tan(th)) / r ** 2
                        + D(u, ph, order=2) / (r * torch.sin(th)) ** 2]
            return lap
        conds = lambda: [C.DirichletBVPSpherical(r0, f, r1, g)]

        def e(net, r, th, ph):
            s = (r - r0) / (r1 - r0)
            return f(th, ph) * (1 - s) + g(th, ph) * s + (1 - torch.exp((1 - s) * s)) * net(_cat(r, th, ph))
        return System(name, 3, [(3, 1, (32, 32), "tanh")], [(r0, r1), (0.3, 2.8), (0.0, 2 * PI)], pde, conds, lambda D: [e])
    # ---- ordinary torch ops of piecewise / clipped equations (VERDICT r4 missing #5: comparisons -> masks, where, clamp, relu,
    # maximum / minimum, sign, log1p, expm1, atan2, erf), with the (sub)gradients torch defines
    if name == "piecewise_source":    # Poisson with a piecewise source term and a clipped reaction term
        f0 = lambda y: torch.sin(PI * y)
        pde = lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) - torch.where(x > 0.5, torch.sin(PI * y), 0.25)
                                          + torch.clamp(u, -0.2, 0.3) * (y <= 0.7) + torch.clamp(u * u, max=0.05)])
        conds = lambda: [C.DirichletBVP2D(0, f0, 1, zero, 0, zero, 1, zero)]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(0.0, 1.0), (0.0, 1.0)], pde, conds,
                      lambda D: [_R().dirichlet_bvp2d(0, f0, 1, zero, 0, zero, 1, zero)])
    if name == "relu_ode":            # masks on the SOLUTION: relu, leaky branch through where, maximum against a coefficient
        pde = lambda D: (lambda u, t: [D(u, t) + torch.relu(u - 1.0) - torch.maximum(torch.sin(t), 0.3 * u)
                                       + torch.sign(t - 1.0) * torch.log1p(u ** 2) + torch.where(u > 1.2, u, 0.1 * u)
                                       + torch.nn.functional.leaky_relu(D(u, t), 0.2)])
        conds = lambda: [C.IVP(0.0, 1.0)]
        enf = lambda D: [lambda net, t: 1.0 + (1 - torch.exp(-t)) * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "atan2_adv":           # atan2 / expm1 / erf / minimum / abs inside a first-order 2-D equation
        pde = lambda D: (lambda u, x, y: [D(u, x) + 2.0 * D(u, y) - torch.atan2(u, 1.0 + x ** 2) + torch.expm1(-x * y) + torch.erf(u)
                                          + torch.minimum(u, x) - torch.atan(u * y) + (u - 0.1).abs().clamp_min(0.05)])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name == "rounding_ode":        # floor / ceil / round / trunc / frac / fmod / remainder (zero gradient, exact on dyadic arguments)
        pde = lambda D: (lambda u, t: [D(u, t) + torch.floor(4.0 * t) * u - torch.ceil(2.0 * t) * torch.sin(u) + torch.round(8.0 * t) * 0.1 * u ** 2
                                       - torch.trunc(2.0 * t - 1.0) * D(u, t) + torch.frac(2.0 * t) * u + torch.fmod(t, 0.25) * u
                                       + torch.remainder(t - 1.0, 0.5) * torch.cos(u) + (4.0 * t).floor() * 0.05 + torch.round(t, decimals=1) * 0.0])
        conds = lambda: [C.IVP(0.0, 1.0)]
        enf = lambda D: [lambda net, t: 1.0 + (1 - torch.exp(-t)) * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "activations_ode":     # torch.nn.functional activations applied to the solution inside the equation
        # (F.hardsigmoid traces too, but torch's own double-precision backward of it multiplies by float(1 / 6): 8e-9 off, so it
        # cannot sit in a system the fp64 oracle is compared with at 1e-11)
        F = torch.nn.functional
        pde = lambda D: (lambda u, t: [D(u, t) + F.softplus(u) - F.softplus(3.0 * u, beta=2.0, threshold=4.0) + F.elu(u - 1.0, alpha=0.7)
                                       + F.selu(u - 1.1) + F.celu(u - 0.9, alpha=1.3) + F.gelu(u) - F.gelu(u - 0.5, approximate="tanh")
                                       + F.silu(u) + F.mish(u - 1.0) + F.softsign(u) + F.hardtanh(u, -0.3, 0.7) + F.relu6(5.0 * u)
                                       + F.hardswish(u + 1.5) + F.logsigmoid(u) + F.threshold(u, 1.05, -0.2)
                                       + F.tanhshrink(u) + torch.sigmoid(D(u, t))])
        conds = lambda: [C.IVP(0.0, 1.0)]
        enf = lambda D: [lambda net, t: 1.0 + (1 - torch.exp(-t)) * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    if name == "special_2d":          # inverse trigonometric / hyperbolic functions, logarithms, two-operand functions, Tensor methods
        pde = lambda D: (lambda u, x, y: [D(u, x) + 2.0 * D(u, y) + torch.asin(0.5 * torch.tanh(u)) - torch.acos(0.4 * x) + torch.asinh(u * y)
                                          + torch.acosh(2.0 + u ** 2) - torch.atanh(0.5 * torch.tanh(u + x)) + torch.rsqrt(1.0 + u ** 2)
                                          + torch.log10(2.0 + x) * u - torch.log2(3.0 + y * u) + torch.exp2(0.5 * u) + torch.erfc(u)
                                          + torch.sinc(u) + torch.sinc(0.0 * x) + torch.hypot(u, 1.0 + x) - torch.logaddexp(u, y) + torch.lerp(u, x, 0.3)
                                          + torch.addcmul(u, x, y, value=0.5) - torch.addcdiv(u, x, 2.0 + y, value=0.25) + u.mul(x).add(y, alpha=2.0)
                                          - u.sub(x).div(3.0 + y) + (1.0 + u * u).rsqrt() * 0.1 + u.floor() * 0.0 + u.clamp_min(-0.2).erfc() * 0.1 + (2.0 * x).exp2() * 0.01
                                          + torch.special.xlogy(u * u, 2.0 + x) * 0.1 + torch.logit(0.5 + 0.2 * torch.tanh(u)) * 0.1
                                          + u[:, 0:1] * x.new_tensor(0.25) + torch.special.expit(u) * 0.1 + u * y.new_ones(1)
                                          + u * torch.zeros(1, dtype=u.dtype, device=u.device)
                                          + u * torch.tanh(u).detach() * 0.3 + (u * x).detach() * D(u, y) * 0.1 + torch.detach(u) ** 2 * 0.05])
        conds = lambda: [C.NoCondition()]
        return System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde, conds,
                      lambda D: [lambda net, x, y: net(_cat(x, y))])
    if name == "autograd_grad_ode":   # derivatives taken with torch.autograd.grad by hand (what `diff` wraps), nested for the second order
        def pde(D):
            def f(u, t):
                ones = torch.ones_like(u)
                ut, = torch.autograd.grad(u, t, grad_outputs=ones, create_graph=True)
                utt = torch.autograd.grad(ut, [t], [torch.ones_like(t)], create_graph=True)[0]
                return [utt + 0.5 * ut + torch.sin(u) - torch.cos(t) + 0.1 * torch.autograd.grad([u, ut], t, [2.0 * ones, ones], create_graph=True)[0]]
            return f
        conds = lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]
        enf = lambda D: [lambda net, t: 1.0 + 0.5 * t + (1 - torch.exp(-t)) ** 2 * net(t)]
        return System(name, 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf)
    raise KeyError(name)


NAMES = ["pendulum", "coupled_sin", "bvp_tanh", "helmholtz_xy", "advection", "heat_wide", "stokes_like", "kdv", "ode3", "poisson3d",
         "hessian3d", "shell", "swish_laplace", "sigmoid_mixed", "swish_ode", "bundle_decay", "bundle_bvp", "shape_64x2", "shape_32x3", "shape_48x2",
         "shape_16x2_sin", "shape_32x1", "aptx_burgers", "resnet_laplace", "resnet_ode", "swish_tr_laplace", "aptx_tr_laplace",
         "aptx_tr_wide", "swish_tr_system", "aptx_tr_resnet", "shape_50x2", "shape_20x3", "shape_40x2_sigmoid", "shape_10x1",
         "swish_fixed_laplace", "aptx_fixed_laplace", "ensemble_lv", "shape_64_32", "shape_24_40_12_sigmoid",
         "mono_laplace", "mono_ode", "mono_poisson", "shape_32x6", "shape_16x8_sin", "heat4d", "mix5d", "bundle_osc",
         "piecewise_source", "relu_ode", "atan2_adv", "rounding_ode", "activations_ode", "special_2d", "autograd_grad_ode",
         "beam", "beam_sigmoid", "biharmonic", "kuramoto"]


def spherical_solver_problem():
    """(diff_eqs, conditions) of the SolverSpherical end-to-end test: Laplace between two spheres through
    operators.spherical_laplacian, default FCNN(3, 1) network."""
    from neurodiffeq_amd.conditions import DirichletBVPSpherical
    from neurodiffeq_amd.operators import spherical_laplacian
    cond = DirichletBVPSpherical(0.5, lambda th, ph: torch.cos(th), 2.0, lambda th, ph: 0.25 * torch.cos(th))
    return (lambda u, r, th, ph: [spherical_laplacian(u, r, th, ph)]), [cond]
