"""Condition constructions shared by tests/golden/make_golden.py (reference modules) and tests/test_conditions_golden.py
(neurodiffeq_amd modules).  ``SPECS[name](C, N)`` returns ``cond.enforce(net, *coords)`` in fp64 for a network built
whose parameters are set from a private generator, on fixed coordinates."""
import math

import torch

F64 = torch.float64


def _coords(n_coords, n=24, seed=1, lo=0.05, hi=0.95):
    g = torch.Generator().manual_seed(seed)
    return [(lo + (hi - lo) * torch.rand(n, 1, generator=g, dtype=F64)).requires_grad_(True) for _ in range(n_coords)]


def _net(N, n_in, n_out=1, seed=2):
    """FCNN(n_in, n_out, (16, 16)) in fp64 with parameters drawn here (independent of the package's default dtype)"""
    net = N.FCNN(n_in, n_out, hidden_units=(16, 16)).double()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.rand(p.shape, generator=g, dtype=F64) - 0.5)
    return net


def _du(u, x):
    return torch.autograd.grad(u, x, grad_outputs=torch.ones_like(u), create_graph=True)[0]


def _ibvp(C, N, **kw):
    x, t = _coords(2)
    return C.IBVP1D(0.0, 1.0, 0.0, lambda x: torch.sin(math.pi * x), **kw).enforce(_net(N, 2), x, t)


def _debvp(C, N, **kw):
    (x,) = _coords(1)
    return C.DoubleEndedBVP1D(0.0, 1.0, **kw).enforce(_net(N, 1), x)


def _ensemble(C, N):
    (t,) = _coords(1)
    return C.EnsembleCondition(C.IVP(0.0, 1.0), C.IVP(0.0, -2.0, u_0_prime=0.3), C.DirichletBVP(0.0, 0.5, 1.0, 1.5)).enforce(_net(N, 1, 3), t)


def _basis(C, N, inf):
    (r,) = _coords(1, lo=0.6, hi=1.9)
    R0, R1 = torch.linspace(-1, 1, 9, dtype=F64), torch.linspace(2, 3, 9, dtype=F64)
    cond = C.InfDirichletBVPSphericalBasis(0.5, R0, R1, order=1) if inf else C.DirichletBVPSphericalBasis(0.5, R0, 2.0, R1)
    return cond.enforce(_net(N, 1, 9), r)


f_ang = lambda th, ph: torch.cos(th) * torch.sin(ph)
g_ang = lambda th, ph: torch.sin(th) + ph

SPECS = {
    "no_condition": lambda C, N: C.NoCondition().enforce(_net(N, 2), *_coords(2)),
    "ivp": lambda C, N: C.IVP(0.3, 1.5).enforce(_net(N, 1), *_coords(1)),
    "ivp_prime": lambda C, N: C.IVP(0.3, 1.5, u_0_prime=-0.7).enforce(_net(N, 1), *_coords(1)),
    "dirichlet_bvp": lambda C, N: C.DirichletBVP(0.0, 1.0, 1.0, -3.0).enforce(_net(N, 1), *_coords(1)),
    "bundle_ivp": lambda C, N: C.BundleIVP(t_0=0.0, bundle_param_lookup={"u_0": 0, "u_0_prime": 1}).enforce(_net(N, 3), *_coords(3)),
    "bundle_ivp_t0": lambda C, N: C.BundleIVP(u_0=1.0, bundle_param_lookup={"t_0": 0}).enforce(_net(N, 2), *_coords(2)),
    "bundle_bvp": lambda C, N: C.BundleDirichletBVP(0.0, None, 1.0, 2.0, bundle_param_lookup={"u_0": 0}).enforce(_net(N, 2), *_coords(2)),
    "bvp2d": lambda C, N: C.DirichletBVP2D(0, lambda y: torch.sin(math.pi * y), 1, lambda y: y ** 2, 0, lambda x: x * 0,
                                           1, lambda x: x * (1 - x) + 1).enforce(_net(N, 2), *_coords(2)),
    "ibvp_dd": lambda C, N: _ibvp(C, N, x_min_val=lambda t: t, x_max_val=lambda t: t ** 2),
    "ibvp_dn": lambda C, N: _ibvp(C, N, x_min_val=lambda t: t, x_max_prime=lambda t: torch.cos(t)),
    "ibvp_nd": lambda C, N: _ibvp(C, N, x_min_prime=lambda t: 1 + t, x_max_val=lambda t: t ** 2),
    "ibvp_nn": lambda C, N: _ibvp(C, N, x_min_prime=lambda t: 1 + t, x_max_prime=lambda t: torch.cos(t)),
    "debvp_dd": lambda C, N: _debvp(C, N, x_min_val=1.0, x_max_val=-1.0),
    "debvp_dn": lambda C, N: _debvp(C, N, x_min_val=1.0, x_max_prime=0.5),
    "debvp_nd": lambda C, N: _debvp(C, N, x_min_prime=-0.5, x_max_val=2.0),
    "debvp_nn": lambda C, N: _debvp(C, N, x_min_prime=-0.5, x_max_prime=0.5),
    "ensemble": _ensemble,
    "sph_two": lambda C, N: C.DirichletBVPSpherical(0.5, f_ang, 2.0, g_ang).enforce(_net(N, 3), *_coords(3, lo=0.6, hi=1.9)),
    "sph_one": lambda C, N: C.DirichletBVPSpherical(0.5, f_ang).enforce(_net(N, 3), *_coords(3, lo=0.6, hi=1.9)),
    "sph_inf": lambda C, N: C.InfDirichletBVPSpherical(0.5, f_ang, g_ang, order=2).enforce(_net(N, 3), *_coords(3, lo=0.6, hi=1.9)),
    "sph_basis": lambda C, N: _basis(C, N, False),
    "sph_basis_inf": lambda C, N: _basis(C, N, True),
}
