"""Function bases with the reference's names (neurodiffeq/function_basis.py).  On the hot path:
``RealSphericalHarmonics`` (config C4 expands the solution as sum_k R_k(r) Y_k(theta, phi), pde_spherical.py:253-254) --
its closed forms are plain arithmetic on (N, 1) columns, so they trace symbolically like any other user code and their
theta / phi derivatives inside ``spherical_laplacian`` come out of symbolic differentiation."""
from abc import ABC, abstractmethod

import torch

from .neurodiffeq import safe_diff as diff
from .symbolic import Sym, SymMat


def _sin(x):
    return x.sin() if isinstance(x, Sym) else torch.sin(x)


def _cos(x):
    return x.cos() if isinstance(x, Sym) else torch.cos(x)


def _ones(x):
    return Sym(x.g, x.g.const(1.0)) if isinstance(x, Sym) else torch.ones_like(x)


def _cat(cols):
    return SymMat(cols) if isinstance(cols[0], Sym) else torch.cat(cols, dim=1)


class FunctionBasis(ABC):
    @abstractmethod
    def __call__(self, *args, **kwargs):
        pass  # pragma: no cover


class BasisOperator(ABC):
    @abstractmethod
    def __call__(self, *args, **kwargs):
        pass  # pragma: no cover


class CustomBasis(FunctionBasis):
    def __init__(self, fns):
        self.fns = fns

    def __call__(self, *xs):
        return _cat([fn(*xs) for fn in self.fns])


def _real_harmonics_table():
    """Real spherical harmonics Y_l^m for l <= 4, ordered (l, m = -l..l); normalisation WITHOUT the sqrt(1/pi) factor,
    as in the reference (function_basis.py:195-229, https://en.wikipedia.org/wiki/Table_of_spherical_harmonics)."""
    s, c = _sin, _cos
    c2p = lambda p: c(2 * p)
    return [
        # l = 0
        [lambda t, p: _ones(t) * 0.5],
        # l = 1
        [lambda t, p: s(t) * s(p) * 0.866025404,
         lambda t, p: c(t) * 0.866025404,
         lambda t, p: s(t) * c(p) * 0.866025404],
        # l = 2
        [lambda t, p: s(t) ** 2 * s(p) * c(p) * 1.936491673,
         lambda t, p: s(t) * c(t) * s(p) * 1.936491673,
         lambda t, p: (2 * c(t) ** 2 - s(t) ** 2) * 0.559016994,
         lambda t, p: s(t) * c(t) * c(p) * 1.936491673,
         lambda t, p: s(t) ** 2 * c2p(p) * 0.968245837],
        # l = 3
        [lambda t, p: s(t) ** 3 * (3 * c(p) ** 2 * s(p) - s(p) ** 3) * 1.045825033,
         lambda t, p: s(t) ** 2 * c(t) * c(p) * s(p) * 5.123475383,
         lambda t, p: s(t) * (4 * c(t) ** 2 - s(t) ** 2) * s(p) * 0.810092587,
         lambda t, p: (2 * c(t) ** 3 - 3 * c(t) * s(t) ** 2) * 0.661437828,
         lambda t, p: s(t) * (4 * c(t) ** 2 - s(t) ** 2) * c(p) * 0.810092587,
         lambda t, p: c(t) * s(t) ** 2 * c2p(p) * 2.561737691,
         lambda t, p: s(t) ** 3 * (c(p) ** 3 - 3 * s(p) ** 2 * c(p)) * 1.045825033],
        # l = 4
        [lambda t, p: s(t) ** 4 * (s(p) * c(p) * c2p(p)) * 4.437059837,
         lambda t, p: s(t) ** 3 * c(t) * (3 * c(p) ** 2 * s(p) - s(p) ** 3) * 3.1374751,
         lambda t, p: s(t) ** 2 * (s(p) * c(p)) * (7 * c(t) ** 2 - 1) * 1.677050983,
         lambda t, p: s(t) * c(t) * s(p) * (7 * c(t) ** 2 - 3) * 1.185854123,
         lambda t, p: (35 * c(t) ** 4 - 30 * c(t) ** 2 + 3) * 0.1875,
         lambda t, p: s(t) * c(t) * c(p) * (7 * c(t) ** 2 - 3) * 1.185854123,
         lambda t, p: s(t) ** 2 * c2p(p) * (7 * c(t) ** 2 - 1) * 0.838525492,
         lambda t, p: s(t) ** 3 * c(t) * (c(p) ** 3 - 3 * c(p) * s(p) ** 2) * 3.1374751,
         lambda t, p: s(t) ** 4 * (c(p) ** 4 - 6 * c(p) ** 2 * s(p) ** 2 + s(p) ** 4) * 1.109264959],
    ]


class RealSphericalHarmonics(FunctionBasis):
    """(N,1) theta, phi -> (N, (max_degree+1)^2) matrix of real spherical harmonics (function_basis.py:232-271)."""

    def __init__(self, max_degree=4):
        super().__init__()
        if max_degree >= 5:
            raise NotImplementedError(f"max_degree = {max_degree} not implemented for {self.__class__.__name__} yet")
        self.max_degree = max_degree
        self.harmonics = [y for band in _real_harmonics_table()[:max_degree + 1] for y in band]

    def __call__(self, theta, phi):
        if len(theta.shape) != 2 or theta.shape[1] != 1:
            raise ValueError(f"theta must be of shape (-1, 1); got {theta.shape}")
        if theta.shape != phi.shape:
            raise ValueError(f"theta/phi must be of the same shape; got f{theta.shape} and f{phi.shape}")
        return _cat([y(theta, phi) for y in self.harmonics])


class HarmonicsLaplacian(BasisOperator):
    """Laplacian of sum_k R_k(r) Y_k: Y_k (d2(r R_k)/dr2 / r - l(l+1) R_k / r^2), avoiding the 1/sin(theta) terms
    (function_basis.py:274-300)."""

    def __init__(self, max_degree=4):
        self.harmonics_fn = RealSphericalHarmonics(max_degree=max_degree)
        self.laplacian_coefficients = torch.tensor(
            [-l * (l + 1) * 1.0 for l in range(max_degree + 1) for _ in range(-l, l + 1)])

    def __call__(self, R, r, theta, phi):
        k = R.shape[1]
        radial = _cat([diff(R[:, j:j + 1] * r, r, order=2) for j in range(k)]) / r
        coeff = self.laplacian_coefficients if isinstance(R, SymMat) else self.laplacian_coefficients.to(R)
        angular = coeff * R / r ** 2
        products = (radial + angular) * self.harmonics_fn(theta, phi)
        return products.sum(dim=1, keepdim=True)
