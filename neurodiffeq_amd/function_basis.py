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
    # (function_basis.py:18-19: P_0 of a column that requires grad is itself attached -- callers differentiate it)
    return Sym(x.g, x.g.const(1.0)) if isinstance(x, Sym) else torch.ones_like(x, requires_grad=x.requires_grad)


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


def _legendre_coefficients(degree):
    """Coefficients of P_degree, highest power first, by Bonnet's recursion (n+1) P_{n+1} = (2n+1) x P_n - n P_{n-1}
    (what scipy.special.legendre holds, without needing scipy on the hot path)."""
    import numpy as np
    prev, cur = np.array([1.0]), np.array([1.0, 0.0])
    if degree == 0:
        return prev
    for n in range(1, degree):
        nxt = ((2 * n + 1) * np.append(cur, 0.0) - n * np.concatenate([[0.0, 0.0], prev])) / (n + 1)
        prev, cur = cur, nxt
    return cur


class LegendrePolynomial:
    """P_degree(x) on (N, 1) columns (function_basis.py:11-22)."""

    def __init__(self, degree):
        self.degree = degree
        self.coefficients = _legendre_coefficients(degree)

    def __call__(self, x):
        if self.degree == 0:
            return _ones(x)
        if self.degree == 1:
            return x * 1
        return sum(float(c) * x ** (self.degree - i) for i, c in enumerate(self.coefficients))


class LegendreBasis(FunctionBasis):
    """[P_0(x), ..., P_max_degree(x)] (function_basis.py:45-51)."""

    def __init__(self, max_degree):
        self.basis_module = CustomBasis([LegendrePolynomial(d) for d in range(max_degree + 1)])

    def __call__(self, x):
        return self.basis_module(x)


class ZonalSphericalHarmonics(FunctionBasis):
    """Spherical harmonics of order 0: sqrt((2l+1)/(4 pi)) P_l(cos theta) for the given degrees (function_basis.py:54-88)."""

    def __init__(self, max_degree=None, degrees=None):
        import numpy as np
        if max_degree is None and degrees is None:
            raise ValueError("Either `max_degree` or `degrees` must be specified")
        if max_degree is not None and degrees is not None:
            import warnings
            warnings.warn(f"degrees={degrees} specified, ignoring max_degree={max_degree}")
        self.max_degree = max_degree
        self.degrees = list(range(max_degree + 1)) if degrees is None else degrees
        norms = [float(np.sqrt((2 * l + 1) / (4 * np.pi))) for l in self.degrees]
        polys = [LegendrePolynomial(l) for l in self.degrees]
        self.basis_module = CustomBasis([lambda theta, c=c, fn=fn: fn(_cos(theta)) * c for c, fn in zip(norms, polys)])

    def __call__(self, theta, phi):
        return self.basis_module(theta)


class ZonalSphericalHarmonicsLaplacian(BasisOperator):
    """Laplacian of sum_l R_l(r) Y_l^0(theta) acting on the coefficient columns (function_basis.py:92-116)."""

    def __init__(self, max_degree=None, degrees=None):
        self.harmonics_fn = ZonalSphericalHarmonics(max_degree=max_degree, degrees=degrees)
        self.laplacian_coefficients = torch.tensor([-l * (l + 1) for l in self.harmonics_fn.degrees], dtype=torch.float)

    def __call__(self, base_coeffs, r, theta, phi):
        k = base_coeffs.shape[1]
        radial = _cat([diff(base_coeffs[:, j:j + 1] * r, r, order=2) for j in range(k)]) / r
        coeff = self.laplacian_coefficients if isinstance(base_coeffs, SymMat) else self.laplacian_coefficients.to(base_coeffs)
        products = (radial + coeff * base_coeffs / r ** 2) * self.harmonics_fn(theta, phi)
        return products.sum(dim=1, keepdim=True)


def _deprecated_alias(cls, old_name):
    class _Alias(cls):
        def __init__(self, *args, **kwargs):
            import warnings
            warnings.warn(f"{old_name} is deprecated, use {cls.__name__} instead", FutureWarning)
            super().__init__(*args, **kwargs)
    _Alias.__name__ = old_name
    return _Alias


ZeroOrderSphericalHarmonics = _deprecated_alias(ZonalSphericalHarmonics, "ZeroOrderSphericalHarmonics")
ZeroOrderSphericalHarmonicsLaplacian = _deprecated_alias(ZonalSphericalHarmonicsLaplacian, "ZeroOrderSphericalHarmonicsLaplacian")


class RealFourierSeries(FunctionBasis):
    """[1/2, sin(phi), cos(phi), sin(2 phi), cos(2 phi), ...] up to max_degree (function_basis.py:121-155)."""

    def __init__(self, max_degree=12):
        self.max_degree = max_degree
        terms = [lambda th: _ones(th) * 0.5]
        for d in range(1, max_degree + 1):
            terms.append(lambda th, d=d: _sin(d * th))
            terms.append(lambda th, d=d: _cos(d * th))
        self.basis_module = CustomBasis(terms)

    def __call__(self, phi):
        return self.basis_module(phi)


class FourierLaplacian(BasisOperator):
    """Polar Laplacian of sum_i R_i(r) F_i(phi) acting on the coefficient columns (function_basis.py:158-190)."""

    def __init__(self, max_degree=12):
        self.harmonics_fn = RealFourierSeries(max_degree=max_degree)
        self.laplacian_coefficients = torch.tensor([0] + [-d ** 2 for d in range(1, max_degree + 1) for _ in range(2)],
                                                   dtype=torch.float)

    def __call__(self, R, r, phi):
        k = R.shape[1]
        radial = _cat([diff(R[:, j:j + 1], r) / r + diff(R[:, j:j + 1], r, order=2) for j in range(k)])
        coeff = self.laplacian_coefficients if isinstance(R, SymMat) else self.laplacian_coefficients.to(R)
        products = (radial + coeff * R / r ** 2) * self.harmonics_fn(phi)
        return products.sum(dim=1, keepdim=True)


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
