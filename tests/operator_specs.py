"""Operator / function-basis evaluations shared by tests/golden/make_golden.py (reference modules) and
tests/test_operators_golden.py (neurodiffeq_amd modules): ``SPECS[name](O, B)`` with O = operators module, B =
function_basis module, on fixed closed-form fields in fp64.  Every entry returns one (N, k) tensor."""
import torch

F64 = torch.float64


def _pts(n=20, seed=4):
    g = torch.Generator().manual_seed(seed)
    mk = lambda lo, hi: (lo + (hi - lo) * torch.rand(n, 1, generator=g, dtype=F64)).requires_grad_(True)
    return mk(0.5, 2.0), mk(0.3, 2.8), mk(0.2, 6.0)


def _fields(a, b, c):
    u = torch.sin(a) * torch.exp(0.3 * b) + a * b * torch.cos(c)
    v = torch.cos(a * b) + c ** 2 / (1 + a)
    w = a ** 2 * torch.tanh(b * c / 5)
    return u, v, w


def _cat(x):
    return torch.cat(list(x), dim=1) if isinstance(x, (list, tuple)) else x


def _vec(fn_name):
    def run(O, B):
        a, b, c = _pts()
        return _cat(getattr(O, fn_name)(*_fields(a, b, c), a, b, c))
    return run


def _scal(fn_name):
    def run(O, B):
        a, b, c = _pts()
        return _cat(getattr(O, fn_name)(_fields(a, b, c)[0], a, b, c))
    return run


def _coeffs(r, k):
    return torch.cat([torch.sin((j + 1) * r) / (j + 1) + r ** 2 * 0.1 * j for j in range(k)], dim=1)


def _harm_lap(O, B):
    r, th, ph = _pts()
    return B.HarmonicsLaplacian(3)(_coeffs(r, 16), r, th, ph)


def _zonal_lap(O, B):
    r, th, ph = _pts()
    return B.ZonalSphericalHarmonicsLaplacian(degrees=[0, 2, 4])(_coeffs(r, 3), r, th, ph)


def _fourier_lap(O, B):
    r, th, ph = _pts()
    return B.FourierLaplacian(3)(_coeffs(r, 7), r, ph)


SPECS = {name: _vec(name) for name in ("div", "curl", "vector_laplacian", "spherical_div", "spherical_curl",
                                       "spherical_vector_laplacian", "cylindrical_div", "cylindrical_curl",
                                       "cylindrical_vector_laplacian")}
SPECS.update({name: _scal(name) for name in ("grad", "laplacian", "spherical_grad", "spherical_laplacian",
                                             "cylindrical_grad", "cylindrical_laplacian")})
SPECS.update({
    "spherical_to_cartesian": lambda O, B: _cat(O.spherical_to_cartesian(*_pts())),
    "cartesian_to_spherical": lambda O, B: _cat(O.cartesian_to_spherical(*O.spherical_to_cartesian(*_pts()))),
    "cylindrical_to_cartesian": lambda O, B: _cat(O.cylindrical_to_cartesian(*_pts())),
    "cartesian_to_cylindrical": lambda O, B: _cat(O.cartesian_to_cylindrical(*O.cylindrical_to_cartesian(*_pts()))),
    "real_spherical_harmonics": lambda O, B: B.RealSphericalHarmonics(4)(*_pts()[1:]),
    "harmonics_laplacian": _harm_lap,
    "legendre_basis": lambda O, B: B.LegendreBasis(6)(torch.cos(_pts()[1])),
    "zonal_harmonics": lambda O, B: B.ZonalSphericalHarmonics(max_degree=5)(*_pts()[1:]),
    "zonal_laplacian": _zonal_lap,
    "fourier_series": lambda O, B: B.RealFourierSeries(4)(_pts()[2]),
    "fourier_laplacian": _fourier_lap,
    "custom_basis": lambda O, B: B.CustomBasis([lambda x, y: x * y, lambda x, y: x - y])(*_pts()[:2]),
})
