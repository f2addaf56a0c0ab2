"""Differential operators with the reference's names and (N, 1) conventions (neurodiffeq/operators.py).

Every operator is written in terms of :func:`diff`, so inside a fused solver step (traced values) it expands to a
symbolic expression over derivative streams -- e.g. ``spherical_laplacian`` differentiates the *expressions*
``r**2 * u_r`` and ``sin(theta) * u_theta`` (operators.py:203-207) without any autograd sweep -- while on ordinary
tensors it behaves exactly like the reference (autograd sweeps, zeros for unused variables)."""
import torch

from .neurodiffeq import safe_diff as diff
from .symbolic import Sym


def _split_u_x(*us_xs):
    if len(us_xs) == 0 or len(us_xs) % 2 != 0:
        raise RuntimeError("Number of us and xs must be equal and positive")
    half = len(us_xs) // 2
    return us_xs[:half], us_xs[half:]


def _sin(x):
    return x.sin() if isinstance(x, Sym) else torch.sin(x)


def _cos(x):
    return x.cos() if isinstance(x, Sym) else torch.cos(x)


# --------------------------------------------------------------------------------------------- cartesian
def grad(u, *xs):
    """[du/dx_1, ..., du/dx_n]  (operators.py:15-33: one autograd.grad for all xs; unused -> zeros that require grad)."""
    if isinstance(u, Sym) or any(isinstance(x, Sym) for x in xs):
        return [diff(u, x) for x in xs]
    gs = torch.autograd.grad(u, xs, grad_outputs=torch.ones_like(u), create_graph=True, allow_unused=True)
    return [torch.zeros_like(x, requires_grad=True) if g is None else g.requires_grad_(True) for x, g in zip(xs, gs)]


def div(*us_xs):
    us, xs = _split_u_x(*us_xs)
    return sum(diff(u, x) for u, x in zip(us, xs))


def curl(u_x, u_y, u_z, x, y, z):
    dxy, dxz = grad(u_x, y, z)
    dyx, dyz = grad(u_y, x, z)
    dzx, dzy = grad(u_z, x, y)
    return dzy - dyz, dxz - dzx, dyx - dxy


def laplacian(u, *xs):
    return sum(diff(g, x) for g, x in zip(grad(u, *xs), xs))


def vector_laplacian(u_x, u_y, u_z, x, y, z):
    return laplacian(u_x, x, y, z), laplacian(u_y, x, y, z), laplacian(u_z, x, y, z)


# --------------------------------------------------------------------------------------------- spherical (r, theta, phi)
def spherical_grad(u, r, theta, phi):
    u_r, u_t, u_p = grad(u, r, theta, phi)
    return u_r, u_t / r, u_p / (r * _sin(theta))


def spherical_div(u_r, u_theta, u_phi, r, theta, phi):
    s = _sin(theta)
    radial = diff(u_r * r ** 2, r) / r
    angular = (diff(u_theta * s, theta) + diff(u_phi, phi)) / s
    return (radial + angular) / r


def spherical_curl(u_r, u_theta, u_phi, r, theta, phi):
    ur_t, ur_p = grad(u_r, theta, phi)
    ut_r, ut_p = grad(u_theta, r, phi)
    up_r, up_t = grad(u_phi, r, theta)
    s, c = _sin(theta), _cos(theta)
    c_r = (up_t + (u_phi * c - ut_p) / s) / r
    c_t = (ur_p / s - u_phi) / r - up_r
    c_p = ut_r + (u_theta - ur_t) / r
    return c_r, c_t, c_p


def _spherical_scalar_laplacian(u, r, theta, phi, s, r2):
    u_r, u_t, u_p = grad(u, r, theta, phi)
    return (diff(r2 * u_r, r) + diff(s * u_t, theta) / s + diff(u_p, phi) / s ** 2) / r2


def spherical_laplacian(u, r, theta, phi):
    return _spherical_scalar_laplacian(u, r, theta, phi, _sin(theta), r ** 2)


def spherical_vector_laplacian(u_r, u_theta, u_phi, r, theta, phi):
    s, c, r2 = _sin(theta), _cos(theta), r ** 2
    lap_r = _spherical_scalar_laplacian(u_r, r, theta, phi, s, r2)
    lap_t = _spherical_scalar_laplacian(u_theta, r, theta, phi, s, r2)
    lap_p = _spherical_scalar_laplacian(u_phi, r, theta, phi, s, r2)
    ur_t, ur_p = grad(u_r, theta, phi)
    ut_t, ut_p = grad(u_theta, theta, phi)
    up_p = diff(u_phi, phi)
    v_r = lap_r - 2 * (u_r + ut_t + (c * u_theta + up_p) / s) / r2
    v_t = lap_t + (2 * ur_t - (u_theta + 2 * c * up_p) / s ** 2) / r2
    v_p = lap_p + ((2 * c * ut_p - u_phi) / s + 2 * ur_p) / (r2 * s)
    return v_r, v_t, v_p


def spherical_to_cartesian(r, theta, phi):
    rho = r * _sin(theta)
    return rho * _cos(phi), rho * _sin(phi), r * _cos(theta)


def cartesian_to_spherical(x, y, z):
    rho2 = x ** 2 + y ** 2
    return torch.sqrt(rho2 + z ** 2), torch.atan2(torch.sqrt(rho2), z), torch.atan2(y, x)


# --------------------------------------------------------------------------------------------- cylindrical (rho, phi, z)
def cylindrical_grad(u, rho, phi, z):
    u_r, u_p, u_z = grad(u, rho, phi, z)
    return u_r, u_p / rho, u_z


def cylindrical_div(u_rho, u_phi, u_z, rho, phi, z):
    return diff(u_rho, rho) + (u_rho + diff(u_phi, phi)) / rho + diff(u_z, z)


def cylindrical_curl(u_rho, u_phi, u_z, rho, phi, z):
    ur_p, ur_z = grad(u_rho, phi, z)
    up_r, up_z = grad(u_phi, rho, z)
    uz_r, uz_p = grad(u_z, rho, phi)
    return uz_p / rho - up_z, ur_z - uz_r, up_r + (u_phi - ur_p) / rho


def cylindrical_laplacian(u, rho, phi, z):
    u_r, u_p, u_z = grad(u, rho, phi, z)
    return diff(u_r, rho) + u_r / rho + diff(u_p, phi) / rho ** 2 + diff(u_z, z)


def cylindrical_vector_laplacian(u_rho, u_phi, u_z, rho, phi, z):
    rho2 = rho ** 2
    up_p = diff(u_phi, phi)
    ur_p = diff(u_rho, phi)
    return (cylindrical_laplacian(u_rho, rho, phi, z) - (u_rho + 2 * up_p) / rho2,
            cylindrical_laplacian(u_phi, rho, phi, z) + (2 * ur_p - u_phi) / rho2,
            cylindrical_laplacian(u_z, rho, phi, z))


def cylindrical_to_cartesian(rho, phi, z):
    return rho * _cos(phi), rho * _sin(phi), z


def cartesian_to_cylindrical(x, y, z):
    return torch.sqrt(x ** 2 + y ** 2), torch.atan2(y, x), z
