"""Conditions (re-parameterisations that make initial/boundary conditions hold exactly) with the reference's class
names, constructor arguments and ``enforce`` / ``parameterize`` protocol (neurodiffeq/conditions.py:8-57).

``parameterize`` bodies are plain arithmetic on their arguments, so the SAME code runs
  * on torch tensors (composite path, solution evaluation), and
  * on :class:`neurodiffeq_amd.symbolic.Sym` proxies while a solver traces the system for the fused gfx950 path --
    there the network output is a symbol and ``enforce`` asks the active trace for it instead of calling the net.
User-defined subclasses that only override ``parameterize`` with ordinary arithmetic are traced the same way.
"""
import warnings

import torch

from ._version_utils import deprecated_alias
from .neurodiffeq import safe_diff as diff
from .symbolic import Sym, SymMat, TraceUnsupported, current_graph


def _const_like(ref, value):
    """A column filled with ``value`` shaped like ``ref`` (a traced constant inside a trace)."""
    if isinstance(ref, Sym):
        return Sym(ref.g, ref.g.const(value))
    return torch.full_like(ref, float(value))


def _exp(x):
    return x._un("exp") if isinstance(x, (Sym, SymMat)) else torch.exp(x)


def _boundary_column(x, value):
    """A column holding the boundary coordinate ``value`` at every point, differentiable like a coordinate: a fresh
    leaf tensor (conditions.py:577-579), or -- while tracing -- a virtual coordinate of the graph, which makes the
    network call on it a further evaluation site of the same parameters."""
    if isinstance(x, Sym):
        return Sym(x.g, x.g.vcoord(value), leaf=True)
    return value * torch.ones_like(x, requires_grad=True)


def _raw_output(net, coordinates, ith_unit):
    """Network output on cat(coords, 1) (conditions.py:52-55), or its symbol while tracing."""
    if any(isinstance(c, Sym) for c in coordinates):
        return current_graph().net_symbol(net, coordinates, ith_unit)
    out = net(torch.cat(coordinates, dim=1))
    if ith_unit is not None:
        out = out[:, ith_unit].view(-1, 1)
    return out


class BaseCondition:
    """Base class: ``enforce(net, *coords)`` = ``parameterize(net(cat(coords)), *coords)``."""

    def __init__(self):
        self.ith_unit = None

    def parameterize(self, output_tensor, *input_tensors):
        raise ValueError(f"Abstract {self.__class__.__name__} cannot be parameterized")

    def enforce(self, net, *coordinates):
        return self.parameterize(_raw_output(net, coordinates, self.ith_unit), *coordinates)

    def set_impose_on(self, ith_unit):
        warnings.warn(f"`{self.__class__.__name__}.set_impose_on` is deprecated and will be removed in the future",
                      DeprecationWarning)
        self.ith_unit = ith_unit


class NoCondition(BaseCondition):
    """Identity re-parameterisation (conditions.py:205-222)."""

    def parameterize(self, output_tensor, *input_tensors):
        return output_tensor


class EnsembleCondition(BaseCondition):
    """One sub-condition per output unit of a multi-output network (conditions.py:157-202)."""

    def __init__(self, *sub_conditions, force=False):
        super().__init__()
        for i, c in enumerate(sub_conditions):
            if c.__class__.enforce != BaseCondition.enforce:
                msg = (f"{c.__class__.__name__} (index={i})'s overrides BaseCondition's `.enforce` method. "
                       f"Ensembl'ing is likely not going to work.")
                if not force:
                    raise ValueError(msg + "\nTry with `force=True` if you know what you are doing.")
                warnings.warn(msg)
        self.conditions = sub_conditions

    def parameterize(self, output_tensor, *input_tensors):
        if isinstance(output_tensor, Sym):
            output_tensor = SymMat([output_tensor])
        if output_tensor.shape[1] != len(self.conditions):
            raise ValueError(f"number of output units ({output_tensor.shape[1]}) "
                             f"differs from number of conditions ({len(self.conditions)})")
        cols = [c.parameterize(output_tensor[:, i].view(-1, 1), *input_tensors) for i, c in enumerate(self.conditions)]
        return torch.cat(cols, dim=1)


class IVP(BaseCondition):
    """u(t0) = u0, optionally u'(t0) = u0'  (conditions.py:225-267)."""

    @deprecated_alias(x_0="u_0", x_0_prime="u_0_prime")      # (conditions.py:242; both spellings at once: KeyError)
    def __init__(self, t_0, u_0=None, u_0_prime=None):
        super().__init__()
        self.t_0, self.u_0, self.u_0_prime = t_0, u_0, u_0_prime

    def parameterize(self, output_tensor, t):
        decay = 1 - _exp(-t + self.t_0)
        if self.u_0_prime is None:
            return self.u_0 + decay * output_tensor
        return self.u_0 + (t - self.t_0) * self.u_0_prime + (decay ** 2) * output_tensor


class _BundleConditionMixin:
    """Conditions whose parameters (t_0, u_0, ...) may be *inputs* sampled by the generator instead of constants
    (conditions.py:78-135): ``bundle_param_lookup`` maps a parameter name to its position among the extra coordinates
    ``theta`` handed to ``parameterize(output, t, *theta)``."""

    def __init__(self, bundle_param_lookup=None, allowed_params=None):
        self.bundle_param_lookup = bundle_param_lookup or {}
        if isinstance(allowed_params, str):
            allowed_params = set(allowed_params)
        if allowed_params:
            illegal = set(self.bundle_param_lookup) - set(allowed_params)
            if illegal:
                raise ValueError(f"The following parameter(s) are not allowed in `bundle_parameters_lookup`: {illegal}.\n"
                                 f"Supported parameter name(s) are: {allowed_params}.")

    def _get_parameter(self, param_name, thetas, override_name=None):
        if param_name in self.bundle_param_lookup:
            return thetas[self.bundle_param_lookup[param_name]]
        return getattr(self, override_name or param_name)


class BundleIVP(BaseCondition, _BundleConditionMixin):
    """IVP whose t_0 / u_0 / u_0' may be bundle inputs (conditions.py:270-345)."""

    @deprecated_alias(x_0="u_0", x_0_prime="u_0_prime", bundle_conditions="bundle_param_lookup")     # (conditions.py:295)
    def __init__(self, t_0=None, u_0=None, u_0_prime=None, bundle_param_lookup=None):
        BaseCondition.__init__(self)
        _BundleConditionMixin.__init__(self, bundle_param_lookup, allowed_params=["t_0", "u_0", "u_0_prime"])
        self.t_0, self.u_0, self.u_0_prime = t_0, u_0, u_0_prime

    def parameterize(self, output_tensor, t, *theta):
        t_0 = self._get_parameter("t_0", theta)
        u_0 = self._get_parameter("u_0", theta)
        u_0_prime = self._get_parameter("u_0_prime", theta)
        decay = 1 - _exp(-t + t_0)
        if u_0_prime is None:
            return u_0 + decay * output_tensor
        return u_0 + (t - t_0) * u_0_prime + (decay ** 2) * output_tensor


class BundleDirichletBVP(BaseCondition, _BundleConditionMixin):
    """Two-point Dirichlet condition whose ends / end values may be bundle inputs (conditions.py:348-395)."""

    @deprecated_alias(bundle_conditions="bundle_param_lookup")                                        # (conditions.py:363)
    def __init__(self, t_0, u_0, t_1, u_1, bundle_param_lookup=None):
        BaseCondition.__init__(self)
        _BundleConditionMixin.__init__(self, bundle_param_lookup, allowed_params=["t_0", "u_0", "t_1", "u_1"])
        self.t_0, self.u_0, self.t_1, self.u_1 = t_0, u_0, t_1, u_1

    def parameterize(self, output_tensor, t, *theta):
        u_0, u_1 = self._get_parameter("u_0", theta), self._get_parameter("u_1", theta)
        t_0, t_1 = self._get_parameter("t_0", theta), self._get_parameter("t_1", theta)
        s = (t - t_0) / (t_1 - t_0)
        return u_0 * (1 - s) + u_1 * s + (1 - _exp((1 - s) * s)) * output_tensor


class DirichletBVP(BaseCondition):
    """u(t0) = u0, u(t1) = u1  (conditions.py:398-435)."""

    @deprecated_alias(x_0="u_0", x_1="u_1")                                                          # (conditions.py:412)
    def __init__(self, t_0, u_0, t_1, u_1):
        super().__init__()
        self.t_0, self.u_0, self.t_1, self.u_1 = t_0, u_0, t_1, u_1

    def parameterize(self, output_tensor, t):
        s = (t - self.t_0) / (self.t_1 - self.t_0)
        return self.u_0 * (1 - s) + self.u_1 * s + (1 - _exp((1 - s) * s)) * output_tensor


class DirichletBVP2D(BaseCondition):
    """Dirichlet data on the four edges of [x0,x1] x [y0,y1]  (conditions.py:438-509):
    u = A(x,y) + xt (1-xt) yt (1-yt) N, with A the transfinite interpolant of f0, f1 (x edges) and g0, g1 (y edges)."""

    def __init__(self, x_min, x_min_val, x_max, x_max_val, y_min, y_min_val, y_max, y_max_val):
        super().__init__()
        self.x0, self.f0 = x_min, x_min_val
        self.x1, self.f1 = x_max, x_max_val
        self.y0, self.g0 = y_min, y_min_val
        self.y1, self.g1 = y_max, y_max_val

    def parameterize(self, output_tensor, x, y):
        xt = (x - self.x0) / (self.x1 - self.x0)
        yt = (y - self.y0) / (self.y1 - self.y0)
        xa, xb = _const_like(x, self.x0), _const_like(x, self.x1)

        def edge_minus_corners(g):          # g(x) minus the linear blend of its corner values
            return g(x) - ((1 - xt) * g(xa) + xt * g(xb))

        a = (1 - xt) * self.f0(y) + xt * self.f1(y) \
            + (1 - yt) * edge_minus_corners(self.g0) + yt * edge_minus_corners(self.g1)
        return a + xt * (1 - xt) * yt * (1 - yt) * output_tensor


class IBVP1D(BaseCondition):
    """Initial condition u(x,t0) = u0(x) plus a Dirichlet or Neumann condition at each end of [x0, x1]
    (conditions.py:512-712).  The Neumann forms evaluate the network at boundary points as well and differentiate
    it there; while tracing, such a boundary column is a virtual coordinate of the graph and the call on it a further
    evaluation site of the same parameters (engine.FusedSystem runs the MLP kernels once per site)."""

    def __init__(self, x_min, x_max, t_min, t_min_val, x_min_val=None, x_min_prime=None, x_max_val=None,
                 x_max_prime=None):
        super().__init__()
        given = [c is not None for c in (x_min_val, x_min_prime, x_max_val, x_max_prime)]
        if sum(given) != 2 or (x_min_val and x_min_prime) or (x_max_val and x_max_prime):
            raise NotImplementedError("Sorry, this boundary condition is not implemented.")
        self.x_min, self.x_min_val, self.x_min_prime = x_min, x_min_val, x_min_prime
        self.x_max, self.x_max_val, self.x_max_prime = x_max, x_max_val, x_max_prime
        self.t_min, self.t_min_val = t_min, t_min_val

    def _kind(self):
        return ("d" if self.x_min_val else "n") + ("d" if self.x_max_val else "n")

    def enforce(self, net, x, t):
        kind = self._kind()
        u = _raw_output(net, (x, t), self.ith_unit)
        if kind == "dd":
            return self.parameterize(u, x, t)
        extra = []
        for need, xb in ((kind[0] == "n", self.x_min), (kind[1] == "n", self.x_max)):
            if need:
                xe = _boundary_column(x, xb)
                extra += [_raw_output(net, (xe, t), self.ith_unit), xe]
        return self.parameterize(u, x, t, *extra)

    def parameterize(self, u, x, t, *additional_tensors):
        kind = self._kind()
        w = self.x_max - self.x_min
        xt = (x - self.x_min) / w
        tc = _const_like(t, self.t_min)
        grow = 1 - _exp(-(t - self.t_min))
        delta = lambda fn: fn(t) - fn(tc)          # boundary datum minus its value at t0
        if kind == "dd":
            a = self.t_min_val(x) + xt * delta(self.x_max_val) + (1 - xt) * delta(self.x_min_val)
            return a + xt * (1 - xt) * grow * u
        if kind == "dn":
            u1, x1 = additional_tensors
            a = delta(self.x_min_val) + self.t_min_val(x) + xt * w * delta(self.x_max_prime)
            return a + xt * grow * (u - w * diff(u1, x1) - u1)
        if kind == "nd":
            u0, x0 = additional_tensors
            a = delta(self.x_max_val) + self.t_min_val(x) + (xt - 1) * w * delta(self.x_min_prime)
            return a + (1 - xt) * grow * (u + w * diff(u0, x0) - u0)
        u0, x0, u1, x1 = additional_tensors
        a = self.t_min_val(x) - 0.5 * (1 - xt) ** 2 * w * delta(self.x_min_prime) \
            + 0.5 * xt ** 2 * w * delta(self.x_max_prime)
        d0 = diff(u0, x0)
        return a + grow * (u - xt * w * d0 + 0.5 * xt ** 2 * w * (d0 - diff(u1, x1)))


class DoubleEndedBVP1D(BaseCondition):
    """Two-point boundary conditions in one variable, each end either Dirichlet (``x_*_val``) or Neumann
    (``x_*_prime``) (conditions.py:715-884).  A Neumann end needs the network and its derivative AT that end, so
    ``enforce`` is overridden to evaluate the network there as well (a further evaluation site on the fused path)."""

    def __init__(self, x_min, x_max, x_min_val=None, x_min_prime=None, x_max_val=None, x_max_prime=None):
        super().__init__()
        given = sum(c is not None for c in (x_min_val, x_min_prime, x_max_val, x_max_prime))
        if given != 2 or (x_min_val and x_min_prime) or (x_max_val and x_max_prime):
            raise NotImplementedError("Sorry, this boundary condition is not implemented.")
        self.x_min, self.x_min_val, self.x_min_prime = x_min, x_min_val, x_min_prime
        self.x_max, self.x_max_val, self.x_max_prime = x_max, x_max_val, x_max_prime

    def _kind(self):
        return ("d" if self.x_min_val is not None else "n") + ("d" if self.x_max_val is not None else "n")

    def enforce(self, net, x):
        kind = self._kind()
        u = _raw_output(net, (x,), self.ith_unit)
        if kind == "dd":
            return self.parameterize(u, x)
        extra = []
        for need, xb in ((kind[0] == "n", self.x_min), (kind[1] == "n", self.x_max)):
            if need:
                xe = _boundary_column(x, xb)
                extra += [_raw_output(net, (xe,), self.ith_unit), xe]
        return self.parameterize(u, x, *extra)

    def parameterize(self, u, x, *additional_tensors):
        kind = self._kind()
        w = self.x_max - self.x_min
        xt = (x - self.x_min) / w
        if kind == "dd":
            return self.x_min_val * (1 - xt) + self.x_max_val * xt + xt * (1 - xt) * u
        if kind == "dn":
            u1, x1 = additional_tensors
            a = (1 - xt) * self.x_min_val + 0.5 * xt ** 2 * self.x_max_prime * w
            return a + xt * (u - u1 + self.x_min_val - diff(u1, x1) * w)
        if kind == "nd":
            u0, x0 = additional_tensors
            a = xt * self.x_max_val - 0.5 * (1 - xt) ** 2 * self.x_min_prime * w
            return a + (1 - xt) * (u - u0 + self.x_max_val + diff(u0, x0) * w)
        u0, x0, u1, x1 = additional_tensors
        a = -0.5 * (1 - xt) ** 2 * w * self.x_min_prime + 0.5 * xt ** 2 * w * self.x_max_prime
        return a + 0.5 * xt ** 2 * (u - u1 - 0.5 * diff(u1, x1) * w) + 0.5 * (1 - xt) ** 2 * (u - u0 + 0.5 * diff(u0, x0) * w)


class IrregularBoundaryCondition(BaseCondition):
    """Base class of conditions on irregular domains (conditions.py:138-154): ``in_domain`` tells monitors which points
    to draw; every point by default."""

    def in_domain(self, *coordinates):
        import numpy as np
        return np.ones_like(coordinates[0], dtype=bool)


# ------------------------------------------------------------------------------------------------- spherical shells
def _abs(x):
    return x._un("abs") if isinstance(x, (Sym, SymMat)) else torch.abs(x)


def _tanh(x):
    return x._un("tanh") if isinstance(x, (Sym, SymMat)) else torch.tanh(x)


class DirichletBVPSpherical(BaseCondition):
    """u(r0,θ,φ) = f(θ,φ) and optionally u(r1,θ,φ) = g(θ,φ) on a spherical shell (conditions.py:887-945)."""

    def __init__(self, r_0, f, r_1=None, g=None):
        super().__init__()
        if (r_1 is None) ^ (g is None):
            raise ValueError(f"r_1 and g must be both/neither set to None; got r_1={r_1}, g={g}")
        self.r_0, self.r_1, self.f, self.g = r_0, r_1, f, g

    def parameterize(self, output_tensor, r, theta, phi):
        if self.r_1 is None:
            return (1 - _exp(-_abs(r - self.r_0))) * output_tensor + self.f(theta, phi)
        s = (r - self.r_0) / (self.r_1 - self.r_0)
        return self.f(theta, phi) * (1 - s) + self.g(theta, phi) * s + (1. - _exp((1 - s) * s)) * output_tensor


class InfDirichletBVPSpherical(BaseCondition):
    """u(r0) = f, u(r→∞) = g (conditions.py:948-1001)."""

    def __init__(self, r_0, f, g, order=1):
        super().__init__()
        self.r_0, self.f, self.g, self.order = r_0, f, g, order

    def parameterize(self, output_tensor, r, theta, phi):
        dr = r - self.r_0
        decay, rise = _exp(-self.order * dr), _tanh(dr)
        return self.f(theta, phi) * decay + self.g(theta, phi) * rise + decay * rise * output_tensor


class DirichletBVPSphericalBasis(BaseCondition):
    """Dirichlet data for the vector of harmonic coefficients R(r) (one network output per basis function) on one or
    two spherical boundaries (conditions.py:1004-1096); ``R_0`` / ``R_1`` are scalars or length-k rows."""

    def __init__(self, r_0, R_0, r_1=None, R_1=None, max_degree=None):
        super().__init__()
        if (r_1 is None) ^ (R_1 is None):
            raise ValueError(f"r_1 and R_1 must be both/neither set to None; got r_1={r_1}, R_1={R_1}")
        if max_degree is not None:
            warnings.warn("`max_degree` is deprecated and ignored", FutureWarning)
        self.r_0, self.r_1, self.R_0, self.R_1 = r_0, r_1, R_0, R_1

    def parameterize(self, output_tensor, r):
        if self.r_1 is None:
            return (1 - _exp(-r + self.r_0)) * output_tensor + self.R_0
        s = (r - self.r_0) / (self.r_1 - self.r_0)
        return self.R_0 * (1 - s) + self.R_1 * s + (1. - _exp((1 - s) * s)) * output_tensor


class InfDirichletBVPSphericalBasis(BaseCondition):
    """Coefficient-vector version of :class:`InfDirichletBVPSpherical` (conditions.py:1099-1166)."""

    def __init__(self, r_0, R_0, R_inf, order=1, max_degree=None):
        super().__init__()
        if max_degree is not None:
            warnings.warn("`max_degree` is deprecated and ignored", FutureWarning)
        self.r_0, self.R_0, self.R_inf, self.order = r_0, R_0, R_inf, order

    def parameterize(self, output_tensor, r):
        dr = r - self.r_0
        decay, rise = _exp(-self.order * dr), _tanh(dr)
        return self.R_0 * decay + self.R_inf * rise + decay * rise * output_tensor
