"""``diff`` -- the reference's derivative operator (neurodiffeq/neurodiffeq.py:7-82), with two evaluation modes.

* On traced values (inside a fused solver step) ``diff`` is symbolic: derivatives of network outputs become the
  derivative streams the gfx950 forward kernel computes (no autograd graph, no reverse sweeps).
* On ordinary tensors it has the reference's semantics exactly: ``order`` reverse sweeps with ``create_graph=True``,
  zeros (that still ``requires_grad``) when ``u`` does not depend on ``t``, ``ValueError`` unless both are ``(N, 1)``.
"""
import warnings
from functools import wraps

import torch

from ._version_utils import deprecated_alias
from .symbolic import Sym, sym_diff


_alias_x_to_u = deprecated_alias(x="u")       # ``x=``: the deprecated name of the first argument (neurodiffeq.py:6,37,63)


@_alias_x_to_u
def unsafe_diff(u, t, order=1):
    if isinstance(u, Sym) or isinstance(t, Sym):
        return sym_diff(u, t, order=order)
    der = u
    for _ in range(order):
        der, = torch.autograd.grad(der, t, grad_outputs=torch.ones_like(der), create_graph=True, allow_unused=True)
        if der is None:
            return torch.zeros_like(t, requires_grad=True)
        der.requires_grad_()
    return der


@_alias_x_to_u
def safe_diff(u, t, order=1):
    if isinstance(u, Sym) or isinstance(t, Sym):
        return sym_diff(u, t, order=order)
    if len(u.shape) != 2 or len(t.shape) != 2 or u.shape[1] != 1 or t.shape[1] != 1:
        raise ValueError(f"Input shapes must both be (n_samples, 1); got {tuple(u.shape)} (dependent variable) and "
                         f"{tuple(t.shape)} (independent variable). Consider reshaping with `x = x.view(-1, 1)`, or "
                         f"use `unsafe_diff` for the legacy behaviour.")
    if u.shape != t.shape:
        raise ValueError(f"Input shapes must be the same; got {tuple(u.shape)} != {tuple(t.shape)}. "
                         f"Use `unsafe_diff` for the legacy behaviour.")
    return unsafe_diff(u, t, order=order)


@_alias_x_to_u
def diff(u, t, order=1, shape_check=True):
    return safe_diff(u, t, order=order) if shape_check else unsafe_diff(u, t, order=order)
