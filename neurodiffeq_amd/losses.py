"""Loss functions ``loss_fn(residual (N, n_eq), funcs, coords) -> scalar`` (reference: neurodiffeq/losses.py:5-35).

``l2`` (= the solver default, solvers.py:218), ``l1`` and ``infinity`` are per-point terms averaged over the batch: the
fused gfx950 path evaluates them (and their adjoint seeds) inside the generated pointwise code.  The Sobolev norms are
the l2 loss of the residual list extended by d(sum_e r_e)/dx_a; the solver traces them that way: first-order systems
stay within second-order network streams, second-order PDEs use the third-order streams (``ndq_mlp_desc.mask3``: tanh /
sin / sigmoid networks) -- third-order equations use the fourth-order streams (round 6); only what would need fifth-order streams runs on the composite path."""
import torch

from .operators import grad


def _l1_norm(residual, funcs, coords):
    return residual.abs().mean()


def _l2_norm(residual, funcs, coords):
    return (residual ** 2).mean()


def _infinity_norm(residual, funcs, coords):
    return residual.abs().max(dim=1)[0].mean()


def _h1_norm(residual, funcs, coords):
    return (torch.cat([residual, *grad(residual, *coords)], dim=1) ** 2).mean()


def _h1_semi_norm(residual, funcs, coords):
    return (torch.cat(grad(residual, *coords), dim=1) ** 2).mean()


_losses = {"l1": _l1_norm, "l2": _l2_norm, "infinity": _infinity_norm, "h1": _h1_norm, "h1 semi": _h1_semi_norm}
