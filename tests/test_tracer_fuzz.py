"""Random equations through the tracer (CPU): seeded expression trees over the torch functions the tracer claims to cover --
arithmetic, powers, elementary / special functions, piecewise functions (``where``, ``clamp``, ``relu``, ``maximum``),
rounding of coordinates, stop-gradients, first and second derivatives taken with ``diff`` or with ``torch.autograd.grad`` --
evaluated twice from the same source string: by the product (tracer + symbolic differentiation + generated pointwise code
compiled with gcc, between the jet oracle's network streams and VJP: ``test_trace_codegen.host_closure``) and by the fp64
autograd oracle running the very same callable on real tensors.  Residuals, loss and parameter gradient must agree.

The hand-written zoo systems (tests/zoo.py) pin the functions one by one; this test pins their COMPOSITIONS (chain rule
through nested piecewise functions, constant folding, hash-consing of common subexpressions, derivative rules meeting the
adjoint emitter in combinations nobody wrote down)."""
import random

import numpy as np
import pytest
import torch

from neurodiffeq_amd import conditions as C
from tests import zoo
from tests.test_trace_codegen import host_closure, rel_l2

F = torch.nn.functional

# unary functions: (source template, safe for any real argument?)  -- arguments of the unsafe ones are wrapped below
UNARY = ["torch.sin({a})", "torch.cos({a})", "torch.tanh({a})", "torch.sigmoid({a})", "torch.atan({a})", "torch.erf({a})",
         "torch.exp(-({a}) ** 2)", "torch.log1p(({a}) ** 2)", "torch.sqrt(1.0 + ({a}) ** 2)", "torch.rsqrt(1.5 + ({a}) ** 2)",
         "torch.abs({a})", "torch.relu({a})", "F.softplus({a})", "F.silu({a})", "F.gelu({a})", "F.elu({a})", "F.softsign({a})",
         "torch.clamp({a}, -0.4, 0.6)", "torch.asinh({a})", "torch.expm1(-torch.abs({a}))", "F.leaky_relu({a}, 0.1)",
         "torch.sinh(0.3 * ({a}))", "torch.cosh(0.3 * ({a}))", "({a}).detach()", "torch.log(2.0 + torch.tanh({a}))",
         "({a}) ** 2", "({a}) ** 3", "(1.0 + ({a}) ** 2) ** 0.75", "(1.0 + ({a}) ** 2) ** -0.5", "torch.sinc(0.5 * ({a}))",
         "torch.asin(0.6 * torch.tanh({a}))", "torch.logaddexp({a}, 0.3 * ({a}))", "F.hardtanh({a}, -0.3, 0.8)", "F.mish({a})"]
BINARY = ["(({a}) + ({b}))", "(({a}) - ({b}))", "(({a}) * ({b}))", "(({a}) / (1.5 + ({b}) ** 2))", "torch.maximum({a}, {b})",
          "torch.minimum({a}, {b})", "torch.where(({a}) > ({b}), {a}, 0.5 * ({b}))", "torch.atan2({a}, 1.5 + ({b}) ** 2)",
          "torch.hypot({a}, 1.0 + 0.0 * ({b}))", "torch.lerp({a}, {b}, 0.3)", "(({a}) * ({b}).detach())"]


def _expr(rng, leaves, depth, coord):
    if depth == 0 or rng.random() < 0.15:
        return rng.choice(leaves)
    r = rng.random()
    if r < 0.55:
        return rng.choice(UNARY).format(a=_expr(rng, leaves, depth - 1, coord))
    if r < 0.95:
        return rng.choice(BINARY).format(a=_expr(rng, leaves, depth - 1, coord), b=_expr(rng, leaves, depth - 1, coord))
    return f"({rng.uniform(-1.5, 1.5):.3f} + 0.0 * {coord})"          # (a constant column: torch functions want tensors)


def _system(seed):
    """A random first- or second-order equation in one (ODE) or two (PDE) coordinates; returns (System, source)."""
    rng = random.Random(seed)
    if seed % 2 == 0:
        # ODE: leaves u, t, u_t (by diff or by torch.autograd.grad), piecewise-constant functions of the coordinate
        ut = rng.choice(["D(u, t)", "torch.autograd.grad(u, t, grad_outputs=torch.ones_like(u), create_graph=True)[0]"])
        leaves = ["u", "t", "ut", "torch.floor(4.0 * t)", "torch.frac(2.0 * t)", "(0.7 + 0.0 * t)", "(u * t)"]
        body = _expr(rng, leaves, 4, "t")
        second = rng.random() < 0.5
        src = (f"lambda D: (lambda u, t: (lambda ut: [{'D(u, t, order=2) + ' if second else ''}ut + 0.3 * ({body})])({ut}))")
        conds = (lambda: [C.IVP(0.0, 1.0, u_0_prime=0.5)]) if second else (lambda: [C.IVP(0.0, 1.0)])
        enf = (lambda D: [lambda net, t: 1.0 + 0.5 * t + (1 - torch.exp(-t)) ** 2 * net(t)]) if second else \
            (lambda D: [lambda net, t: 1.0 + (1 - torch.exp(-t)) * net(t)])
        pde = eval(src, {"torch": torch, "F": F})          # noqa: S307 -- source generated above from fixed templates
        return zoo.System(f"fuzz{seed}", 1, [(1, 1, (32, 32), "tanh")], [(0.0, 2.0)], pde, conds, enf), src
    leaves = ["u", "x", "y", "ux", "uy", "torch.round(4.0 * x)", "(0.4 + 0.0 * y)", "(x * y)", "(u * x)"]
    body = _expr(rng, leaves, 4, "x")
    lap = rng.random() < 0.5
    src = ("lambda D: (lambda u, x, y: (lambda ux, uy: ["
           + ("D(u, x, order=2) + D(u, y, order=2) + " if lap else "ux + 2.0 * uy + ")
           + f"0.3 * ({body})])(D(u, x), D(u, y)))")
    pde = eval(src, {"torch": torch, "F": F})              # noqa: S307
    return zoo.System(f"fuzz{seed}", 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde,
                      lambda: [C.NoCondition()], lambda D: [lambda net, x, y: net(zoo._cat(x, y))]), src


@pytest.mark.parametrize("seed", range(48))
def test_random_equation_on_host_matches_autograd_oracle(seed):
    from oracle import autograd_ref as R
    torch.manual_seed(100 + seed)
    system, src = _system(seed)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(48, seed=seed)
    onets, enforcers, opde = system.oracle(flat)
    want = R.closure(onets, enforcers, opde, coords)
    want_grad = R.get_flat_grad(onets).numpy()
    prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]), flat.double().numpy(),
                                                  f64=True)
    # fp64 build of the generated code against fp64 autograd: nothing but rounding may differ (the piecewise functions meet
    # their kinks on a set of measure zero; the coordinates' rounding functions are fed dyadic multiples)
    assert rel_l2(resid, want["residuals"].numpy()) < 1e-10, src
    assert abs(loss - want["loss"].item()) <= 1e-10 * abs(want["loss"].item()), src
    assert rel_l2(grad, want_grad) < 1e-9, src
