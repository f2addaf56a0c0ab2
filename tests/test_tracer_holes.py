"""The silent-wrong holes VERDICT r5 reproduced in the tracer (weak #1) and ADVICE r5's findings, as tests: every probe either
matches the fp64 autograd oracle running the same callable on real tensors (1e-10) or REFUSES (TraceUnsupported -> the loud
composite path, which is the reference's own closure).  Never a number baked into the kernel that the reference re-reads
every batch (solvers.py:380), never a diff() target mistaken for a coordinate (neurodiffeq.py:22-24).

The fuzz half extends tests/test_tracer_fuzz.py with leaves that depend on the batch size (``x.shape[0]``, ``len(x)``,
``torch.ones(x.shape[0], 1)``, ``x.new_tensor``) and with diff targets derived from a coordinate (``x + 0.0``)."""
import random

import numpy as np
import pytest
import torch

from neurodiffeq_amd import conditions as C
from neurodiffeq_amd.symbolic import TraceUnsupported
from tests import zoo
from tests.test_trace_codegen import host_closure, rel_l2
from tests.test_tracer_fuzz import BINARY, UNARY, _expr

F = torch.nn.functional


def _ng(f):
    with torch.no_grad():
        return f()


def _pde_system(name, src):
    pde = eval(src, {"torch": torch, "F": F, "np": np, "_ng": _ng})          # noqa: S307 -- fixed templates below
    return zoo.System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)], pde,
                      lambda: [C.NoCondition()], lambda D: [lambda net, x, y: net(zoo._cat(x, y))])


def _run(system, seed=0, n=48):
    from oracle import autograd_ref as R
    torch.manual_seed(100 + seed)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(n, seed=seed)
    onets, enforcers, opde = system.oracle(flat)
    # (the reference's default precision, neurodiffeq/__init__.py:22: factories inside the equations make doubles as well)
    was = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        want = R.closure(onets, enforcers, opde, coords)
        want_grad = R.get_flat_grad(onets).numpy()
        prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]),
                                                      flat.double().numpy(), f64=True)
    finally:
        torch.set_default_dtype(was)
    return (rel_l2(resid, want["residuals"].numpy()), abs(loss - want["loss"].item()) / max(abs(want["loss"].item()), 1e-300),
            rel_l2(grad, want_grad))


# (source, must it refuse?)  -- the reference evaluates all of them (solvers.py:380)
PROBES = {
    "inv_shape0": ("lambda D: lambda u, x, y: [D(u, x) * (1.0 / x.shape[0])]", False),
    "div_len": ("lambda D: lambda u, x, y: [D(u, x) + u / len(x)]", True),
    "size0_sqrt": ("lambda D: lambda u, x, y: [(D(u, x) + u) / x.size(0) ** 0.5]", False),
    "numel": ("lambda D: lambda u, x, y: [D(u, x) + u / u.numel()]", False),
    "float_shape0": ("lambda D: lambda u, x, y: [D(u, x) + u * float(x.shape[0])]", True),
    "linspace": ("lambda D: lambda u, x, y: [D(u, x) + torch.linspace(0, 1, x.shape[0]).reshape(-1, 1) * u]", True),
    "arange_len": ("lambda D: lambda u, x, y: [D(u, x) + torch.arange(len(x)).reshape(-1, 1) * u]", True),
    "arange_shape": ("lambda D: lambda u, x, y: [D(u, x) + torch.arange(x.shape[0]).reshape(-1, 1) * u]", True),
    "np_ones": ("lambda D: lambda u, x, y: [D(u, x) + torch.as_tensor(np.ones((x.shape[0], 1))) * u]", True),
    "ones_vector": ("lambda D: lambda u, x, y: [D(u, x) + torch.ones(x.shape[0]) * u]", True),
    "shape_compare_number": ("lambda D: lambda u, x, y: [D(u, x) + (u if x.shape[0] == 48 else 2.0 * u)]", True),
    "shape_eq_mask": ("lambda D: lambda u, x, y: [D(u, x) + (x.shape[0] == 48) * u + (x.shape[0] > 100) * u]", False),
    "full_of_inverse_n": ("lambda D: lambda u, x, y: [D(u, x) + torch.full((x.shape[0], 1), 1.0) / x.shape[0] * u - u * x.size(0) ** -1]", False),
    "diff_x_plus_0": ("lambda D: lambda u, x, y: [D(u, x + 0.0) + u]", True),
    "diff_x_times_1": ("lambda D: lambda u, x, y: [D(u, x * 1.0) + u]", True),
    "diff_clone": ("lambda D: lambda u, x, y: [D(u, x.clone()) + u]", True),
    "diff_view": ("lambda D: lambda u, x, y: [D(u, x.view(-1, 1)) + u]", True),
    "diff_2x": ("lambda D: lambda u, x, y: [D(u, 2.0 * x) + u]", True),
    # ... and what must stay INSIDE the traced family, now for the right reason
    "ones_shape0": ("lambda D: lambda u, x, y: [D(u, x) + 0.3 * torch.ones(x.shape[0], 1) * u]", False),
    "ones_shape": ("lambda D: lambda u, x, y: [D(u, x) + 0.3 * torch.ones(x.shape) * u + torch.zeros((x.size(0), 1))]", False),
    "full_shape": ("lambda D: lambda u, x, y: [D(u, x) + torch.full((x.shape[0], 1), 0.25) * u]", False),
    "new_tensor": ("lambda D: lambda u, x, y: [D(u, x) + x.new_tensor(0.3) * u + u.new_ones(u.shape[0], 1)]", False),
    "reshape_shape": ("lambda D: lambda u, x, y: [D(u, x).reshape(x.shape[0], 1) + u.view(-1, 1) + u.reshape(x.shape)]", False),
    "shape_compare_shape": ("lambda D: lambda u, x, y: [D(u, x) + (u if x.shape == y.shape and u.shape[1] == 1 else 2.0 * u)]", False),
    "eq_mask": ("lambda D: lambda u, x, y: [D(u, x) + torch.where(torch.round(4.0 * x) == 0.0, u, 2.0 * u) + (x != y) * u]", False),
    "sinc_by_hand": ("lambda D: lambda u, x, y: [D(u, x) + (torch.sin(x) / x).where(x != 0, torch.ones_like(x)) * u]", False),
    "eq_torch": ("lambda D: lambda u, x, y: [D(u, x) + torch.eq(torch.floor(2.0 * x), 0.0) * u + torch.ne(torch.floor(2.0 * y), 0.0) * u]", False),
    # everyday tensor idioms the reference evaluates (VERDICT r5 missing #5), traced since round 6
    "mod": ("lambda D: lambda u, x, y: [D(u, x) + (x % 0.3) * u + (u % 0.7)]", False),
    "masked_fill": ("lambda D: lambda u, x, y: [D(u, x) + u.masked_fill(x > 0.2, 0.0) + torch.masked_fill(D(u, y), y < 0.0, 2.0)]", False),
    "matmul": ("lambda D: lambda u, x, y: [torch.cat([u, D(u, x)], 1) @ torch.tensor([[1.0], [0.5]]) + "
               "(torch.cat([x, y], dim=1) @ torch.tensor([[0.5, 1.0], [2.0, -1.0]]))[:, 1:2] * u]", False),
    "nan_to_num": ("lambda D: lambda u, x, y: [D(u, x) + torch.nan_to_num(torch.sqrt(x), nan=0.25) * u]", False),
    "row_norm": ("lambda D: lambda u, x, y: [D(u, x) + torch.norm(torch.cat([D(u, x), D(u, y)], 1), dim=1, keepdim=True) "
                 "+ torch.linalg.norm(torch.cat([u, x], 1), ord=1, dim=1, keepdim=True)]", False),
    "cross": ("lambda D: lambda u, x, y: [D(u, x) + torch.cross(torch.cat([u, x, y], 1), torch.cat([y, u, x], 1), dim=1)[:, 1:2]]", False),
    # a torch.no_grad() block inside the equations: its results are constants for autograd (round 6: Sym.__init__)
    "no_grad_block": ("lambda D: lambda u, x, y: [D(u, x) + u * _ng(lambda: torch.sigmoid(3.0 * u))]", False),
    "bool_of_a_value": ("lambda D: lambda u, x, y: [D(u, x) + (torch.round(2.0 * x)).bool() * u]", False),
    "cast_to_half": ("lambda D: lambda u, x, y: [D(u, x) + u.to(torch.float16).to(u.dtype)]", True),
    "cast_to_long": ("lambda D: lambda u, x, y: [D(u, x) + (3.0 * x).long() * u]", True),
    "logit_eps": ("lambda D: lambda u, x, y: [D(u, x) + torch.logit(torch.sigmoid(3.0 * u), eps=0.2)]", False),
    # torch.bool semantics (round 6, second half): `mask + mask` is a logical OR in torch, `&` / `|` / `~` exist for bool masks
    # only, logical_* take any dtype ("non-zero is True"), a row sum of masks COUNTS
    "mask_plus_mask": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0) + (y > 0)) * u + ((x > 0) + (y > 0) + (x > y)) * u]", False),
    "not_mask_plus_mask": ("lambda D: lambda u, x, y: [D(u, x) + ((~(x > 0)) + (y > 0)) * u + ((x > 0) + True) * u]", False),
    "mask_plus_int": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0) + 1) * u + ((x > 0) / 2) * u + ((x > 0) ** 2) * u]", False),
    "mask_float_sum": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0).float() + (y > 0).float()) * u]", False),
    "mask_xor": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0) ^ (y > 0)) * u + torch.logical_xor(x > 0, y > 0) * u]", False),
    "mask_maximum": ("lambda D: lambda u, x, y: [D(u, x) + torch.maximum(x > 0, y > 0) * u + torch.minimum(x > 0, y > 0) * u]", False),
    "mask_row_count": ("lambda D: lambda u, x, y: [D(u, x) + torch.cat([x > 0, y > 0], 1).sum(dim=1, keepdim=True) * u]", False),
    "mask_where_of_masks": ("lambda D: lambda u, x, y: [D(u, x) + (torch.where(x > 0, y > 0, x > y) + (x > 0.5)) * u]", False),
    "logical_of_floats": ("lambda D: lambda u, x, y: [D(u, x) + torch.logical_and(torch.round(2 * x), torch.round(2 * y)) * u "
                          "+ torch.logical_or(torch.round(2 * x), torch.round(2 * y)) * u + torch.logical_not(torch.round(2 * x)) * u]", False),
    "mask_minus_mask": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0) - (y > 0)) * u]", "torch raises"),        # torch raises
    "minus_mask": ("lambda D: lambda u, x, y: [D(u, x) + (-(x > 0)) * u]", "torch raises"),
    "invert_a_float": ("lambda D: lambda u, x, y: [D(u, x) + (~x) * u]", "torch raises"),
    "and_of_floats": ("lambda D: lambda u, x, y: [D(u, x) + (x & y) * u]", "torch raises"),
    "or_float_mask": ("lambda D: lambda u, x, y: [D(u, x) + (x | (y > 0)) * u]", "torch raises"),
    "column_minus_mask": ("lambda D: lambda u, x, y: [D(u, x) + (u - (x > 0))]", "torch raises"),
    "one_minus_mask": ("lambda D: lambda u, x, y: [D(u, x) + (1 - (x > 0)) * u]", "torch raises"),
    "mask_minus_number": ("lambda D: lambda u, x, y: [D(u, x) + ((x > 0) - 0.5) * u]", "torch raises"),
    "abs_of_a_mask": ("lambda D: lambda u, x, y: [D(u, x) + abs(x > 0) * u]", "torch raises"),
    "where_on_a_float_condition": ("lambda D: lambda u, x, y: [D(u, x) + torch.where(x, u, y)]", "torch raises"),
    "where_on_a_float32_mask": ("lambda D: lambda u, x, y: [D(u, x) + torch.where((x > 0).float(), u, y)]", "torch raises"),
    "masked_fill_on_a_float_mask": ("lambda D: lambda u, x, y: [D(u, x) + u.masked_fill(x, 0.0)]", "torch raises"),
    "math_functions_of_masks": ("lambda D: lambda u, x, y: [D(u, x) + torch.sin(x > 0) * u + torch.exp(y > 0) * u "
                                "+ torch.asin((x > 0) * (y > 0)) * u + torch.sqrt(x > 0) * u + (1 - (x > 0).double()) * u]", False),
    "where_on_composed_masks": ("lambda D: lambda u, x, y: [D(u, x) + torch.where((x > 0) & (y > 0), u, y) + torch.where(~(x > 0), u, y) "
                                "+ u.masked_fill((x > 0) | (y < 0), 0.5) + torch.where(torch.round(2 * x).bool(), u, y)]", False),
    "coordinate_switched_off": ("lambda D: lambda u, x, y: [D(u, x.requires_grad_(False)) + u]", "torch raises"),
    # float32 values inside the fp64 build (the probes run the fp64 host pipeline): the reference rounds / computes in float32
    "float_of_a_value": ("lambda D: lambda u, x, y: [D(u, x) + u.float() * x]", True),
    "to_float32": ("lambda D: lambda u, x, y: [D(u, x) + u.to(torch.float32) * x + y.type(torch.float32) * u]", True),
    "mask_float_times_number": ("lambda D: lambda u, x, y: [D(u, x) + (x > 0).float() * 0.1 * u]", True),
    "ones_float32_times_number": ("lambda D: lambda u, x, y: [D(u, x) + 0.1 * torch.ones_like(u, dtype=torch.float32) * u]", True),
    "mask_float_times_column": ("lambda D: lambda u, x, y: [D(u, x) + (x > 0).float() * u + u * (y > 0).to(torch.float32) "
                                "+ torch.ones_like(u, dtype=torch.float32) * u + (x > 0).to(u.dtype) * u + (x > 0).type_as(u) * u]", False),
    "double_of_a_value": ("lambda D: lambda u, x, y: [D(u, x) + u.double() * x + u.to(torch.float64) * y]", False),
    # torch differentiates these two with float32 constants whatever the dtype (derivatives.yaml / hardsigmoid_backward)
    "celu_hardsigmoid": ("lambda D: lambda u, x, y: [D(u, x) + F.celu(u, alpha=0.7) + F.hardsigmoid(u) + F.celu(D(u, y)) + F.hardswish(u)]", False),
}


@pytest.mark.parametrize("name", sorted(PROBES))
def test_probe_matches_autograd_or_refuses(name):
    src, must_refuse = PROBES[name]
    system = _pde_system(name, src)
    if must_refuse == "torch raises":
        # the reference itself fails on these (bool - bool, ~float, float & float ...): the trace must not quietly compute
        # something instead -- whatever it raises sends the solver to the reference's closure, which raises torch's error
        from tests.test_trace_codegen import trace
        nets, conds, pde = system.product()
        with pytest.raises(Exception):                   # noqa: B017, PT011
            trace(nets, conds, pde, 2, f64=True)
        with pytest.raises(Exception):                   # noqa: B017, PT011
            _run(system)
        return
    if must_refuse:
        with pytest.raises((TraceUnsupported, TypeError)) as e:       # (TypeError: torch's own argument parser met the token)
            _run(system)
        if e.type is TypeError:
            assert "_BatchDim" in str(e.value) or "Sym" in str(e.value), e.value
        return
    r, l, g = _run(system)
    assert r < 1e-10 and l < 1e-10 and g < 1e-9, (name, r, l, g)


def test_batch_size_never_reaches_the_trace_as_a_number():
    """Whatever a callable does with x.shape[0] / len(x) / x.numel(): no Python number comes out of it -- arithmetic gives a
    traced value backed by a kernel ARGUMENT (Graph.nbatch: a frozen parameter the engine refills with the global batch size),
    everything that needs a number refuses."""
    from neurodiffeq_amd.symbolic import Graph, Sym, SymMat, trace_scope
    g = Graph(2)
    with trace_scope(g):
        x, y = Sym(g, g.coord(0), leaf=True), Sym(g, g.coord(1), leaf=True)
        m = SymMat([x, y])
        n = x.shape[0]
        for f in (lambda: len(x), lambda: len(m), lambda: int(n), lambda: float(n), lambda: bool(n), lambda: range(n), lambda: [0] * n,
                  lambda: np.sqrt(n), lambda: n // 2, lambda: x.size()[0] % 2, lambda: divmod(n, 2), lambda: round(n),
                  lambda: x ** n, lambda: x[:n], lambda: x[n - 1], lambda: (1 if n > 5 else 2), lambda: (1 if n == 48 else 2),
                  lambda: torch.linspace(0, 1, n), lambda: torch.arange(n), lambda: torch.rand(n, 1), lambda: torch.ones(n)):
            with pytest.raises((TraceUnsupported, TypeError)):
                f()
        node = g.nbatch()
        assert g.nodes[node][0] == "param" and g.nodes[node][1] in g.frozen          # a frozen kernel argument, no adjoint
        for f in (lambda: n + 1, lambda: 1 + n, lambda: n * 2.0, lambda: 1.0 / n, lambda: n / 2, lambda: n ** 0.5, lambda: 2 ** n,
                  lambda: -n, lambda: abs(n), lambda: x.numel() * 1.0, lambda: m.shape[0] + 0, lambda: x.size(0) - 1, lambda: x * n,
                  lambda: x / n, lambda: torch.sin(x) * n, lambda: n < 5, lambda: n >= 5, lambda: n == 5, lambda: n != 5):
            r = f()
            assert isinstance(r, Sym) and node in g.reachable([r.i]), r
        assert isinstance(m * n, SymMat)
        assert x.shape == y.shape and x.shape[1] == 1 and m.shape[1] == 2 and x.dim() == 2 and len(x.shape) == 2
        assert (n == y.shape[0]) is True and (n != m.size(0)) is False
    assert not hasattr(torch.ones, "__wrapped__") and not hasattr(torch.linspace, "__wrapped__")      # (factories restored)


def test_constants_keep_the_precision_of_the_build():
    """x.new_tensor(0.3) under the fp64 build is the double 0.3, not the fp32 one (2.8e-8 against a 1e-9 contract)."""
    from neurodiffeq_amd.symbolic import Graph, Sym, trace_scope
    for f64 in (False, True):
        g = Graph(1)
        g.f64 = f64
        with trace_scope(g):
            x = Sym(g, g.coord(0), leaf=True)
            assert x.dtype == (torch.float64 if f64 else torch.float32)
            node = (x.new_tensor(0.3) * x).i
            assert g.nodes[node][0] == "mul" and 0.3 in [g.cval(a) for a in g.nodes[node][1:]]
            node = (x.new_full((1,), 0.3) * x).i
            assert 0.3 in [g.cval(a) for a in g.nodes[node][1:]]


def test_autograd_grad_without_create_graph_is_a_constant_of_the_trace():
    from neurodiffeq_amd.symbolic import Graph, Sym, sym_diff, trace_scope
    g = Graph(1)
    with trace_scope(g):
        x = Sym(g, g.coord(0), leaf=True)
        u = torch.sin(x) * x
        d0 = torch.autograd.grad(u, x, torch.ones_like(u))[0]
        d1 = torch.autograd.grad(u, x, torch.ones_like(u), create_graph=True)[0]
        assert g.nodes[d0.i][0] == "detach" and g.nodes[d0.i][1] == d1.i
        assert sym_diff(d0 * x, x).i == d0.i and g.cval(sym_diff(d1, x).i) is None       # d/dx [sg(u') x] = sg(u')
        with pytest.raises(TraceUnsupported):
            x.grad                                            # noqa: B018 -- not a method of a traced column
        with pytest.raises(TraceUnsupported):
            torch.special.expit(x, out=torch.zeros(1))


# ------------------------------------------------------------------ fuzz: shape-dependent leaves and derived diff targets
REFUSING = ["(u * (1.0 / len(x)))", "(u * float(x.shape[0]))", "torch.linspace(0, 1, x.shape[0]).reshape(-1, 1)",
            "D(u, x + 0.0)", "D(u * x, y * 1.0)", "(u + torch.arange(x.shape[0]).reshape(-1, 1))"]
TRACING = ["(u / x.shape[0])", "(x.size(0) ** 0.5 * u)", "(u / u.numel())", "(x * (1.0 / x.shape[0]))","torch.ones(x.shape[0], 1)", "(0.3 * torch.ones(x.shape))", "x.new_tensor(0.3)", "torch.full((y.shape[0], 1), -0.5)",
           "u.reshape(x.shape[0], 1)", "((x == y) * 1.0)", "((torch.round(2.0 * x) != 0.0) * 1.0)"]


def _fuzz_system(seed):
    rng = random.Random(1000 + seed)
    leaves = ["u", "x", "y", "ux", "uy", "(x * y)"] + TRACING
    if seed % 3 == 0:
        leaves += REFUSING
    body = _expr(rng, leaves, 4, "x")
    src = ("lambda D: (lambda u, x, y: (lambda ux, uy: [ux + 2.0 * uy + "
           + f"0.3 * ({body}) + 0.0 * {rng.choice(leaves)}])(D(u, x), D(u, y)))")
    return _pde_system(f"shapefuzz{seed}", src), src


@pytest.mark.parametrize("seed", range(36))
def test_random_equation_with_shape_dependent_leaves(seed):
    system, src = _fuzz_system(seed)
    must_refuse = any(leaf in src for leaf in REFUSING)
    if must_refuse:
        with pytest.raises((TraceUnsupported, TypeError)):
            _run(system, seed)
        return
    r, l, g = _run(system, seed)
    assert r < 1e-10 and l < 1e-10 and g < 1e-9, (src, r, l, g)


def test_the_fuzzer_exercises_both_outcomes():
    srcs = [_fuzz_system(seed)[1] for seed in range(36)]
    refusing = sum(any(leaf in s for leaf in REFUSING) for s in srcs)
    shaped = sum(any(leaf in s for leaf in TRACING) and not any(leaf in s for leaf in REFUSING) for s in srcs)
    assert refusing >= 6 and shaped >= 12, (refusing, shaped)


# ---- a condition that overrides `enforce` and calls the network itself (hard constraints written by hand, conditions.py:52-55)
class _ByHand(C.BaseCondition):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def enforce(self, net, *coords):
        return self.fn(net, *coords)


BY_HAND = {
    "plain": ("lambda net, x, y: net(torch.cat([x, y], 1))", False),
    "swapped": ("lambda net, x, y: net(torch.cat((y, x), dim=-1))", False),
    "hard_constraint": ("lambda net, x, y: (1 - x ** 2) * (1 - y ** 2) * net(torch.cat([x, y], 1)) + torch.sin(3.0 * x)", False),
    "forward_spelled_out": ("lambda net, x, y: 1.0 + (1 - torch.exp(-(x + 1.0))) * net.forward(torch.cat([x, y], 1))", False),
    "detached_copy": ("lambda net, x, y: net(torch.cat([x, y], 1)).detach() * x + net(torch.cat([x, y], 1))", False),
    # the network somewhere else than at the coordinate columns: derivative streams are per coordinate -- refused
    "one_column_twice": ("lambda net, x, y: net(torch.cat([x, x], 1))", True),
    "scaled_input": ("lambda net, x, y: net(torch.cat([2.0 * x - 1.0, y], 1))", True),
    "reflected": ("lambda net, x, y: 0.5 * (net(torch.cat([x, y], 1)) + net(torch.cat([-x, y], 1)))", True),
    "detached_input": ("lambda net, x, y: net(torch.cat([x.detach(), y], 1))", True),
}


@pytest.mark.parametrize("name", sorted(BY_HAND))
def test_network_called_by_hand_inside_enforce(name):
    src, must_refuse = BY_HAND[name]
    fn = eval(src, {"torch": torch})          # noqa: S307 -- fixed templates above
    system = zoo.System(name, 2, [(2, 1, (32, 32), "tanh")], [(-1.0, 1.0), (-1.0, 1.0)],
                        lambda D: (lambda u, x, y: [D(u, x, order=2) + D(u, y, order=2) + u * D(u, x)]),
                        lambda: [_ByHand(fn)], lambda D: [fn])
    if must_refuse:
        with pytest.raises(TraceUnsupported):
            _run(system)
        return
    r, l, g = _run(system)
    assert r < 1e-10 and l < 1e-10 and g < 1e-9, (name, r, l, g)


@pytest.mark.parametrize("src,resid_nan,grad_nan", [
    # the well-known pitfall: autograd multiplies an exact 0.0 by the local derivative of the branch NOT taken
    ("lambda D: lambda u, x, y: [D(u * torch.where(x > 0, torch.sqrt(x), 0.0 * x), x)]", True, True),
    ("lambda D: lambda u, x, y: [D(u, x) + torch.where(u > 0, torch.sqrt(u), 0.0 * u)]", False, True),
    # ... and the same masks where every branch has a finite derivative stay clean
    ("lambda D: lambda u, x, y: [D(u * torch.where(x > 0, torch.sqrt(torch.where(x > 0, x, 1.0 + 0.0 * x)), 0.0 * x), x)]", False, False),
    ("lambda D: lambda u, x, y: [D(torch.relu(u) * torch.clamp(x, min=0.0), x) + torch.where(x > 0, u / (x + 2.0), u)]", False, False),
])
def test_nan_of_a_branch_not_taken_is_the_references_nan(src, resid_nan, grad_nan):
    """`torch.where(x > 0, torch.sqrt(x), 0)` differentiated where x <= 0: nan in the reference (0 * inf), so nan here -- a
    traced select that returned a clean 0 there would train on where the reference stops (Graph.diff, rule for `where`)."""
    from oracle import autograd_ref as R
    from tests.test_trace_codegen import host_closure
    system = _pde_system("nan", src)
    torch.manual_seed(100)
    nets, conds, pde = system.product()
    flat = R.get_flat(nets)
    coords = system.sample(48, seed=0)
    onets, enforcers, opde = system.oracle(flat)
    was = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        want = R.closure(onets, enforcers, opde, coords)
        wg = R.get_flat_grad(onets).numpy()
        prog, funcs, resid, loss, grad = host_closure(nets, conds, pde, np.stack([c.numpy() for c in coords]), flat.double().numpy(), f64=True)
    finally:
        torch.set_default_dtype(was)
    wr = want["residuals"].numpy()
    assert np.isnan(wr).any() == resid_nan and np.isnan(wg).any() == grad_nan          # (what the reference does)
    assert np.array_equal(np.isnan(resid), np.isnan(wr)) and np.array_equal(np.isnan(grad), np.isnan(wg))
    ok = ~np.isnan(wr)
    assert np.allclose(resid[ok], wr[ok], rtol=1e-10, atol=1e-12)
    if not grad_nan:
        assert rel_l2(grad, wg) < 1e-9
