"""Symbolic tracing of the pointwise part of the PINN hot path.

The reference evaluates ``cond.parameterize`` (conditions.py:41-57), the user's ``diff_eqs`` (solvers.py:380) and
every ``diff()`` inside them (neurodiffeq.py:21-34) as graphs of ATen ops that autograd walks ``order`` times.
Here the same Python callables are run ONCE on :class:`Sym` proxies.  A network output is an opaque function of
the coordinates whose partial derivatives are named symbols ("streams") that the fused HIP forward kernel
(csrc/ndq_mlp.h) provides; ``diff`` therefore becomes plain symbolic differentiation of a hash-consed expression
DAG, and the DAG (residuals + function values + reverse-mode adjoint w.r.t. the stream symbols) is emitted as one
fused HIP kernel by :mod:`neurodiffeq_amd.codegen`.

Every traced value has the reference's ``(N, 1)`` column semantics.
"""
import functools
import math
import numbers

import torch

__all__ = ["Sym", "SymMat", "SymScalar", "Graph", "TraceUnsupported", "is_sym", "sym_diff"]


class TraceUnsupported(Exception):
    """Raised when user code does something the tracer cannot express; the solver then uses the composite path."""


class MetricTraceUnsupported(TraceUnsupported):
    """A *metric* (solvers.py:377-379) is not a batch mean of traced per-point values (``.max()``, the square root of a
    mean, ...).  Metrics do not take part in training: the solver keeps the fused path and evaluates such metrics on
    the host from the function values the kernels write out."""


LEAVES = ("const", "coord", "net", "param", "data")

# op -> (arity).  Unary elementwise functions are listed in UNARY.
UNARY = ("neg", "sin", "cos", "tan", "exp", "log", "tanh", "sqrt", "abs", "sinh", "cosh", "sigmoid", "recip", "sign",
         "log1p", "expm1", "erf", "atan", "floor", "ceil", "round", "trunc", "detach")
# further binary nodes: atan2(a, b) and the MASKS gt(a, b) = [a > b], ge(a, b) = [a >= b] -- per-point 0.0 / 1.0 columns with
# zero derivative, what a comparison of traced columns gives (`x > 0.5`); ternary: where(m, a, b) = m != 0 ? a : b, which
# selects (it does not blend: an inf / nan in the branch not taken stays out of the value AND of the gradient, like
# torch.where).  relu / clamp / maximum / minimum are written in these (Sym.relu ...), with torch's own subgradients at ties.
BINARY = ("add", "sub", "mul", "div", "atan2", "gt", "ge")


class Graph:
    """Hash-consed expression DAG.  Node = (op, args...) with integer ids; leaves:
    ('const', value) | ('coord', i) | ('net', site, out_idx, multiindex).

    A *site* is one (network, coordinate tuple) pair the system evaluates: normally one per network, at the batch
    coordinates; conditions with Neumann ends also evaluate their network on the boundary, i.e. at a tuple in which
    one entry is a *virtual coordinate* (index >= n_coords, a constant column).  Site k < n_nets is network k's first
    site, further sites of any network follow; ``site_net[site]`` maps back to the parameter set."""

    def __init__(self, n_coords):
        self.n_coords = n_coords
        self.nodes = []
        self._ids = {}
        self._dcache = {}
        self.net_deps = {}       # site -> tuple of coordinate indices in the order fed to the net
        self.net_nout = {}       # site -> number of output units
        self.vcoords = {}        # virtual coordinate index (>= n_coords) -> its constant value
        self.site_net = []       # site -> network index (filled by register_nets / net_symbol)
        self.captured = []       # (1-element tensor baked in as a constant, its version counter at trace time)
        # trainable scalars inside the equations (nn.Parameter coefficients of inverse problems: the reference simply
        # re-runs diff_eqs under autograd, solvers.py:380) -- leaf ('param', j), read from a device vector at run time,
        # its gradient one more per-point adjoint sum -- and per-point data columns ((N, 1) tensors aligned with the
        # batch, e.g. a measured source term on a PredefinedGenerator's points) -- leaf ('data', j), one more input row
        self.params = []
        self.data = []
        # numbers that enter the trace from OUTSIDE it (Python floats read from a closure / dict / attribute, 1-element
        # tensors baked in as constants), in the order the callables hand them over.  The reference re-evaluates the user's
        # callables every batch (solvers.py:380), so such a number may differ from one epoch to the next -- a viscosity ramp, a
        # curriculum on ``solver.local_epoch``; as a literal of the generated kernel every new value would be a new hipcc run.
        # Positions listed in ``volatile`` become RUNTIME constants instead: leaves ('param', j) backed by a frozen host
        # scalar (``frozen``: their indices; no adjoint, no optimiser), read from the same device vector as the trainable
        # scalars.  Which positions are volatile is found out by the solver when a re-trace differs from the compiled
        # one in nothing but such numbers (engine.trace_system: eq_probe / suggest_volatile); any choice of positions gives
        # a faithful program -- the set only decides what is a literal and what an argument.
        self.ext_log = []
        self.volatile = frozenset()
        self.frozen = set()
        self._rtensors = {}      # external position -> its frozen scalar

    def external(self, v):
        """Node of a number coming from outside the trace (see ``ext_log`` above)."""
        v = float(v)
        pos = len(self.ext_log)
        self.ext_log.append(v)
        if pos in self.volatile:
            t = self._rtensors.get(pos)
            if t is None:
                t = self._rtensors[pos] = torch.zeros((), dtype=torch.float64)
            t.fill_(v)
            i = self.param(t)
            self.frozen.add(self.nodes[i][1])
            return i
        return self.const(v)

    # -------------------------------------------------------------- networks
    def register_nets(self, nets, n_outs, skips=None):
        """skips[k]: the bias-free ``nn.Linear`` skip connection of network k when it is handled SYMBOLICALLY (networks wider
        than 64 units: the kernels of csrc/ndq_wide.h / ndq_deep.h serve plain FCNNs, so ``Resnet``'s ``out += S x``
        (networks.py:73-106) is added here, every entry of S a trainable kernel argument -- ``param_elem``), else None."""
        self._net_ids = {id(n): k for k, n in enumerate(nets)}
        self._net_nouts = list(n_outs)
        self._net_skips = list(skips) if skips is not None else [None] * len(nets)
        self.site_net = list(range(len(nets)))
        self._sites = {}         # (network index, deps) -> site

    def vcoord(self, value):
        """A new virtual coordinate: a column that holds ``value`` at every point (node id)."""
        idx = self.n_coords + len(self.vcoords)
        self.vcoords[idx] = float(value)
        return self.coord(idx)

    def net_symbol(self, net, coords, ith_unit=None):
        """Symbol for ``net(cat(coords, 1))`` (conditions.py:52-55) -- only for the solver's own networks, evaluated at
        (real or virtual) coordinate columns."""
        k = getattr(self, "_net_ids", {}).get(id(net))
        if k is None:
            raise TraceUnsupported("a network that does not belong to the solver was called inside the traced region")
        deps = []
        for c in coords:
            node = self.nodes[c.i] if isinstance(c, Sym) else None
            if node is None or node[0] != "coord":
                raise TraceUnsupported("network evaluated at something other than coordinate columns")
            deps.append(node[1])
        deps = tuple(deps)
        if len(set(deps)) != len(deps):
            # net(cat([x, x], 1)): d/dx is the SUM of two input derivatives -- the stream sets are per coordinate
            raise TraceUnsupported("network evaluated with one coordinate column in two of its inputs")
        site = self._sites.get((k, deps))
        if site is None:
            if k not in self.net_deps:           # the network's first site keeps the network's own index
                site = k
            else:
                site = len(self.site_net)
                self.site_net.append(k)
            self._sites[(k, deps)] = site
            self.net_deps[site] = deps
        n_out = self._net_nouts[k]
        self.net_nout[site] = n_out
        skip = self._net_skips[k]
        k = site

        def output(o):
            u = Sym(self, self.net(k, o))
            if skip is not None:                 # + sum_a S[o][a] x_a: derivatives w.r.t. the coordinates come out of diff()
                d = len(coords)
                for a, c in enumerate(coords):
                    u = u + Sym(self, self.param_elem(skip.weight, o * d + a)) * c
            return u
        if ith_unit is not None:
            return output(ith_unit)
        if n_out == 1:
            return output(0)
        return SymMat([output(o) for o in range(n_out)])

    # -------------------------------------------------------------- construction
    def _mk(self, key):
        i = self._ids.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self._ids[key] = i
        return i

    def const(self, v):
        v = float(v)
        if v == 0.0:
            v = 0.0  # normalise -0.0
        return self._mk(("const", v))

    def coord(self, i):
        return self._mk(("coord", int(i)))

    def net(self, net_idx, out_idx, mi=()):
        mi = tuple(mi)
        if not (mi and mi[0] == "L"):       # ("L", a, b, ..) = Laplacian stream: sum over a of d2/dx_a^2
            mi = tuple(sorted(mi))
        return self._mk(("net", int(net_idx), int(out_idx), mi))

    def param(self, t):
        """Leaf for a trainable 1-element tensor (one leaf per tensor object)."""
        for j, p in enumerate(self.params):
            if p is t:
                return self._mk(("param", j))
        self.params.append(t)
        return self._mk(("param", len(self.params) - 1))

    def param_elem(self, t, k):
        """Leaf for ENTRY k (flat index) of a trainable tensor: ``params`` holds the pair (tensor, k); read from the same device
        vector as the scalars, its gradient lands in ``t.grad`` at k (engine.attach_theta_grads)."""
        for j, p in enumerate(self.params):
            if isinstance(p, tuple) and p[0] is t and p[1] == k:
                return self._mk(("param", j))
        self.params.append((t, int(k)))
        return self._mk(("param", len(self.params) - 1))

    def nbatch(self):
        """Leaf for the GLOBAL batch size as a number of the equations (x.shape[0] in arithmetic: `_BatchDim`): a frozen
        kernel argument like the runtime constants above, refilled by the engine before every launch sequence
        (engine.FusedSystem.set_batch_size) -- one program still serves every batch size."""
        t = getattr(self, "_nbatch_t", None)
        if t is None:
            t = self._nbatch_t = torch.zeros((), dtype=torch.float64)
        i = self.param(t)
        self.frozen.add(self.nodes[i][1])
        return i

    def datacol(self, t):
        """Leaf for an (N, 1) column of per-point data (one leaf per tensor object)."""
        for j, p in enumerate(self.data):
            if p is t:
                return self._mk(("data", j))
        self.data.append(t)
        return self._mk(("data", len(self.data) - 1))

    def cval(self, i):
        n = self.nodes[i]
        return n[1] if n[0] == "const" else None

    def add(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(ca + cb)
        if ca == 0.0:
            return b
        if cb == 0.0:
            return a
        if a > b:
            a, b = b, a
        return self._mk(("add", a, b))

    def sub(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(ca - cb)
        if cb == 0.0:
            return a
        if ca == 0.0:
            return self.unary("neg", b)
        if a == b:
            return self.const(0.0)
        return self._mk(("sub", a, b))

    def mul(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(ca * cb)
        if ca == 0.0 or cb == 0.0:
            return self.const(0.0)
        if ca == 1.0:
            return b
        if cb == 1.0:
            return a
        if ca == -1.0:
            return self.unary("neg", b)
        if cb == -1.0:
            return self.unary("neg", a)
        if a > b:
            a, b = b, a
        return self._mk(("mul", a, b))

    def div(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(ca / cb)
        if ca == 0.0:
            return self.const(0.0)
        if cb == 1.0:
            return a
        return self._mk(("div", a, b))

    def powi(self, a, n):
        n = int(n)
        if n == 0:
            return self.const(1.0)
        if n == 1:
            return a
        ca = self.cval(a)
        if ca is not None:
            return self.const(ca ** n)
        if n < 0:
            return self.div(self.const(1.0), self.powi(a, -n))
        return self._mk(("powi", a, n))

    def powc(self, a, c):
        c = float(c)
        if c == int(c) and abs(c) <= 64:
            return self.powi(a, int(c))
        ca = self.cval(a)
        if ca is not None:
            return self.const(ca ** c)
        if c == 0.5:
            return self.unary("sqrt", a)
        return self._mk(("powc", a, c))

    _FOLD = {
        "neg": lambda v: -v, "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "log": math.log,
        "tanh": math.tanh, "sqrt": math.sqrt, "abs": abs, "sinh": math.sinh, "cosh": math.cosh,
        "sigmoid": lambda v: 1.0 / (1.0 + math.exp(-v)), "recip": lambda v: 1.0 / v,
        "sign": lambda v: (v > 0) - (v < 0),
        "log1p": math.log1p, "expm1": math.expm1, "erf": math.erf, "atan": math.atan,
        "floor": lambda v: float(math.floor(v)), "ceil": lambda v: float(math.ceil(v)), "round": lambda v: float(round(v)),   # (half to even, like torch.round)
        "trunc": lambda v: float(math.trunc(v)),
        "detach": lambda v: v,
    }

    def unary(self, op, a):
        ca = self.cval(a)
        if ca is not None:
            return self.const(self._FOLD[op](ca))
        if op == "neg" and self.nodes[a][0] == "neg":
            return self.nodes[a][1]
        return self._mk((op, a))

    def atan2(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(math.atan2(ca, cb))
        return self._mk(("atan2", a, b))

    def gt(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(1.0 if ca > cb else 0.0)
        if a == b:
            return self.const(0.0)
        return self._mk(("gt", a, b))

    def ge(self, a, b):
        ca, cb = self.cval(a), self.cval(b)
        if ca is not None and cb is not None:
            return self.const(1.0 if ca >= cb else 0.0)
        return self._mk(("ge", a, b))

    def where(self, m, a, b):
        cm = self.cval(m)
        if cm is not None:
            return a if cm != 0.0 else b
        if a == b:
            return a
        return self._mk(("where", m, a, b))

    # -------------------------------------------------------------- differentiation
    def diff(self, e, ci):
        """d node e / d coordinate ci (symbolic, memoised)."""
        key = (e, ci)
        r = self._dcache.get(key)
        if r is not None:
            return r
        n = self.nodes[e]
        op = n[0]
        D = lambda x: self.diff(x, ci)
        if op == "const":
            r = self.const(0.0)
        elif isinstance(ci, tuple) and op in LEAVES:   # d/d(leaf node ("n", id)): other leaves independent
            r = self.const(1.0 if e == ci[1] else 0.0)
        elif op == "coord":
            r = self.const(1.0 if n[1] == ci else 0.0)
        elif op in ("param", "data"):
            # no autograd path to a coordinate: the reference's diff() of such a tensor contributes nothing either
            r = self.const(0.0)
        elif op == "net":
            _, k, o, mi = n
            if ci not in self.net_deps[k]:
                r = self.const(0.0)
            else:
                if len(mi) >= 4:
                    raise TraceUnsupported("derivatives of network outputs beyond fourth order are outside the fused path")
                r = self.net(k, o, mi + (ci,))
        elif op == "add":
            r = self.add(D(n[1]), D(n[2]))
        elif op == "sub":
            r = self.sub(D(n[1]), D(n[2]))
        elif op == "mul":
            a, b = n[1], n[2]
            r = self.add(self.mul(D(a), b), self.mul(a, D(b)))
        elif op == "div":
            a, b = n[1], n[2]
            # (a/b)' = a'/b - (a/b) b'/b
            r = self.sub(self.div(D(a), b), self.mul(e, self.div(D(b), b)))
        elif op in ("gt", "ge"):
            r = self.const(0.0)              # a mask: piecewise constant
        elif op == "where":
            # m a' + (1 - m) b', not where(m, a', b'): autograd sends an exact 0.0 into the branch that was not taken and MULTIPLIES
            # it by that branch's local derivative -- 0 * inf = nan.  `torch.where(x > 0, torch.sqrt(x), 0.0)` differentiated with
            # respect to x is nan at x <= 0 in the reference (the well-known pitfall); a select would hand back a clean 0 there
            m, da, db = n[1], D(n[2]), D(n[3])
            if self.cval(da) == 0.0 and self.cval(db) == 0.0:
                r = self.const(0.0)
            else:
                r = self.add(self.mul(m, da), self.mul(self.sub(self.const(1.0), m), db))
        elif op == "atan2":
            a, b = n[1], n[2]                # d atan2(a, b) = (b da - a db) / (a^2 + b^2)
            r = self.div(self.sub(self.mul(b, D(a)), self.mul(a, D(b))), self.add(self.mul(a, a), self.mul(b, b)))
        elif op == "powi":
            a, k = n[1], n[2]
            r = self.mul(self.mul(self.const(k), self.powi(a, k - 1)), D(a))
        elif op == "powc":
            a, c = n[1], n[2]
            r = self.mul(self.mul(self.const(c), self.powc(a, c - 1.0)), D(a))
        else:
            a = n[1]
            da = D(a)
            if op == "neg":
                r = self.unary("neg", da)
            elif op == "sin":
                r = self.mul(self.unary("cos", a), da)
            elif op == "cos":
                r = self.unary("neg", self.mul(self.unary("sin", a), da))
            elif op == "tan":
                r = self.mul(self.add(self.const(1.0), self.mul(e, e)), da)
            elif op == "exp":
                r = self.mul(e, da)
            elif op == "log":
                r = self.div(da, a)
            elif op == "tanh":
                r = self.mul(self.sub(self.const(1.0), self.mul(e, e)), da)
            elif op == "sqrt":
                r = self.div(da, self.mul(self.const(2.0), e))
            elif op == "abs":
                r = self.mul(self.unary("sign", a), da)
            elif op in ("sign", "floor", "ceil", "round", "trunc"):
                r = self.const(0.0)          # piecewise constant (torch: zero gradient)
            elif op == "detach":
                r = self.const(0.0)          # a constant as far as autograd is concerned (x.detach() inside a product / sum)
            elif op == "sinh":
                r = self.mul(self.unary("cosh", a), da)
            elif op == "cosh":
                r = self.mul(self.unary("sinh", a), da)
            elif op == "sigmoid":
                r = self.mul(self.mul(e, self.sub(self.const(1.0), e)), da)
            elif op == "recip":
                r = self.unary("neg", self.mul(self.mul(e, e), da))
            elif op == "log1p":
                r = self.div(da, self.add(self.const(1.0), a))
            elif op == "expm1":
                r = self.mul(self.add(e, self.const(1.0)), da)
            elif op == "erf":                # 2 / sqrt(pi) * exp(-a^2)
                r = self.mul(self.mul(self.const(2.0 / math.sqrt(math.pi)), self.unary("exp", self.unary("neg", self.mul(a, a)))), da)
            elif op == "atan":
                r = self.div(da, self.add(self.const(1.0), self.mul(a, a)))
            else:  # pragma: no cover
                raise TraceUnsupported(f"no derivative rule for {op}")
        self._dcache[key] = r
        return r

    def subst(self, e, mapping, _memo=None):
        """Rebuild node e with leaves replaced according to mapping {leaf node id: node id} (simplifying on the way)."""
        memo = {} if _memo is None else _memo
        if e in mapping:
            return mapping[e]
        r = memo.get(e)
        if r is not None:
            return r
        n = self.nodes[e]
        op = n[0]
        S = lambda x: self.subst(x, mapping, memo)
        if op in LEAVES:
            r = e
        elif op in BINARY:
            r = getattr(self, op)(S(n[1]), S(n[2]))
        elif op == "where":
            r = self.where(S(n[1]), S(n[2]), S(n[3]))
        elif op == "powi":
            r = self.powi(S(n[1]), n[2])
        elif op == "powc":
            r = self.powc(S(n[1]), n[2])
        else:
            r = self.unary(op, S(n[1]))
        memo[e] = r
        return r

    # -------------------------------------------------------------- queries
    def reachable(self, roots):
        seen, order = set(), []
        stack = [(r, False) for r in roots]
        while stack:
            i, done = stack.pop()
            if done:
                order.append(i)
                continue
            if i in seen:
                continue
            seen.add(i)
            stack.append((i, True))
            n = self.nodes[i]
            if n[0] in LEAVES:
                continue
            for a in self.children(i):
                if a not in seen:
                    stack.append((a, False))
        return order  # topological (children first)

    def children(self, i):
        n = self.nodes[i]
        op = n[0]
        if op in LEAVES:
            return ()
        if op in BINARY:
            return (n[1], n[2])
        if op == "where":
            return (n[1], n[2], n[3])
        return (n[1],)


_CURRENT = []


def current_graph():
    if not _CURRENT:
        raise TraceUnsupported("no active trace")
    return _CURRENT[-1]


class trace_scope:
    def __init__(self, graph):
        self.graph = graph
        self.mode = None

    def __enter__(self):
        _CURRENT.append(self.graph)
        self._grad = torch.enable_grad()           # (a deterministic autograd mode for every trace and re-trace: Sym.__init__)
        self._grad.__enter__()
        self.mode = _TwinMode(self.graph)
        self.mode.__enter__()
        # torch's factories parse their size arguments BEFORE any __torch_function__ mode is asked, so a _BatchDim token
        # (x.shape[0]) would end in a TypeError of the argument parser: for the duration of the trace the module attributes
        # are wrappers that look for the token first (single-threaded: a trace runs to completion inside this scope)
        self._saved = {}
        if len(_CURRENT) == 1:
            for name in _FACTORIES:
                fn = getattr(torch, name, None)
                if fn is not None:
                    self._saved[name] = fn
                    setattr(torch, name, _factory_wrapper(name, fn, self.mode))
        return self.graph

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(torch, name, fn)
        self.mode.__exit__(*exc)
        self._grad.__exit__(*exc)
        _CURRENT.pop()


_FACTORIES = ("ones", "zeros", "full", "empty", "rand", "randn", "randint", "linspace", "logspace", "arange", "eye", "randperm")


def _factory_wrapper(name, fn, mode):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if any(_has_batch_dim(a) for a in args) or any(_has_batch_dim(v) for v in kwargs.values()):
            return mode._with_batch_dim(name, args, kwargs)
        return fn(*args, **kwargs)
    return wrapper


_TWIN_BINARY = {"mul": "mul", "__mul__": "mul", "multiply": "mul", "add": "add", "__add__": "add", "sub": "sub",
                "__sub__": "sub", "subtract": "sub", "div": "div", "__truediv__": "div", "divide": "div",
                "true_divide": "div"}
_TWIN_REVERSED = {"__rmul__": "mul", "__radd__": "add", "__rsub__": "sub", "__rtruediv__": "div"}
_TWIN_UNARY = {"neg": "neg", "__neg__": "neg", "negative": "neg", "exp": "exp", "log": "log", "sin": "sin", "cos": "cos",
               "tan": "tan", "tanh": "tanh", "sqrt": "sqrt", "abs": "abs", "__abs__": "abs", "sigmoid": "sigmoid",
               "sinh": "sinh", "cosh": "cosh", "reciprocal": "recip"}
_TWIN_SHAPE = ("reshape", "view", "view_as", "unsqueeze", "squeeze", "expand", "expand_as", "clone", "contiguous", "to",
               "float", "double", "type", "flatten", "__getitem__")


class _TwinMode(torch.overrides.TorchFunctionMode):
    """Active while a system is traced.  An equation may combine a TRAINABLE scalar with plain tensors before any traced
    column is involved -- ``nu * f_measured``, ``torch.exp(log_k)`` -- and what the traced column then meets is an ordinary
    tensor with an autograd history, computed once, at trace time.  This mode watches the torch calls of the traced region:
    every result that depends on a trainable leaf gets a symbolic *twin* (built from the twins / leaves of its operands),
    and ``_as_node`` hands out the twin instead of refusing the tensor.  Calls it cannot mirror leave no twin: the tensor
    is then refused as before (TraceUnsupported -> composite path), never baked in as a constant."""

    def __init__(self, graph):
        super().__init__()
        self.g = graph
        if not hasattr(graph, "twins"):
            graph.twins = {}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if any(_has_batch_dim(a) for a in args) or any(_has_batch_dim(v) for v in kwargs.values()):
            return self._with_batch_dim(getattr(func, "__name__", ""), args, kwargs)
        out = func(*args, **kwargs)
        name = getattr(func, "__name__", "")
        if isinstance(out, torch.Tensor) and id(out) in self.g.twins and self.g.twins[id(out)][0] is out:
            # the call returned a tensor that already has a twin: an in-place update (a += 1.0, a.mul_(k), a[0] = v).
            # Its twin describes the OLD value -- drop it, the tensor is refused from here on (TraceUnsupported ->
            # composite path) instead of training a different equation
            if out._version != self.g.twins[id(out)][2] or name.endswith("_") or name.startswith("__i"):
                self.g.twins[id(out)] = (out, None, -1)
            return out
        if isinstance(out, torch.Tensor) and out.requires_grad and out.grad_fn is not None and id(out) not in self.g.twins:
            try:
                node = self._mirror(name, args, kwargs, out)
            except TraceUnsupported:
                node = None
            if node is not None:
                # (the tensor is kept alive: ids are not reused; its version counter tells an in-place edit later on)
                self.g.twins[id(out)] = (out, node, out._version)
        return out

    def _with_batch_dim(self, name, args, kwargs):
        """A torch call that takes the batch size of a traced column (_BatchDim) as an argument.  A constant-fill factory
        of shape (N, 1) is a per-point constant: ``torch.ones(x.shape[0], 1)``, ``torch.full((len_x, 1), 0.3)``,
        ``c.expand(x.shape[0], 1)`` of a one-element tensor.  Everything else -- ``linspace`` / ``arange`` / ``rand`` over
        the batch, an (N,) vector, sizes computed from N -- refuses."""
        g = self.g
        size = kwargs.get("size", None)
        if name in ("ones", "zeros") and (_per_point_size(args) or (not args and size is not None and _per_point_size((size,)))):
            return Sym(g, g.const(1.0 if name == "ones" else 0.0))
        if name == "full":
            sz = args[0] if args else size
            fill = args[1] if len(args) > 1 else kwargs.get("fill_value")
            if sz is not None and _per_point_size((sz,)) and isinstance(fill, (numbers.Number, torch.Tensor, Sym, _BatchDim)):
                return Sym(g, _as_node(g, fill))
        if name in ("expand", "repeat", "tile", "broadcast_to") and args and isinstance(args[0], torch.Tensor) and args[0].numel() == 1 \
                and _per_point_size(args[1:]):
            return Sym(g, self._operand(args[0]))
        raise TraceUnsupported(f"torch.{name} with the batch size of a traced column among its arguments (only constant (N, 1) "
                               "columns -- torch.ones / zeros / full, a one-element tensor expanded -- can be traced)")

    def _operand(self, v):
        g = self.g
        if isinstance(v, torch.Tensor):
            tw = g.twins.get(id(v))
            if tw is not None and tw[0] is v and tw[1] is not None and tw[2] == v._version:
                return tw[1]
            if v.requires_grad and not (v.is_leaf and v.numel() == 1):
                raise TraceUnsupported("untracked tensor with an autograd history")
        return _as_node(g, v)

    def _mirror(self, name, args, kwargs, out=None):
        g = self.g
        if name in _TWIN_BINARY and len(args) >= 2 and not kwargs:
            return getattr(g, _TWIN_BINARY[name])(self._operand(args[0]), self._operand(args[1]))
        if name in _TWIN_REVERSED and len(args) == 2:
            return getattr(g, _TWIN_REVERSED[name])(self._operand(args[1]), self._operand(args[0]))
        if name in _TWIN_UNARY and len(args) == 1:
            return g.unary(_TWIN_UNARY[name], self._operand(args[0]))
        if name in ("pow", "__pow__") and len(args) == 2 and isinstance(args[1], numbers.Number):
            return g.powc(self._operand(args[0]), float(args[1]))
        if name == "square" and len(args) == 1:
            a = self._operand(args[0])
            return g.mul(a, a)
        if name in _TWIN_SHAPE and args and isinstance(args[0], torch.Tensor):
            if name == "__getitem__" and (out is None or out.numel() != args[0].numel()):
                return None                                  # b[0:4] is not "the column b": only shape-preserving indexing is mirrored
            return self._operand(args[0])                    # a scalar stays a scalar, a column a column
        return None


def _row_values(v):
    """A constant row vector (tensor / ndarray / list with more than one element) as a list of floats, else None."""
    if isinstance(v, torch.Tensor) and v.numel() > 1:
        if v.requires_grad:
            tw = getattr(_CURRENT[-1], "twins", {}).get(id(v)) if _CURRENT else None
            if tw is not None and tw[0] is v and tw[1] is not None and tw[2] == v._version:
                return None          # torch expression of trainable scalars and data columns: _as_node returns its twin
            raise TraceUnsupported("a trainable tensor of more than one element inside the equations (trainable SCALARS are "
                                   "traced; a vector would need one kernel argument per entry)")
        if v.dim() == 2 and v.shape[1] == 1:
            return None              # an (N, 1) column of per-point data: _as_node makes it an input row of the kernel
        if not (v.dim() == 1 or (v.dim() == 2 and v.shape[0] == 1)):
            raise TraceUnsupported(f"a concrete tensor of shape {tuple(v.shape)} inside the equations (scalars, constant "
                                   "rows (k,) / (1, k) and per-point columns (N, 1) can be traced)")
        return [float(x) for x in v.detach().reshape(-1).tolist()]
    try:
        import numpy as np
        if isinstance(v, np.ndarray) and v.size > 1:
            return [float(x) for x in v.reshape(-1)]
    except Exception:  # pragma: no cover
        pass
    if isinstance(v, (list, tuple)) and len(v) > 1 and all(isinstance(x, numbers.Number) for x in v):
        return [float(x) for x in v]
    return None


def _as_node(g, v, literal=False):
    """literal: the number is needed as a compile-time constant (an exponent): never a runtime constant (Graph.external)."""
    ext = g.const if literal else g.external
    if isinstance(v, Sym):
        if v.g is not g:
            raise TraceUnsupported("mixing symbols of different traces")
        return v.i
    if isinstance(v, _BatchDim):
        if v.g is not g:
            raise TraceUnsupported("mixing symbols of different traces")
        if literal:
            v._refuse()                  # (an exponent must be a compile-time constant)
        return g.nbatch()
    if isinstance(v, torch.Tensor) and v.requires_grad and not v.is_leaf:
        tw = getattr(g, "twins", {}).get(id(v))          # a torch expression of trainable scalars seen by _TwinMode
        if tw is not None and tw[0] is v and tw[1] is not None and tw[2] == v._version:
            return tw[1]
    if isinstance(v, numbers.Number):
        return ext(v)
    if isinstance(v, torch.Tensor) and v.numel() > 1 and v.dim() == 2 and v.shape[1] == 1 and not v.requires_grad:
        return g.datacol(v)
    if isinstance(v, torch.Tensor) and v.numel() == 1:
        if v.requires_grad:
            if not v.is_leaf:
                raise TraceUnsupported("a non-leaf tensor that requires grad inside the equations (only leaf scalars -- "
                                       "nn.Parameter coefficients -- become trainable kernel arguments)")
            return g.param(v)
        # baked in at trace time like a Python float; the solver re-traces when the tensor is modified in place
        val = v.item()
        g.captured.append((v, v._version, val))
        return ext(val)
    try:
        import numpy as np
        if isinstance(v, np.ndarray) and v.size == 1:
            return ext(v.reshape(-1)[0])
    except Exception:  # pragma: no cover
        pass
    raise TraceUnsupported(f"cannot mix a traced value with {type(v).__name__} of more than one element")


class _BatchDim:
    """``x.shape[0]`` / ``x.size(0)`` / ``x.numel()`` of a traced column: the batch size.  The trace does not have it as a
    NUMBER -- one generated kernel serves every batch the solver hands over (train and validation generators of different
    sizes, ``n_batches`` changed by a callback, a shard of the global batch under data parallelism) -- so it is handed out as
    an opaque token.  It may go back into a SHAPE (``torch.ones(x.shape[0], 1)``, ``u.reshape(x.shape[0], 1)``, ``x.shape ==
    y.shape``), where only "one value per point" matters, and into ARITHMETIC with traced columns / numbers
    (``u / x.shape[0]``, ``1.0 / x.size(0)``, ``x.shape[0] ** 0.5``), where it becomes a kernel ARGUMENT holding the global
    batch size (Graph.nbatch).  What needs a Python number -- ``float()`` / ``int()`` / ``len(x)``, an index, ``range``,
    ``linspace`` / ``arange`` over the batch, a branch on a comparison -- raises TraceUnsupported -> the (loud) composite
    path: never a number baked into the kernel that the reference re-reads every batch (solvers.py:380)."""
    __slots__ = ("g",)

    def __init__(self, g):
        self.g = g

    def _refuse(self, *a, **k):
        raise TraceUnsupported("the batch size (x.shape[0], len(x), x.size(0), x.numel()) is needed as a Python number inside "
                               "the traced region: the fused kernels are compiled for every batch size at once (arithmetic "
                               "with traced columns is fine: it becomes a kernel argument)")

    def _sym(self):
        return Sym(self.g, self.g.nbatch())

    __index__ = __int__ = __float__ = __bool__ = __len__ = __iter__ = _refuse
    __floordiv__ = __rfloordiv__ = __mod__ = __rmod__ = __divmod__ = __rdivmod__ = __round__ = __trunc__ = _refuse
    __array__ = _refuse

    def __add__(self, o): return self._sym() + o
    def __radd__(self, o): return o + self._sym()
    def __sub__(self, o): return self._sym() - o
    def __rsub__(self, o): return o - self._sym()
    def __mul__(self, o): return self._sym() * o
    def __rmul__(self, o): return o * self._sym()
    def __truediv__(self, o): return self._sym() / o
    def __rtruediv__(self, o): return o / self._sym()
    def __pow__(self, o): return self._sym() ** o
    def __rpow__(self, o): return o ** self._sym()
    def __neg__(self): return -self._sym()
    def __pos__(self): return self._sym()
    def __abs__(self): return self._sym()
    # comparisons with a number give per-point masks (a branch on one -- `if n > 100:` -- raises in Sym.__bool__)
    def __lt__(self, o): return self._sym() < o
    def __le__(self, o): return self._sym() <= o
    def __gt__(self, o): return self._sym() > o
    def __ge__(self, o): return self._sym() >= o

    def __eq__(self, other):
        if isinstance(other, _BatchDim):
            return other.g is self.g
        return self._sym() == other

    def __ne__(self, other):
        if isinstance(other, _BatchDim):
            return other.g is not self.g
        return self._sym() != other

    def __hash__(self):
        return id(self.g)

    def __repr__(self):
        return "<batch size>"


class _Shape(tuple):
    """Shape of a traced column / matrix: ``(<batch size>, k)``."""

    def numel(self):
        return self[0]


def _has_batch_dim(v):
    return isinstance(v, _BatchDim) or (isinstance(v, (tuple, list)) and any(_has_batch_dim(x) for x in v))


def _batch_dim(g):
    bd = getattr(g, "_bdim", None)
    if bd is None:
        bd = g._bdim = _BatchDim(g)
    return bd


def _per_point_size(size):
    """Is ``size`` -- the arguments of a factory / reshape -- "(batch size, 1)", i.e. one value per point?"""
    if len(size) == 1 and isinstance(size[0], (tuple, list)):
        size = tuple(size[0])
    return len(size) == 2 and isinstance(size[0], _BatchDim) and isinstance(size[1], int) and size[1] == 1

def _const_size(size):
    """Size of a constant made from a traced column (u.new_ones(1), u.new_zeros(1, 1)): one element -- it enters the trace as a
    number; a per-point constant column is torch.ones_like(u)."""
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    if _per_point_size(size):
        return (1, 1)            # u.new_ones(u.shape[0], 1): the same value at every point -- a number of the trace as well
    n = 1
    for k in size:
        n *= int(k)
    if n != 1:
        raise TraceUnsupported("a constant of more than one element made from a traced column (use torch.ones_like / zeros_like)")
    return tuple(int(k) for k in size)


def _no_such_method(kind):
    def __getattr__(self, name):
        if name.startswith("__") or name in ("g", "i", "cols", "term", "items"):
            raise AttributeError(name)
        # x.f(...) is torch.f(x, ...) for the elementwise functions of the table below (x.mul(y), x.floor(), x.softplus ...)
        h = _TORCH_FUNCS.get(name[:-1] if name.endswith("_") and not name.endswith("__") else name) if kind == "column" else None
        if h is not None and name not in _NOT_METHODS:
            if name.endswith("_"):
                raise TraceUnsupported(f"in-place Tensor.{name} on a traced {kind}")
            return functools.partial(h, self)
        raise TraceUnsupported(f"Tensor.{name} is not supported on a traced {kind}")
    return __getattr__


_NOT_METHODS = frozenset({"cat", "concat", "stack", "ones_like", "zeros_like", "full_like", "mse_loss", "l1_loss", "where",
                          "sum", "mean", "max", "min", "grad", "mm", "linalg_norm", "linalg_vector_norm", "linalg_cross", "special_expit", "special_erf", "special_erfc", "special_log1p",
                          "special_expm1", "special_sinc", "special_exp2", "special_xlogy", "special_xlog1py", "special_logit",
                          "special_round", "_threshold", "threshold", "log_sigmoid", "logsigmoid", "softplus", "elu", "selu",
                          "celu", "gelu", "silu", "mish", "softsign", "hardtanh", "relu6", "hardsigmoid", "hardswish",
                          "leaky_relu", "tanhshrink", "expit"})



class Sym:
    """Proxy for an ``(N, 1)`` column of the batch inside a trace."""
    __array_priority__ = 10000
    __array_ufunc__ = None

    def __init__(self, g, i, leaf=False):
        # leaf: this object IS one of the coordinate columns the tracer handed to the user's callables (the tensors the
        # reference samples and differentiates with respect to, neurodiffeq.py:22).  Everything an operation returns is a
        # NEW tensor in torch -- also when the arithmetic folds it back to the same node (x + 0.0, x * 1.0, x.clone(),
        # x.view(-1, 1)) -- and `diff(u, <new tensor>)` is not `diff(u, x)` (sym_diff)
        if not leaf and not torch.is_grad_enabled() and g.nodes[i][0] not in ("const", "detach"):
            # the operation ran inside the user's `with torch.no_grad():` (or set_grad_enabled(False) / inference_mode): torch
            # records no graph for its result -- a constant as far as autograd is concerned, exactly what x.detach() is (the
            # tracer itself runs the callables under torch.enable_grad(): trace_scope)
            i = g.unary("detach", i)
        self.g, self.i, self.leaf = g, i, leaf

    # ---- tensor-ish surface used by reference-style code
    @property
    def shape(self):
        return _Shape((_batch_dim(self.g), 1))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def numel(self):
        return _batch_dim(self.g)

    nelement = numel

    @property
    def ndim(self):
        return 2

    @property
    def requires_grad(self):
        return True

    def requires_grad_(self, flag=True):
        if not flag:      # on a coordinate this switches the leaf off IN PLACE (diff(u, x) raises in the reference afterwards)
            raise TraceUnsupported("Tensor.requires_grad_(False) inside the traced region")
        return self

    # ---- dtype of the value in torch's terms, as far as it changes what the arithmetic MEANS
    # isbool: a torch.bool tensor (comparisons, ~ & | ^ of them, logical_*, .bool()): `a + b` of two of those is a logical OR
    # in torch, `a * b` an AND, `a - b` / `-a` raise -- not the arithmetic of the 0.0 / 1.0 columns the trace represents them by.
    isbool = False

    def _as_bool(self):
        self.isbool = True
        return self

    def _plain(self):
        """The same values as an ordinary floating-point column (a NEW object: the flags belong to the object)."""
        return Sym(self.g, self.i)

    def _narrowed(self, what):
        """`.float()` / `.to(torch.float32)` / `.half()` ...: the precision of the build in the fp32 build; in the fp64 build the
        reference ROUNDS here (and every later operation with a Python number or another float32 value happens in float32):
        only values that are exact in float32 -- masks -- pass, as a _NarrowSym that can do nothing but meet a double column."""
        if not getattr(self.g, "f64", False):
            return self._plain()
        if self.isbool or self.g.nodes[self.i][0] in ("gt", "ge"):
            return _NarrowSym(self.g, self.i)
        raise TraceUnsupported(f"{what} of a traced value in the fp64 build (the reference rounds to float32 there)")

    def view(self, *shape):
        return self._reshape(shape)

    def reshape(self, *shape):
        return self._reshape(shape)

    def _reshape(self, shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        shape = tuple(shape)
        if _per_point_size(shape) or (len(shape) == 2 and not _has_batch_dim(shape) and shape == (-1, 1)):
            return Sym(self.g, self.i)               # (a view: the same values, another tensor)
        raise TraceUnsupported(f"reshape to {shape} inside the fused path")

    def clone(self, *a, **k):
        return Sym(self.g, self.i)

    def detach(self):
        """x.detach(): the value, with no gradient flowing through it (stop-gradient weights inside an equation or a loss):
        a node of its own whose derivative and adjoint are zero."""
        return self._un("detach")

    def __getitem__(self, idx):
        # the whole (N, 1) column under another spelling: u[:, 0:1], u[:, [0]], u[:, :], u[...]
        # (u[:, 0] would be an (N,) vector: the column semantics of the trace do not carry that shape)
        full = slice(None)
        if _has_batch_dim(idx) or (isinstance(idx, slice) and _has_batch_dim((idx.start, idx.stop, idx.step))):
            raise TraceUnsupported("indexing a traced column with the batch size")
        if idx is Ellipsis or idx == full:
            return Sym(self.g, self.i)
        if isinstance(idx, tuple) and len(idx) == 2 and (idx[0] == full or idx[0] is Ellipsis):
            j = idx[1]
            if j == full or j == slice(0, 1) or j == slice(0, None) or j == slice(None, 1) or j == [0] or j is Ellipsis:
                return Sym(self.g, self.i)
        raise TraceUnsupported("indexing a traced column")

    @property
    def dtype(self):
        # the precision the kernels of this trace compute in (engine.trace_system): constants made "like" a traced column
        # must not be rounded to another one on the way (x.new_tensor(0.3) under the fp64 build)
        f64 = getattr(self.g, "f64", None)
        return torch.get_default_dtype() if f64 is None else (torch.float64 if f64 else torch.float32)

    @property
    def device(self):
        return torch.device("cpu")           # (constants built "on u.device" are folded into the trace as numbers)

    def new_tensor(self, data, **k): return torch.as_tensor(data, dtype=torch.float64)     # (numbers of the trace are doubles)
    def new_full(self, size, fill_value, **k): return torch.full(_const_size(size), float(fill_value), dtype=torch.float64)
    def new_ones(self, *size, **k): return torch.ones(_const_size(size), dtype=torch.float64)
    def new_zeros(self, *size, **k): return torch.zeros(_const_size(size), dtype=torch.float64)
    def expand_as(self, other): return Sym(self.g, self.i)
    def contiguous(self, *a, **k): return self

    def expand(self, *size):
        return self._reshape(size)

    def __bool__(self):
        raise TraceUnsupported("data-dependent control flow on a traced value")

    def __len__(self):
        # len() must return a plain int -- the batch size, which a trace does not have (_BatchDim)
        _batch_dim(self.g)._refuse()

    __hash__ = object.__hash__

    # ---- arithmetic
    def _bin(self, other, op, rev=False):
        g = self.g
        if isinstance(other, SymMat):
            return other._bin(self, op, rev=not rev)
        if isinstance(other, _NarrowSym):
            return other._bin(self, op, rev=not rev)
        if op == "sub" and (self.isbool or (isinstance(other, Sym) and other.isbool)):
            # `u - mask`, `1 - mask`, `mask - mask`: torch raises for ANY subtraction with a bool tensor (use ~mask, mask.float())
            raise TraceUnsupported("subtraction with a bool tensor (torch raises; use ~, ^, logical_xor or mask.float())")
        if self.isbool and _is_boolish(other):
            # torch.bool (op) torch.bool: + is OR, * is AND (both stay bool); / is the float quotient
            if op in ("add", "mul"):
                o = other.i if isinstance(other, Sym) else g.const(1.0 if bool(other) else 0.0)
                r = g.mul(self.i, o) if op == "mul" else g.sub(g.add(self.i, o), g.mul(self.i, o))
                return Sym(g, r)._as_bool()
        row = _row_values(other)
        if row is not None:          # column (N,1) with a constant row (k,) broadcasts to an (N,k) matrix
            return SymMat([self] * len(row))._bin(other, op, rev)
        try:
            o = _as_node(g, other)
        except TraceUnsupported:
            return NotImplemented
        a, b = (o, self.i) if rev else (self.i, o)
        return Sym(g, getattr(g, op)(a, b))

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "div")
    def __rtruediv__(self, o): return self._bin(o, "div", True)
    def __neg__(self):
        if self.isbool:
            raise TraceUnsupported("negation of a bool tensor (torch raises; use ~ or logical_not)")
        return Sym(self.g, self.g.unary("neg", self.i))
    def __pos__(self): return self
    def __abs__(self):
        if self.isbool:
            raise TraceUnsupported("abs of a bool tensor (torch has no such kernel)")
        return Sym(self.g, self.g.unary("abs", self.i))

    def __pow__(self, e):
        if isinstance(e, Sym):
            # a ** b = exp(b log a)
            g = self.g
            return Sym(g, g.unary("exp", g.mul(e.i, g.unary("log", self.i))))
        c = self.g.cval(_as_node(self.g, e, literal=True))
        return Sym(self.g, self.g.powc(self.i, c))

    def __rpow__(self, base):
        g = self.g
        c = g.cval(_as_node(g, base, literal=True))
        return Sym(g, g.unary("exp", g.mul(self.i, g.const(math.log(c)))))

    # methods mirroring torch.Tensor
    def _un(self, op):
        return Sym(self.g, self.g.unary(op, self.i))

    def sin(self): return self._un("sin")
    def cos(self): return self._un("cos")
    def tan(self): return self._un("tan")
    def exp(self): return self._un("exp")
    def log(self): return self._un("log")
    def tanh(self): return self._un("tanh")
    def sqrt(self): return self._un("sqrt")
    def abs(self): return self._un("abs")
    def sinh(self): return self._un("sinh")
    def cosh(self): return self._un("cosh")
    def sigmoid(self): return self._un("sigmoid")
    def reciprocal(self): return self._un("recip")
    def square(self): return self * self
    def pow(self, e): return self ** e
    def mean(self, dim=None, keepdim=False, **k): return _batch_mean(self, dim, keepdim)
    def sign(self): return self._un("sign")
    def sgn(self): return self._un("sign")
    def log1p(self): return self._un("log1p")
    def expm1(self): return self._un("expm1")
    def erf(self): return self._un("erf")
    def atan(self): return self._un("atan")
    def arctan(self): return self._un("atan")
    def atan2(self, other): return _tf_atan2(self, other)
    def arctan2(self, other): return _tf_atan2(self, other)

    # ---- comparisons -> masks (0.0 / 1.0 columns, zero derivative); piecewise functions in torch's own subgradients
    def _cmp(self, other, op, swap):
        g = self.g
        try:
            o = _as_node(g, other)
        except TraceUnsupported:
            return NotImplemented
        a, b = (o, self.i) if swap else (self.i, o)
        return Sym(g, getattr(g, op)(a, b))._as_bool()

    def _eq(self, o):
        ge, le = self._cmp(o, "ge", False), self._cmp(o, "ge", True)
        return NotImplemented if ge is NotImplemented else ge * le          # [a >= b] [b >= a]: 0 where either is nan (bool * bool: bool)

    def __eq__(self, o):                 # masks like the other comparisons (`t == 0`; a Python bool here would pick ONE
        return self._eq(o)               # branch of a torch.where for every point)

    def __ne__(self, o):
        m = self._eq(o)
        return NotImplemented if m is NotImplemented else (1.0 - m._plain())._as_bool()

    def eq(self, o): return self == o
    def ne(self, o): return self != o
    def not_equal(self, o): return self != o
    def __gt__(self, o): return self._cmp(o, "gt", False)
    def __lt__(self, o): return self._cmp(o, "gt", True)
    def __ge__(self, o): return self._cmp(o, "ge", False)
    def __le__(self, o): return self._cmp(o, "ge", True)
    def gt(self, o): return self > o
    def lt(self, o): return self < o
    def ge(self, o): return self >= o
    def le(self, o): return self <= o
    def greater(self, o): return self > o
    def less(self, o): return self < o
    # ~ & | ^ : torch implements them for bool (and integer) tensors only -- on float columns it raises
    def _bits(self, o, what):
        if not self.isbool or not _is_boolish(o):
            raise TraceUnsupported(f"{what} of traced values that are not bool masks (torch raises for floating-point tensors)")
        return o._plain() if isinstance(o, Sym) else (1.0 if bool(o) else 0.0)

    def __invert__(self):
        self._bits(True, "~")
        return (1.0 - self._plain())._as_bool()
    def __and__(self, o): return (self._plain() * self._bits(o, "&"))._as_bool()
    __rand__ = __and__
    def __or__(self, o):
        b = self._bits(o, "|")
        a = self._plain()
        return (a + b - a * b)._as_bool()
    __ror__ = __or__
    def __xor__(self, o):
        b = self._bits(o, "^")
        a = self._plain()
        return (a + b - 2.0 * (a * b))._as_bool()
    __rxor__ = __xor__
    # logical_*: any dtype, "non-zero is True"
    def logical_not(self): return _tf_logical("not", self)
    def logical_and(self, o): return _tf_logical("and", self, o)
    def logical_or(self, o): return _tf_logical("or", self, o)
    def logical_xor(self, o): return _tf_logical("xor", self, o)
    def float(self): return self._narrowed("Tensor.float()")
    def double(self): return self._plain()

    def bool(self):
        # of a mask: itself; of any other column: [x != 0] (arithmetic on the result must see 0 / 1, not the value)
        return Sym(self.g, self.i)._as_bool() if (self.isbool or self.g.nodes[self.i][0] in ("gt", "ge")) else (self != 0.0)

    def _cast(self, args, kwargs):
        """x.to(...) / x.type(...): float32 <-> float64 is the precision of the build either way (the contract is 1e-5 /
        1e-9 against fp64); a cast to a NARROWER or integer type rounds the values in the reference -- refused."""
        for v in list(args) + list(kwargs.values()):
            if isinstance(v, torch.dtype) and v not in (torch.float32, torch.float64):
                raise TraceUnsupported(f"cast of a traced column to {v}")
            if isinstance(v, str) and "Float" not in v and "Double" not in v and v not in ("cpu", "cuda") and not v.startswith("cuda:"):
                raise TraceUnsupported(f"cast of a traced column to {v!r}")
            if isinstance(v, type) and issubclass(v, torch.Tensor) and v not in (torch.FloatTensor, torch.DoubleTensor, torch.Tensor):
                raise TraceUnsupported(f"cast of a traced column to {v.__name__}")
        for v in list(args) + list(kwargs.values()):
            if v is torch.float32 or v is torch.FloatTensor or (isinstance(v, str) and "Float" in v) or \
                    (isinstance(v, torch.Tensor) and v.dtype == torch.float32) or isinstance(v, _NarrowSym):
                return self._narrowed("a cast to float32")
            if isinstance(v, Sym) and v.isbool:
                return self.bool()
        return self._plain()

    def to(self, *a, **k): return self._cast(a, k)
    def type(self, *a, **k): return self._cast(a, k)
    def type_as(self, other): return self._cast((other,), {})
    def where(self, condition, other): return _tf_where(condition, self, other)
    def masked_fill(self, mask, value): return _tf_where(mask, value, self)
    def __mod__(self, o): return _tf_remainder(self, o)          # torch's `%`: the sign of the divisor (floor)
    def __rmod__(self, o): return _tf_remainder(o, self)
    def __matmul__(self, m): return _tf_matmul(self, m)
    def relu(self): return _tf_where(self > 0.0, self, 0.0)
    def clamp(self, min=None, max=None): return _tf_clamp(self, min, max)
    def clip(self, min=None, max=None): return _tf_clamp(self, min, max)
    def clamp_min(self, min): return _tf_clamp(self, min, None)
    def clamp_max(self, max): return _tf_clamp(self, None, max)
    def maximum(self, other): return _tf_maximum(self, other)
    def minimum(self, other): return _tf_minimum(self, other)

    # ---- torch.* functions
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        h = _TORCH_FUNCS.get(name)
        if h is None:
            raise TraceUnsupported(f"torch.{name} is not supported inside the fused path")
        return h(*args, **kwargs)

    def __repr__(self):
        return f"Sym#{self.i}{self.g.nodes[self.i]}"

    __getattr__ = _no_such_method("column")


def _is_boolish(v):
    return (isinstance(v, Sym) and v.isbool) or isinstance(v, bool) or (isinstance(v, torch.Tensor) and v.dtype == torch.bool and v.numel() == 1)


class _NarrowSym(Sym):
    """A float32-typed mask in the fp64 build (``(x > 0).float()``, ``torch.ones_like(u, dtype=torch.float32)``): values that
    are exact in float32.  torch's type promotion makes everything it meets FIRST decide what happens next: with a double column
    the result is an ordinary double column (exact: that is all this class allows, besides sums / products / comparisons among
    its own kind); with a Python number, a 0-dim tensor or a function the reference computes in float32 -- ``m.float() * 0.1``
    is 0.1f -- which the fp64 kernels do not reproduce: TraceUnsupported -> the loud composite path.  Reading ``.i`` raises, so no
    consumer of the trace can take the value for a double by accident."""

    def __init__(self, g, i):
        self.g, self._node, self.leaf = g, i, False

    @property
    def i(self):
        raise TraceUnsupported("a float32 value inside the fp64 build meets something other than a double column (the reference "
                               "computes that in float32)")

    def _plain(self):
        return Sym(self.g, self._node)

    def _narrowed(self, what):
        return self

    def double(self):
        return self._plain()

    def _cast(self, args, kwargs):
        for v in list(args) + list(kwargs.values()):
            if v is torch.float64 or v is torch.DoubleTensor or (isinstance(v, Sym) and not isinstance(v, _NarrowSym) and not v.isbool):
                return self._plain()
            if v is torch.float32 or v is torch.FloatTensor or isinstance(v, _NarrowSym):
                return self
        raise TraceUnsupported("cast of a float32 value inside the fp64 build")

    def bool(self):
        return (self._plain() != 0.0)

    def _bin(self, other, op, rev=False):
        g = self.g
        if isinstance(other, _NarrowSym):
            if op in ("add", "sub", "mul"):              # small integers: exact in float32, and float32 again
                a, b = (other._node, self._node) if rev else (self._node, other._node)
                return _NarrowSym(g, getattr(g, op)(a, b))
            raise TraceUnsupported("a quotient of float32 values inside the fp64 build")
        if isinstance(other, Sym) and not other.isbool:
            a, b = (other.i, self._node) if rev else (self._node, other.i)
            return Sym(g, getattr(g, op)(a, b))
        if isinstance(other, SymMat):
            return other._bin(self, op, rev=not rev)
        if _is_boolish(other) and op in ("mul",):
            o = other.i if isinstance(other, Sym) else g.const(1.0 if bool(other) else 0.0)
            return _NarrowSym(g, g.mul(self._node, o))
        raise TraceUnsupported("a float32 value inside the fp64 build combined with a number / tensor that is not a double column "
                               "(float32 arithmetic in the reference)")

    def _cmp(self, other, op, swap):
        if isinstance(other, _NarrowSym):
            other = other._plain()
        return self._plain()._cmp(other, op, swap)

    def __neg__(self): return _NarrowSym(self.g, self.g.unary("neg", self._node))
    def __pow__(self, e): self.i
    def __rpow__(self, e): self.i
    def __abs__(self): return _NarrowSym(self.g, self.g.unary("abs", self._node))
    def _un(self, op): self.i
    def detach(self): return self
    def clone(self, *a, **k): return _NarrowSym(self.g, self._node)
    def _reshape(self, shape):
        Sym._reshape(self._plain(), shape)
        return _NarrowSym(self.g, self._node)
    def __getitem__(self, idx):
        Sym.__getitem__(self._plain(), idx)
        return _NarrowSym(self.g, self._node)
    def _bits(self, o, what):
        raise TraceUnsupported(f"{what} of a float32 value (torch raises for floating-point tensors)")


def _truth(g, v):
    """``v`` as a bool column in torch's sense (non-zero is True): a traced column, a number, a one-element tensor."""
    if isinstance(v, _NarrowSym):
        v = v._plain()
    if isinstance(v, Sym):
        return v.bool()
    if isinstance(v, SymMat):
        raise TraceUnsupported("logical_* of a traced matrix")
    if isinstance(v, (bool, numbers.Number)):
        return Sym(g, g.const(1.0 if v else 0.0))._as_bool()
    if isinstance(v, torch.Tensor) and v.numel() == 1 and not v.requires_grad:
        return Sym(g, g.const(1.0 if bool(v) else 0.0))._as_bool()
    raise TraceUnsupported(f"logical_* with {type(v).__name__}")


def _tf_logical(kind, a, b=None, **k):
    if k.get("out") is not None:
        raise TraceUnsupported("out= inside the traced region")
    if isinstance(a, SymMat) or isinstance(b, SymMat):
        return _elementwise((lambda x: _tf_logical(kind, x)) if b is None else (lambda x, y: _tf_logical(kind, x, y)),
                            *([a] if b is None else [a, b]))
    g = _first_sym(a, b).g
    x = _truth(g, a)
    if kind == "not":
        return ~x
    y = _truth(g, b)
    return x & y if kind == "and" else (x | y if kind == "or" else x ^ y)


class SymMat:
    """Proxy for an ``(N, k)`` matrix inside a trace: k traced columns (multi-output networks, function bases)."""
    __array_priority__ = 10000
    __array_ufunc__ = None

    def __init__(self, cols):
        self.cols = list(cols)
        self.g = self.cols[0].g

    @property
    def shape(self):
        return _Shape((_batch_dim(self.g), len(self.cols)))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def __len__(self):
        _batch_dim(self.g)._refuse()

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None):
            j = idx[1]
            if isinstance(j, int):
                return self.cols[j]                     # caller reshapes with .view(-1, 1), a no-op here
            if isinstance(j, slice):
                sub = self.cols[j]
                return sub[0] if len(sub) == 1 else SymMat(sub)
        raise TraceUnsupported("only column selection is supported on a traced matrix")

    def _operand_cols(self, other):
        k = len(self.cols)
        if isinstance(other, SymMat):
            if len(other.cols) != k:
                raise TraceUnsupported("traced matrices of different widths")
            return [c.i for c in other.cols]
        row = _row_values(other)
        if row is not None:
            if len(row) != k:
                raise TraceUnsupported("constant row of the wrong width")
            return [self.g.const(v) for v in row]
        n = _as_node(self.g, other)
        return [n] * k

    def _bin(self, other, op, rev=False):
        k = len(self.cols)
        if isinstance(other, SymMat):
            if len(other.cols) != k:
                return NotImplemented                    # (traced matrices of different widths)
            others = other.cols
        else:
            row = _row_values(other)
            if row is not None and len(row) != k:
                return NotImplemented                    # (constant row of the wrong width)
            others = row if row is not None else [other] * k
        out = []
        for c, o in zip(self.cols, others):              # column by column: the dtype rules of Sym._bin hold per column
            r = c._bin(o, op, rev)
            if r is NotImplemented:
                return NotImplemented
            out.append(r)
        return SymMat(out)

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "div")
    def __rtruediv__(self, o): return self._bin(o, "div", True)
    def __neg__(self): return SymMat([-c for c in self.cols])
    def __pow__(self, e): return SymMat([c ** e for c in self.cols])

    def _un(self, op):
        return SymMat([c._un(op) for c in self.cols])

    def abs(self): return self._un("abs")
    def __abs__(self): return self._un("abs")
    def __matmul__(self, m): return _tf_matmul(self, m)
    def __mod__(self, o): return _tf_remainder(self, o)
    def where(self, condition, other): return _tf_where(condition, self, other)
    def masked_fill(self, mask, value): return _tf_where(mask, value, self)
    def square(self): return self * self
    def pow(self, e): return self ** e
    def mean(self, dim=None, keepdim=False, **k): return _batch_mean(self, dim, keepdim)

    def sum(self, dim=None, keepdim=False, keepdims=None, **kw):
        if keepdims is not None:
            keepdim = keepdims
        if dim not in (1, -1):
            raise TraceUnsupported("only row sums (dim=1) of a traced matrix are supported")
        acc = self.cols[0]._plain()                     # (of bool columns: the COUNT, an integer -- not the OR that `+` is)
        for c in self.cols[1:]:
            acc = acc + c
        return acc                                      # (N,) and (N,1) are the same traced column

    def view(self, *shape):
        return self.reshape(*shape)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        shape = tuple(shape)
        if len(shape) == 2 and (isinstance(shape[0], _BatchDim) or shape[0] == -1) and not isinstance(shape[1], _BatchDim) \
                and shape[1] in (-1, len(self.cols)) and shape != (-1, -1):
            return self
        raise TraceUnsupported(f"reshape of a traced matrix to {shape}")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return Sym.__torch_function__(func, types, args, kwargs)

    def __repr__(self):
        return f"SymMat[{len(self.cols)} cols]"

    __getattr__ = _no_such_method("matrix")


class SymScalar:
    """A traced scalar that is a MEAN over the batch of a per-point expression (plus what linear arithmetic makes of
    such means): value = (1 / N) * sum_p term(p).  What custom ``loss_fn`` callables, ``additional_loss`` overrides and
    ``metrics`` (solvers.py:216-226, 587-604, 377-379) return when they are traced; products of two means, batch sums
    (they need N) and anything else that is not linear in means raise TraceUnsupported -> composite path."""
    __array_priority__ = 10000
    __array_ufunc__ = None

    def __init__(self, term):
        self.term = term                  # Sym: the per-point term
        self.g = term.g

    def _lin(self, other, op, rev=False):
        if isinstance(other, SymScalar):
            if op == "add":
                return SymScalar(self.term + other.term)
            if op == "sub":
                return SymScalar(other.term - self.term if rev else self.term - other.term)
            raise TraceUnsupported("product / quotient of two batch means inside a traced loss")
        if isinstance(other, (Sym, SymMat)):
            raise TraceUnsupported("mixing a batch mean with per-point values inside a traced loss")
        c = Sym(self.g, _as_node(self.g, other))      # python number / scalar tensor (trainable ones are refused there)
        if op == "add":
            return SymScalar(self.term + c)
        if op == "sub":
            return SymScalar(c - self.term if rev else self.term - c)
        if op == "mul":
            return SymScalar(self.term * c)
        if op == "div":
            if rev:
                raise TraceUnsupported("division by a batch mean inside a traced loss")
            return SymScalar(self.term / c)
        raise TraceUnsupported(op)

    def __add__(self, o): return self._lin(o, "add")
    def __radd__(self, o): return self._lin(o, "add", True)
    def __sub__(self, o): return self._lin(o, "sub")
    def __rsub__(self, o): return self._lin(o, "sub", True)
    def __mul__(self, o): return self._lin(o, "mul")
    def __rmul__(self, o): return self._lin(o, "mul", True)
    def __truediv__(self, o): return self._lin(o, "div")
    def __rtruediv__(self, o): return self._lin(o, "div", True)
    def __neg__(self): return SymScalar(-self.term)
    def __pos__(self): return self

    def mean(self, *a, **k): return self
    def sum(self, *a, **k): return self
    def squeeze(self, *a, **k): return self
    def reshape(self, *a, **k): return self
    def view(self, *a, **k): return self

    @property
    def shape(self):
        return torch.Size([])

    def dim(self):
        return 0

    def item(self):
        raise TraceUnsupported(".item() on a traced scalar")

    def __float__(self):
        raise TraceUnsupported("float() of a traced scalar")

    def __bool__(self):
        raise TraceUnsupported("data-dependent control flow on a traced scalar")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return Sym.__torch_function__(func, types, args, kwargs)

    def __repr__(self):
        return f"SymScalar(mean of {self.term!r})"

    __getattr__ = _no_such_method("scalar")


class SymScalarVec:
    """Column-wise batch means of a traced matrix (``x.mean(dim=0)``): one :class:`SymScalar` per column."""

    def __init__(self, items):
        self.items = list(items)
        self.g = self.items[0].g

    def _weights(self, other):
        row = _row_values(other)
        if row is not None:
            if len(row) != len(self.items):
                raise TraceUnsupported("constant row of the wrong width")
            return row
        return None

    def __mul__(self, o):
        w = self._weights(o)
        return SymScalarVec([it * (w[j] if w is not None else o) for j, it in enumerate(self.items)])
    __rmul__ = __mul__

    def __truediv__(self, o):
        w = self._weights(o)
        return SymScalarVec([it / (w[j] if w is not None else o) for j, it in enumerate(self.items)])

    def __add__(self, o):
        if isinstance(o, SymScalarVec):
            return SymScalarVec([a + b for a, b in zip(self.items, o.items)])
        w = self._weights(o)
        return SymScalarVec([it + (w[j] if w is not None else o) for j, it in enumerate(self.items)])
    __radd__ = __add__

    def __getitem__(self, j):
        return self.items[j]

    def __len__(self):
        return len(self.items)

    def sum(self, *a, **k):
        acc = self.items[0]
        for it in self.items[1:]:
            acc = acc + it
        return acc

    def mean(self, *a, **k):
        return self.sum() / float(len(self.items))

    @property
    def shape(self):
        return torch.Size([len(self.items)])


def _batch_mean(x, dim=None, keepdim=False, **k):
    """``x.mean()`` / ``x.mean(dim=0)`` of a traced column or matrix."""
    if isinstance(x, (SymScalar, SymScalarVec)):
        return x.mean()
    if isinstance(x, SymMat):
        if dim in (1, -1):                                   # row means: still per-point
            return x.sum(dim=1) / float(len(x.cols))
        if dim in (0, -2):
            return SymScalarVec([SymScalar(c) for c in x.cols])
        if dim is None or tuple(dim) in ((0, 1), (1, 0)):
            return SymScalar(x.sum(dim=1) / float(len(x.cols)))
        raise TraceUnsupported(f"mean over dim={dim} of a traced matrix")
    if dim in (1, -1):
        return x
    if dim is None or dim in (0, -2) or tuple(dim) in ((0, 1), (1, 0)):
        return SymScalar(x)
    raise TraceUnsupported(f"mean over dim={dim} of a traced column")


def _tf_mse_loss(input, target, size_average=None, reduce=None, reduction="mean", **k):
    if reduction != "mean" or size_average is not None or reduce is not None:
        raise TraceUnsupported("only reduction='mean' losses are traced")
    d = input - target
    return _batch_mean(d * d)


def _tf_l1_loss(input, target, size_average=None, reduce=None, reduction="mean", **k):
    if reduction != "mean" or size_average is not None or reduce is not None:
        raise TraceUnsupported("only reduction='mean' losses are traced")
    d = input - target
    return _batch_mean(d._un("abs"))


def _first_sym(*args):
    for a in args:
        if isinstance(a, (Sym, SymMat)):
            return a
    raise TraceUnsupported("no traced operand")


def _no_options(a, k):
    """Options a handler does not know (out=, eps=, a rounding mode ...) are never dropped silently."""
    if a or any(v is not None and v is not False for v in k.values()):
        raise TraceUnsupported(f"unsupported arguments {tuple(a) + tuple(sorted(k))} of an elementwise function on a traced column")


def _floaty(x):
    """A bool mask handed to a MATH function (torch.sin(mask), torch.exp(mask)): torch promotes it to the default float dtype first."""
    if isinstance(x, Sym) and x.isbool:
        return x._plain()
    if isinstance(x, SymMat) and any(c.isbool for c in x.cols):
        return SymMat([c._plain() if c.isbool else c for c in x.cols])
    return x


def _tf_unary(op):
    def f(x, *a, **k):
        _no_options(a, k)
        return _floaty(x)._un(op)
    return f


_PY_OPS = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b}


def _tf_bin(op):
    def f(a, b, *rest, alpha=None, **k):
        s = _first_sym(a, b)
        if isinstance(a, SymMat) or isinstance(b, SymMat) or _row_values(a) is not None or _row_values(b) is not None:
            if alpha is not None:
                b = b * alpha
            return s._bin(b, op) if s is a else s._bin(a, op, True)
        if alpha is not None:
            b = b * alpha
        if isinstance(a, Sym):
            r = a._bin(b, op)
        else:
            r = b._bin(a, op, True)
        if r is NotImplemented:
            raise TraceUnsupported(f"torch.{op} of a traced column and {type(b if isinstance(a, Sym) else a).__name__}")
        return r
    return f


def _const_like(x, value, k):
    """A per-point constant made "like" the traced column x: its dtype is x's unless dtype= says otherwise."""
    g = x.g
    dt = k.get("dtype")
    if dt is torch.bool or (dt is None and isinstance(x, Sym) and x.isbool):
        return Sym(g, g.const(1.0 if value else 0.0))._as_bool()
    if dt in (torch.float16, torch.bfloat16) or (dt is torch.float32 and getattr(g, "f64", False)) or \
            (dt is None and isinstance(x, _NarrowSym)):
        v32 = float(torch.tensor(float(value), dtype=torch.float32 if dt is None else dt))
        if not getattr(g, "f64", False):
            if dt in (torch.float16, torch.bfloat16):
                raise TraceUnsupported(f"a constant of dtype {dt} inside the traced region")
            return Sym(g, g.const(v32))
        if dt in (torch.float16, torch.bfloat16):
            raise TraceUnsupported(f"a constant of dtype {dt} inside the traced region")
        return _NarrowSym(g, g.const(v32))               # (the float32 value, and float32 arithmetic with whatever it meets)
    if dt is not None and dt not in (torch.float32, torch.float64) and float(value) != int(value):
        raise TraceUnsupported(f"a constant of dtype {dt} inside the traced region")
    return Sym(g, g.const(value) if isinstance(value, numbers.Number) else _as_node(g, value))


def _tf_like(value):
    def f(x, *a, **k):
        if isinstance(x, SymMat):
            return SymMat([_const_like(c, value, k) for c in x.cols])
        return _const_like(x, value, k)
    return f


def _tf_cat(tensors, dim=0, **k):
    if dim not in (1, -1):
        raise TraceUnsupported("torch.cat of traced columns along dim 0")
    cols = []
    for t in tensors:
        if isinstance(t, SymMat):
            cols += t.cols
        elif isinstance(t, Sym):
            cols.append(t)
        else:
            raise TraceUnsupported("torch.cat mixing traced and concrete tensors")
    return SymMat(cols)


def _tf_sum(x, dim=None, keepdim=False, **k):
    if isinstance(x, (SymScalar, SymScalarVec)):
        return x.sum()
    if isinstance(x, SymMat):
        return x.sum(dim=dim, keepdim=keepdim)
    if dim in (1, -1):
        return x
    raise TraceUnsupported("reductions over the batch inside the traced region")


def _tf_full_like(x, fill_value, **k):
    if isinstance(x, SymMat):
        return SymMat([_const_like(c, fill_value, k) for c in x.cols])
    return _const_like(x, fill_value, k)


def _tf_pow(a, b):
    if isinstance(a, Sym):
        return a ** b
    return b.__rpow__(a)


def _columns(x, k, g):
    """``x`` (traced column / matrix, number, 1-element tensor) as k traced columns."""
    if isinstance(x, SymMat):
        if len(x.cols) != k:
            raise TraceUnsupported("traced matrices of different widths")
        return list(x.cols)
    if isinstance(x, Sym):
        return [x] * k
    return [Sym(g, _as_node(g, x))] * k


def _elementwise(fn, *operands):
    """Apply ``fn(*columns)`` column by column over traced columns / matrices / scalars (broadcast to the widest)."""
    s = _first_sym(*operands)
    k = max((len(x.cols) for x in operands if isinstance(x, SymMat)), default=0)
    if k == 0:
        return fn(*[x if isinstance(x, Sym) else Sym(s.g, _as_node(s.g, x)) for x in operands])
    cols = [_columns(x, k, s.g) for x in operands]
    return SymMat([fn(*[c[j] for c in cols]) for j in range(k)])


def _where1(m, a, b):
    if isinstance(m, _NarrowSym):
        raise TraceUnsupported("torch.where with a float32 condition (torch wants a bool mask)")
    r = Sym(m.g, m.g.where(m.i, a.i, b.i))
    return r._as_bool() if (a.isbool and b.isbool) else r


def _tf_where(condition, input=None, other=None, **k):
    if input is None or other is None:
        raise TraceUnsupported("torch.where(condition) without the two branches inside the traced region")
    if isinstance(condition, torch.Tensor):
        if condition.numel() != 1:
            raise TraceUnsupported("torch.where on a concrete mask of more than one element")
        return input if bool(condition) else other
    if not isinstance(condition, (Sym, SymMat)):
        if isinstance(condition, bool) and (isinstance(input, (Sym, SymMat)) or isinstance(other, (Sym, SymMat))):
            # a Python bool where torch has a per-point mask: some comparison of traced columns was decided by object
            # identity instead of point by point -- one branch for every point would be a silently different equation
            raise TraceUnsupported("torch.where with a Python bool as the condition of traced branches")
        return input if condition else other
    for c in ([condition] if isinstance(condition, Sym) else condition.cols):
        if not c.isbool and c.g.nodes[c._node if isinstance(c, _NarrowSym) else c.i][0] not in ("gt", "ge"):
            # torch: "where expected condition to be a boolean tensor" -- a float column (also 1 - mask, mask.float()) raises there
            raise TraceUnsupported("torch.where / masked_fill with a condition that is not a bool mask (torch raises)")
    return _elementwise(_where1, condition, input, other)


def _tf_clamp(x, min=None, max=None, **k):
    """torch.clamp: the value inside [min, max] (derivative 1, bounds included), the bound outside (derivative 0)."""
    def one(col, *bounds):
        bounds = list(bounds)
        lo = bounds.pop(0) if min is not None else None
        hi = bounds.pop(0) if max is not None else None
        out = col
        if hi is not None:
            out = _where1(col > hi, hi, out)
        if lo is not None:
            out = _where1(col < lo, lo, out)       # (min wins where min > max, as in torch)
        return out
    return _elementwise(one, x, *[b for b in (min, max) if b is not None])


def _tf_maximum(a, b, **k):
    """torch.maximum / torch.max(a, b): the gradient is split evenly where the operands tie."""
    def one(x, y):
        if x.isbool and y.isbool:
            return x | y                                 # (of two bool masks: the OR, a bool again)
        x, y = Sym(x.g, x.i), Sym(y.g, y.i)              # (the numbers, whatever torch dtype flags the operands carried)
        return _where1(x > y, x, _where1(y > x, y, 0.5 * (x + y)))
    return _elementwise(one, a, b)


def _tf_minimum(a, b, **k):
    def one(x, y):
        if x.isbool and y.isbool:
            return x & y
        x, y = Sym(x.g, x.i), Sym(y.g, y.i)
        return _where1(x < y, x, _where1(y < x, y, 0.5 * (x + y)))
    return _elementwise(one, a, b)


def _tf_max(a, b=None, *rest, **k):
    if b is None or rest or k or not isinstance(b, (Sym, SymMat, torch.Tensor, numbers.Number)) or isinstance(b, bool) or (isinstance(b, int) and not isinstance(a, numbers.Number)):
        raise TraceUnsupported("torch.max over a dimension of traced values")
    return _tf_maximum(a, b)


def _tf_min(a, b=None, *rest, **k):
    if b is None or rest or k or not isinstance(b, (Sym, SymMat, torch.Tensor, numbers.Number)) or isinstance(b, bool) or (isinstance(b, int) and not isinstance(a, numbers.Number)):
        raise TraceUnsupported("torch.min over a dimension of traced values")
    return _tf_minimum(a, b)


def _tf_atan2(a, b, **k):
    return _elementwise(lambda y, x: Sym(y.g, y.g.atan2(y.i, x.i)), a, b)


def _tf_relu(x, inplace=False, **k):
    return _elementwise(lambda c: _where1(c > 0.0, c, Sym(c.g, c.g.const(0.0))), x)


def _tf_leaky_relu(x, negative_slope=0.01, inplace=False, **k):
    return _elementwise(lambda c: _where1(c > 0.0, c, c * float(negative_slope)), x)


def _tf_heaviside(x, values, **k):
    return _elementwise(lambda c, v: _where1(c > 0.0, Sym(c.g, c.g.const(1.0)), _where1(c < 0.0, Sym(c.g, c.g.const(0.0)), v)), x, values)


def _tf_cmp(op, swap):
    def f(a, b, **k):
        return _elementwise(lambda x, y: x._cmp(y, op, swap), a, b)
    return f


def _tf_map(op):
    def f(x, *a, **k):
        _no_options(a, k)
        return _elementwise(lambda c: c._un(op), _floaty(x))
    return f


def _safe_den(x):
    """x where x != 0, else 1: the denominator of a quotient whose x = 0 case is selected away (no nan in value or gradient)."""
    return _where1(abs(x) > 0.0, x, Sym(x.g, x.g.const(1.0)))


def _tf_softplus(x, beta=1.0, threshold=20.0, **k):
    beta, threshold = float(beta), float(threshold)
    # torch: x where beta x > threshold, log1p(exp(beta x)) / beta elsewhere (the argument of exp is clipped so that the branch
    # not taken stays finite)
    return _elementwise(lambda c: _where1(c * beta > threshold, c, (_where1(c * beta > threshold, c * 0.0, c * beta).exp()).log1p() / beta), x)


def _tf_elu(x, alpha=1.0, inplace=False, **k):
    return _elementwise(lambda c: _where1(c > 0.0, c, _where1(c > 0.0, c * 0.0, c).expm1() * float(alpha)), x)


def _tf_selu(x, inplace=False, **k):
    return _tf_elu(x, 1.6732632423543772848170429916717) * 1.0507009873554804934193349852946


def _tf_celu(x, alpha=1.0, inplace=False, **k):
    a = float(alpha)
    # torch DIFFERENTIATES celu with the inverse of alpha rounded through float32 (derivatives.yaml: elu_backward(grad, alpha, 1,
    # 1.0 / alpha.toFloat(), ...)) while the value uses the double: in the fp64 build that is a 1e-8 difference in every gradient
    # through it.  Value from the exact branch, derivatives (all orders) from the one torch differentiates.
    k32 = 1.0 / float(torch.tensor(a, dtype=torch.float32))

    def one(c):
        neg = _where1(c > 0.0, c * 0.0, c)
        val = (neg / a).expm1() * a
        if k32 == 1.0 / a:
            return _where1(c > 0.0, c, val)
        dv = (neg * k32).expm1() * a
        return _where1(c > 0.0, c, dv + (val - dv).detach())
    return _elementwise(one, x)


def _hardsigmoid(c):
    # value clamp(c / 6 + 1 / 2, 0, 1); torch's backward multiplies by 1.0f / 6.0f inside (-3, 3) whatever the dtype
    # (hardsigmoid_backward: `one_sixth`), zero second derivative -- value from the exact formula, derivatives from that one
    val = _tf_clamp(c / 6.0 + 0.5, 0.0, 1.0)
    inside = (c > -3.0) & (c < 3.0)
    dv = _where1(inside, c * 0.1666666716337204, c * 0.0)
    return dv + (val - dv).detach()


def _tf_gelu(x, approximate="none", **k):
    if approximate == "tanh":
        return _elementwise(lambda c: 0.5 * c * (1.0 + (0.7978845608028654 * (c + 0.044715 * c * c * c)).tanh()), x)
    return _elementwise(lambda c: 0.5 * c * (1.0 + (c * 0.7071067811865476).erf()), x)


def _tf_hardtanh(x, min_val=-1.0, max_val=1.0, inplace=False, **k):
    return _tf_clamp(x, float(min_val), float(max_val))


def _tf_threshold(x, threshold, value, inplace=False, **k):
    return _elementwise(lambda c: _where1(c > float(threshold), c, Sym(c.g, c.g.const(float(value)))), x)


def _tf_logaddexp(a, b, **k):
    return _elementwise(lambda x, y: _tf_maximum(x, y) + (-(abs(x - y))).exp().log1p(), a, b)


def _tf_sinc(x, **k):
    def one(c):
        d = _safe_den(c) * math.pi
        return _where1(abs(c) > 0.0, d.sin() / d, Sym(c.g, c.g.const(1.0)))
    return _elementwise(one, x)


def _tf_fmod(a, b, **k):
    return _elementwise(lambda x, y: x - y * (x / y)._un("trunc"), a, b)


def _tf_remainder(a, b, **k):
    return _elementwise(lambda x, y: x - y * (x / y)._un("floor"), a, b)


def _tf_matmul(a, m, **k):
    """(N, k) traced matrix / column times a CONSTANT (k, m) matrix (a change of basis, a small mixing matrix): m linear
    combinations of the columns, point by point.  ``torch.einsum`` / a traced right operand stay outside the traced family."""
    if isinstance(m, (Sym, SymMat)) or not isinstance(a, (Sym, SymMat)):
        raise TraceUnsupported("matrix products with a traced RIGHT operand (only `columns @ constant matrix` is traced)")
    cols = a.cols if isinstance(a, SymMat) else [a]
    if isinstance(m, torch.Tensor):
        if m.requires_grad:
            raise TraceUnsupported("a trainable matrix inside the equations")
        m = m.detach().cpu().double().tolist()
    else:
        try:
            import numpy as np
            m = np.asarray(m, dtype=float).tolist()
        except Exception as e:      # noqa: BLE001
            raise TraceUnsupported(f"matrix product with {type(m).__name__}") from e
    if not isinstance(m, list) or not m or not all(isinstance(row, list) and len(row) == len(m[0]) for row in m) or len(m) != len(cols):
        raise TraceUnsupported("matrix product: the constant operand must be a (k, m) matrix matching the k traced columns")
    g = cols[0].g
    out = []
    for j in range(len(m[0])):
        acc = None
        for c, row in zip(cols, m):
            term = Sym(g, g.mul(c.i, g.const(float(row[j]))))         # (entries of a matrix are literals of the kernel)
            acc = term if acc is None else acc + term
        out.append(acc)
    return out[0] if len(out) == 1 else SymMat(out)


def _tf_norm(x, p=None, dim=None, keepdim=False, ord=None, **k):
    """Row norms of a traced matrix (``torch.norm(torch.cat([ux, uy], 1), dim=1, keepdim=True)``: |grad u| point by point);
    p / ord in {2 (default), 1, inf}.  A norm over the batch (dim=0 / no dim) is an operation ACROSS points: refused."""
    ord_ = ord if ord is not None else (2 if p is None else p)
    if isinstance(dim, (list, tuple)) and len(dim) == 1:
        dim = dim[0]
    if dim not in (1, -1):
        raise TraceUnsupported("a norm over the batch inside the traced region (only row norms, dim=1, are per-point)")
    cols = x.cols if isinstance(x, SymMat) else [x]
    if ord_ in (2, 2.0, "fro"):
        acc = None
        for c in cols:
            acc = c * c if acc is None else acc + c * c
        return acc.sqrt()
    if ord_ in (1, 1.0):
        acc = None
        for c in cols:
            acc = abs(c) if acc is None else acc + abs(c)
        return acc
    if ord_ in (float("inf"), math.inf):
        acc = None
        for c in cols:
            acc = abs(c) if acc is None else _tf_maximum(acc, abs(c))
        return acc
    raise TraceUnsupported(f"norm of order {ord_!r} of a traced matrix")


def _tf_cross(a, b, dim=None, **k):
    """torch.cross / torch.linalg.cross of two traced (N, 3) matrices along dim 1."""
    if not (isinstance(a, SymMat) and isinstance(b, SymMat) and len(a.cols) == 3 and len(b.cols) == 3) or dim not in (None, 1, -1):
        raise TraceUnsupported("torch.cross: two traced (N, 3) matrices, dim=1")
    (a0, a1, a2), (b0, b1, b2) = a.cols, b.cols
    return SymMat([a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0])


def _tf_nan_to_num(x, nan=0.0, posinf=None, neginf=None, **k):
    def one(c):
        big = 1.7976931348623157e308 if getattr(c.g, "f64", False) else 3.4028234663852886e38
        hi = big if posinf is None else float(posinf)
        lo = -big if neginf is None else float(neginf)
        v = _where1(c > big, Sym(c.g, c.g.const(hi)), _where1(c < -big, Sym(c.g, c.g.const(lo)), c))
        return _where1(c == c, v, Sym(c.g, c.g.const(float(nan))))    # (x == x is false exactly at nan: Graph.ge does not fold)
    return _elementwise(one, x)


def _tf_round(x, decimals=0, **k):
    if decimals:
        s = 10.0 ** int(decimals)
        return _elementwise(lambda c: (c * s)._un("round") / s, x)
    return _elementwise(lambda c: c._un("round"), x)


def _tf_lerp(a, b, weight, **k):
    return _elementwise(lambda x, y, w: x + w * (y - x), a, b, weight)


def _tf_addcmul(x, t1, t2, value=1.0, **k):
    return _elementwise(lambda c, p, q: c + float(value) * p * q, x, t1, t2)


def _tf_addcdiv(x, t1, t2, value=1.0, **k):
    return _elementwise(lambda c, p, q: c + float(value) * p / q, x, t1, t2)


def _tf_elem(fn):
    def f(x, *a, **k):
        _no_options(a, k)            # (torch.logit(x, eps=...) has a handler of its own)
        return _elementwise(fn, _floaty(x))
    return f


def _tf_logit(x, eps=None, **k):
    def one(c):
        if eps is not None:
            c = _tf_clamp(c, float(eps), 1.0 - float(eps))
        return (c / (1.0 - c)).log()
    return _elementwise(one, x)


def _tf_xlogy(a, b, **k):
    # x log y with 0 where x == 0 (whatever y is)
    return _elementwise(lambda x, y: _where1(abs(x) > 0.0, x * _where1(abs(x) > 0.0, y, y * 0.0 + 1.0).log(), x * 0.0), a, b)


def _tf_xlog1py(a, b, **k):
    return _elementwise(lambda x, y: _where1(abs(x) > 0.0, x * _where1(abs(x) > 0.0, y, y * 0.0).log1p(), x * 0.0), a, b)


def _tf_autograd_grad(outputs, inputs, grad_outputs=None, retain_graph=None, create_graph=False, only_inputs=True,
                      allow_unused=None, is_grads_batched=False, materialize_grads=False):
    """torch.autograd.grad on traced columns -- what `diff` wraps (neurodiffeq.py:21-34), written out by hand in user code.  Every
    point's value depends on that point's coordinates only, so the vector-Jacobian product with ``grad_outputs`` is
    grad_outputs * d output / d input, point by point; several outputs add up."""
    if is_grads_batched:
        raise TraceUnsupported("torch.autograd.grad(is_grads_batched=True) inside the traced region")
    outs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
    ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
    if grad_outputs is None:
        raise TraceUnsupported("torch.autograd.grad of a per-point column without grad_outputs")
    gos = list(grad_outputs) if isinstance(grad_outputs, (list, tuple)) else [grad_outputs]
    if len(gos) != len(outs) or not all(isinstance(o, Sym) for o in outs) or not all(isinstance(i, Sym) for i in ins):
        raise TraceUnsupported("torch.autograd.grad: traced (N, 1) columns with one grad_outputs entry per output")
    res = []
    for x in ins:
        total = None
        for o, go in zip(outs, gos):
            term = sym_diff(o, x) * go
            total = term if total is None else total + term
        # create_graph=False (torch's default): the result carries no graph -- nothing differentiates through it, neither a
        # further diff() nor the loss; `diff` itself always asks for create_graph=True (neurodiffeq.py:22,29)
        res.append(total if create_graph else total.detach())
    return tuple(res)


_TORCH_FUNCS = {
    "grad": _tf_autograd_grad,
    "detach": _tf_map("detach"),
    "xlogy": _tf_xlogy, "xlog1py": _tf_xlog1py, "logit": _tf_logit, "expit": _tf_unary("sigmoid"),
    "floor": _tf_map("floor"), "ceil": _tf_map("ceil"), "trunc": _tf_map("trunc"), "fix": _tf_map("trunc"), "round": _tf_round,
    "frac": _tf_elem(lambda c: c - c._un("trunc")), "fmod": _tf_fmod, "remainder": _tf_remainder,
    "asin": _tf_elem(lambda c: _tf_atan2(c, (1.0 - c * c).sqrt())), "arcsin": _tf_elem(lambda c: _tf_atan2(c, (1.0 - c * c).sqrt())),
    "acos": _tf_elem(lambda c: _tf_atan2((1.0 - c * c).sqrt(), c)), "arccos": _tf_elem(lambda c: _tf_atan2((1.0 - c * c).sqrt(), c)),
    "asinh": _tf_elem(lambda c: (c + (c * c + 1.0).sqrt()).log()), "arcsinh": _tf_elem(lambda c: (c + (c * c + 1.0).sqrt()).log()),
    "acosh": _tf_elem(lambda c: (c + (c * c - 1.0).sqrt()).log()), "arccosh": _tf_elem(lambda c: (c + (c * c - 1.0).sqrt()).log()),
    "atanh": _tf_elem(lambda c: 0.5 * ((1.0 + c) / (1.0 - c)).log()), "arctanh": _tf_elem(lambda c: 0.5 * ((1.0 + c) / (1.0 - c)).log()),
    "rsqrt": _tf_elem(lambda c: c ** -0.5), "log10": _tf_elem(lambda c: c.log() * (1.0 / math.log(10.0))),
    "log2": _tf_elem(lambda c: c.log() * (1.0 / math.log(2.0))), "exp2": _tf_elem(lambda c: (c * math.log(2.0)).exp()),
    "erfc": _tf_elem(lambda c: 1.0 - c.erf()), "sinc": _tf_sinc, "hypot": lambda a, b, **k: _elementwise(lambda x, y: (x * x + y * y).sqrt(), a, b),
    "logaddexp": _tf_logaddexp, "lerp": _tf_lerp, "addcmul": _tf_addcmul, "addcdiv": _tf_addcdiv,
    "softplus": _tf_softplus, "elu": _tf_elu, "selu": _tf_selu, "celu": _tf_celu, "gelu": _tf_gelu,
    "silu": _tf_elem(lambda c: c * c.sigmoid()), "mish": lambda x, **k: _elementwise(lambda c: c * _tf_softplus(c).tanh(), x),
    "softsign": _tf_elem(lambda c: c / (1.0 + abs(c))), "hardtanh": _tf_hardtanh, "relu6": _tf_elem(lambda c: _tf_clamp(c, 0.0, 6.0)),
    "hardsigmoid": _tf_elem(_hardsigmoid), "hardswish": _tf_elem(lambda c: c * _tf_clamp(c / 6.0 + 0.5, 0.0, 1.0)),
    "log_sigmoid": _tf_elem(lambda c: -_tf_softplus(-c)), "logsigmoid": _tf_elem(lambda c: -_tf_softplus(-c)),
    "threshold": _tf_threshold, "_threshold": _tf_threshold, "tanhshrink": _tf_elem(lambda c: c - c.tanh()),
    "where": _tf_where, "clamp": _tf_clamp, "clip": _tf_clamp, "masked_fill": lambda x, mask, value, **k: _tf_where(mask, value, x),
    "matmul": _tf_matmul, "mm": _tf_matmul, "nan_to_num": _tf_nan_to_num,
    "norm": _tf_norm, "linalg_norm": _tf_norm, "linalg_vector_norm": _tf_norm, "cross": _tf_cross, "linalg_cross": _tf_cross,
    "clamp_min": lambda x, min, **k: _tf_clamp(x, min, None), "clamp_max": lambda x, max, **k: _tf_clamp(x, None, max),
    "relu": _tf_relu, "leaky_relu": _tf_leaky_relu, "heaviside": _tf_heaviside,
    "maximum": _tf_maximum, "minimum": _tf_minimum, "max": _tf_max, "min": _tf_min, "fmax": _tf_maximum, "fmin": _tf_minimum,
    "sign": _tf_map("sign"), "sgn": _tf_map("sign"), "log1p": _tf_map("log1p"), "expm1": _tf_map("expm1"), "erf": _tf_map("erf"),
    "atan": _tf_map("atan"), "arctan": _tf_map("atan"), "atan2": _tf_atan2, "arctan2": _tf_atan2,
    "eq": lambda a, b, **k: _elementwise(lambda x, y: x == y, a, b), "ne": lambda a, b, **k: _elementwise(lambda x, y: x != y, a, b),
    "not_equal": lambda a, b, **k: _elementwise(lambda x, y: x != y, a, b),
    "gt": _tf_cmp("gt", False), "greater": _tf_cmp("gt", False), "lt": _tf_cmp("gt", True), "less": _tf_cmp("gt", True),
    "ge": _tf_cmp("ge", False), "greater_equal": _tf_cmp("ge", False), "le": _tf_cmp("ge", True), "less_equal": _tf_cmp("ge", True),
    "logical_not": functools.partial(_tf_logical, "not"), "logical_and": functools.partial(_tf_logical, "and"),
    "logical_or": functools.partial(_tf_logical, "or"), "logical_xor": functools.partial(_tf_logical, "xor"),
    "sin": _tf_unary("sin"), "cos": _tf_unary("cos"), "tan": _tf_unary("tan"), "exp": _tf_unary("exp"),
    "log": _tf_unary("log"), "tanh": _tf_unary("tanh"), "sqrt": _tf_unary("sqrt"), "abs": _tf_unary("abs"),
    "sinh": _tf_unary("sinh"), "cosh": _tf_unary("cosh"), "sigmoid": _tf_unary("sigmoid"),
    "reciprocal": _tf_unary("recip"), "neg": _tf_unary("neg"), "negative": _tf_unary("neg"),
    "absolute": _tf_unary("abs"),
    "square": lambda x: x * x,
    "add": _tf_bin("add"), "sub": _tf_bin("sub"), "subtract": _tf_bin("sub"), "mul": _tf_bin("mul"),
    "multiply": _tf_bin("mul"), "div": _tf_bin("div"), "divide": _tf_bin("div"), "true_divide": _tf_bin("div"),
    "pow": _tf_pow,
    "ones_like": _tf_like(1.0), "zeros_like": _tf_like(0.0), "full_like": _tf_full_like,
    "clone": lambda x, **k: x.clone() if isinstance(x, Sym) else x,
    "cat": _tf_cat, "concat": _tf_cat, "sum": _tf_sum, "mean": _batch_mean,
    "mse_loss": _tf_mse_loss, "l1_loss": _tf_l1_loss,
}


# torch.special.* arrives as "special_<name>"
for _n in ("expit", "erf", "erfc", "log1p", "expm1", "sinc", "exp2", "xlogy", "xlog1py", "logit", "round", "softmax_none"):
    if _n in _TORCH_FUNCS:
        _TORCH_FUNCS["special_" + _n] = _TORCH_FUNCS[_n]


def captured_unchanged(g):
    """Are the one-element tensors the trace baked in as numbers (Graph.captured) still what they were?  By version counter AND
    -- host tensors -- by value: ``nu.data.mul_(0.7)`` does not bump the counter.  (Device tensors by counter only
    here; _pystate.StateWatch compares their content.)"""
    for t, version, val in g.captured:
        if t._version != version or (t.device.type == "cpu" and t.item() != val and val == val):
            return False
    return True


def is_sym(x):
    return isinstance(x, (Sym, SymMat))


def sym_diff(u, t, order=1):
    """``diff`` on traced values: symbolic d^order u / dt^order.  ``t`` must be one of the batch coordinates
    (the reference differentiates w.r.t. the sampled coordinate tensors, neurodiffeq.py:22)."""
    if not isinstance(t, Sym):
        raise TraceUnsupported("diff(u, t): t must be a traced coordinate")
    g = t.g
    nt = g.nodes[t.i]
    if nt[0] != "coord":
        raise TraceUnsupported("diff(u, t): t must be a batch coordinate, not an expression")
    if not t.leaf:
        # x + 0.0, x * 1.0, x.clone(), x.view(-1, 1): the arithmetic folds them to the coordinate's node, but in torch each is
        # a NEW tensor -- the reference gives zeros if u was computed from x (neurodiffeq.py:23-24), d u / d t if it was
        # computed from t.  A trace cannot tell the two apart: refuse (-> composite path), as for `2.0 * x`
        raise TraceUnsupported("diff(u, t): t is a tensor derived from a batch coordinate (x + 0.0, x.clone(), x.view(...)), "
                               "not the coordinate itself")
    if not isinstance(u, Sym):
        # a python scalar / constant: derivative is zero, like the reference's "unused" branch (neurodiffeq.py:23-24)
        return Sym(g, g.const(0.0))
    if g.nodes[u.i][0] == "detach":
        raise TraceUnsupported("diff() of a detached value (torch.autograd.grad has no path to differentiate along)")
    e = u.i
    for _ in range(int(order)):
        e = g.diff(e, nt[1])
    return Sym(g, e)
