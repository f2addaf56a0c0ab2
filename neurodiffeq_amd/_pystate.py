"""Which Python state can a traced callable see, and has it changed?

The reference re-evaluates the user's ``diff_eqs`` and the conditions every batch (``solvers.py:369-395``), so a viscosity
held in a dict, a Reynolds number ramped by a callback or a boundary value stored on a condition object take effect at the
next epoch.  The fused path executes those callables ONCE, on symbolic columns: every Python float they read becomes a
literal of the generated kernel.  ``StateWatch`` records, at trace time, the leaves of Python state the callables can
reach -- closure cells, the module globals their code names, default arguments, attributes of bound ``self`` objects and
of the condition objects (and the plain class attributes behind them), up to six container levels deep -- as (where, expected stamp) entries and compiles them into
ONE checker function (a chain of ``and``-ed comparisons, ~0.05 us per entry: the native epoch is host-bound at the headline
size, a microsecond here is 4 % of the step).  ``dirty()`` runs it: nothing for the usual stateless lambda.  A dirty watch
is not yet a changed equation; the solver then re-traces (``program.eq_probe``: the graph is hash-consed, so an unchanged
system returns the same node ids) and only a different trace makes it rebuild the kernels (the build cache is keyed by
generated source).

The walk is a heuristic with bounded depth: state it cannot see (a value fetched through another module's function, a
container nested deeper than ``max_depth``) is covered by the solver's periodic unconditional re-trace.
"""
import functools
import numbers
import types

import torch

_LEAF_TYPES = (numbers.Number, str, bytes, type(None))
_LIBRARY_ROOTS = ("neurodiffeq_amd", "torch", "numpy", "math", "functools", "operator", "scipy")
_MISSING = object()
_OPAQUE = (torch.Tensor, torch.nn.Module, torch.optim.Optimizer, types.ModuleType, type)


def _user_class(k):
    """A class whose plain attributes can be equation state: not a builtin, not library code, not a torch module."""
    return (isinstance(k, type) and (getattr(k, "__module__", "") or "").split(".")[0] not in _LIBRARY_ROOTS + ("builtins", "abc", "typing", "collections", "types")
            and not issubclass(k, (torch.nn.Module, torch.optim.Optimizer, BaseException)))


def _is_leaf(v):
    return isinstance(v, _LEAF_TYPES) and not isinstance(v, torch.Tensor)


class StateWatch:
    def __init__(self, roots, max_depth=6, max_items=64):
        self.entries = []          # (expression template over O[...] / M, kind, payload)
        self.objs = []             # objects the expressions index: they stay alive, an id() can never come back as another object
        self._index = {}
        self._seen = set()
        self.max_depth, self.max_items = max_depth, max_items
        for r in roots:
            self._visit(r, 0)
        self._check = self._compile()

    # ------------------------------------------------------------------ building
    def _ref(self, obj):
        i = self._index.get(id(obj))
        if i is None:
            i = self._index[id(obj)] = len(self.objs)
            self.objs.append(obj)
        return f"O[{i}]"

    def _add(self, expr, value, depth):
        """``expr``: Python source that re-reads the value from the kept objects."""
        if _is_leaf(value):
            if value != value:                        # nan: only its type is pinned
                self.entries.append(f"type({expr}) is {self._ref(type(value))}")
            else:
                self.entries.append(f"((v := {expr}) == {self._ref(value)} and type(v) is {self._ref(type(value))})")
            return
        if isinstance(value, torch.Tensor):
            self.entries.append(f"((v := {expr}) is {self._ref(value)} and v._version == {value._version})")
            return
        try:
            import numpy as np
            if isinstance(value, np.ndarray):
                if value.size <= 64:
                    self.entries.append(f"((v := {expr}) is {self._ref(value)} and v.tobytes() == {self._ref(value.tobytes())})")
                else:
                    self.entries.append(f"({expr}) is {self._ref(value)}")
                return
        except Exception:  # pragma: no cover
            pass
        self.entries.append(f"({expr}) is {self._ref(value)}")
        self._visit(value, depth + 1)

    def _visit(self, v, depth):
        if isinstance(v, type) and depth <= self.max_depth and id(v) not in self._seen:
            self._seen.add(id(v))
            self._class(v, self._ref(v), depth)         # `class Cfg: nu = 0.1` used as a namespace
            return
        if depth > self.max_depth or id(v) in self._seen or _is_leaf(v) or isinstance(v, _OPAQUE):
            return                 # (tensors are stamped where they are referenced; modules are not state)
        self._seen.add(id(v))
        if isinstance(v, types.MethodType):
            self._visit(v.__func__, depth)
            self._object(v.__self__, depth)
        elif isinstance(v, types.FunctionType):
            if (getattr(v, "__module__", "") or "").split(".")[0] not in _LIBRARY_ROOTS:      # library code is not user state
                self._function(v, depth)
        elif isinstance(v, functools.partial):
            self._visit(v.func, depth)
            self._add(f"{self._ref(v)}.args", v.args, depth)
            self._add(f"{self._ref(v)}.keywords", v.keywords, depth)
        elif isinstance(v, dict):
            r = self._ref(v)
            self.entries.append(f"len({r}) == {len(v)}")
            for i, k in enumerate(list(v)):
                if i >= self.max_items:
                    break
                if _is_leaf(k):
                    self._add(f"{r}.get({self._ref(k)}, M)", v[k], depth)
        elif isinstance(v, list):
            r = self._ref(v)
            self.entries.append(f"len({r}) == {len(v)}")
            if len(v) <= self.max_items:
                for i in range(len(v)):
                    self._add(f"{r}[{i}]", v[i], depth)
        elif isinstance(v, tuple):
            # immutable: its leaves cannot change (the reference to the tuple is pinned by identity where it was read);
            # mutable members are state
            if len(v) <= self.max_items:
                for i in range(len(v)):
                    if not _is_leaf(v[i]) and not isinstance(v[i], _OPAQUE):
                        self._visit(v[i], depth + 1)
        elif callable(v) and not hasattr(v, "__dict__"):
            return                 # builtins
        else:
            self._object(v, depth)

    def _function(self, fn, depth):
        for cell in fn.__closure__ or ():
            try:
                value = cell.cell_contents
            except ValueError:     # empty cell
                continue
            self._add(f"{self._ref(cell)}.cell_contents", value, depth)
        names, codes = set(), [fn.__code__]
        while codes:               # the function's own code and the code of lambdas / comprehensions nested in it
            co = codes.pop()
            names.update(co.co_names)
            codes.extend(c for c in co.co_consts if isinstance(c, types.CodeType))
        g = fn.__globals__
        for name in sorted(names):
            if name in g and not isinstance(g[name], (types.ModuleType, types.BuiltinFunctionType)):
                value = g[name]
                if isinstance(value, type) and not _user_class(value):
                    continue
                if isinstance(value, types.FunctionType) and (getattr(value, "__module__", "") or "").split(".")[0] in _LIBRARY_ROOTS:
                    continue       # `diff`, `torch.sin` ...: library functions are not user state
                self._add(f"{self._ref(g)}.get({name!r}, M)", value, depth)
        if fn.__defaults__:
            self._add(f"{self._ref(fn)}.__defaults__", fn.__defaults__, depth)
        if fn.__kwdefaults__:
            self._add(f"{self._ref(fn)}.__kwdefaults__", fn.__kwdefaults__, depth)

    def _object(self, obj, depth):
        if isinstance(obj, _OPAQUE) or _is_leaf(obj):
            return
        d = getattr(obj, "__dict__", None)
        if not isinstance(d, dict):
            return
        self._seen.add(id(obj))
        own = getattr(obj, "_own_attrs", ())     # a solver's own bookkeeping (epoch counters, histories ...) is not equation state
        r = self._ref(obj)
        for i, name in enumerate(list(d)):
            if i >= 4 * self.max_items:
                break
            if name in own or name == "_own_attrs" or not name.isidentifier():
                continue
            self._add(f"getattr({r}, {name!r}, M)", d[name], depth)
        # plain values defined on the class and read through the instance (`class Eq: nu = 0.1`): watched THROUGH the
        # instance, so that an instance attribute set later, shadowing the class value, is seen as well
        self._class(type(obj), r, depth, skip=set(d) | set(own))

    def _class(self, cls, via, depth, skip=()):
        n = 0
        for k in cls.__mro__:
            if not _user_class(k):
                continue
            for name, value in list(vars(k).items()):
                if name.startswith("__") or name in skip or not name.isidentifier() or n >= self.max_items:
                    continue
                if _is_leaf(value) or isinstance(value, (dict, list, tuple, torch.Tensor)) or type(value).__module__ == "numpy":
                    n += 1
                    skip = set(skip) | {name}
                    self._add(f"getattr({via}, {name!r}, M)", value, depth)

    def _compile(self):
        if not self.entries:
            return None
        src = "def _check(O, M):\n    return (" + "\n            and ".join(self.entries) + ")\n"
        ns = {}
        exec(compile(src, "<neurodiffeq_amd._pystate>", "exec"), ns)     # noqa: S102 -- source built from indices into self.objs only
        return ns["_check"]

    # ------------------------------------------------------------------ checking
    def dirty(self):
        if self._check is None:
            return False
        try:
            return not self._check(self.objs, _MISSING)
        except Exception:   # noqa: BLE001 -- state that can no longer be read has changed
            return True

    def __len__(self):
        return len(self.entries)
