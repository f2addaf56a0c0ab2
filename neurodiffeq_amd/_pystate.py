"""Which Python state can a traced callable see, and has it changed?

The reference re-evaluates the user's ``diff_eqs`` and the conditions every batch (``solvers.py:369-395``), so a viscosity
held in a dict, a Reynolds number ramped by a callback or a boundary value stored on a condition object take effect at the
next epoch.  The fused path executes those callables ONCE, on symbolic columns: every Python float they read becomes a
literal of the generated kernel.  ``StateWatch`` records, at trace time, the leaves of Python state the callables can
reach -- closure cells, the module globals their code names, default arguments, attributes of bound ``self`` objects and
of the condition objects, one or two container levels deep -- as (getter, expected value) pairs.  ``dirty()`` walks that
list: a handful of comparisons per epoch, nothing for the usual stateless lambda.  A dirty watch is not yet a changed
equation; the solver then re-traces (``program.eq_probe``: the graph is hash-consed, so an unchanged system returns the
same node ids) and only a different trace makes it rebuild the kernels (the build cache is keyed by generated source).

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


def _is_leaf(v):
    return isinstance(v, _LEAF_TYPES) and not isinstance(v, torch.Tensor)


def _stamp(v):
    """What is remembered of a value: leaves by (type, value); tensors / arrays by identity + version or content; anything
    else (containers, objects, functions) by identity -- their contents get entries of their own."""
    if _is_leaf(v):
        return ("leaf", type(v), v)
    if isinstance(v, torch.Tensor):
        return ("tensor", id(v), v._version)
    try:
        import numpy as np
        if isinstance(v, np.ndarray):
            return ("ndarray", id(v), v.tobytes() if v.size <= 64 else None)
    except Exception:  # pragma: no cover
        pass
    return ("id", id(v))


class StateWatch:
    def __init__(self, roots, max_depth=3, max_items=64):
        self.entries = []          # (getter, stamp)
        self.keep = []             # visited objects stay alive: an id() can never come back as another object
        self._seen = set()
        self.max_depth, self.max_items = max_depth, max_items
        for r in roots:
            self._visit(r, 0)

    # ------------------------------------------------------------------ building
    def _add(self, getter, depth):
        try:
            v = getter()
        except Exception:   # noqa: BLE001 -- unreadable state is simply not watched
            return
        self.entries.append((getter, _stamp(v)))
        if not _is_leaf(v):
            self._visit(v, depth + 1)

    def _visit(self, v, depth):
        if depth > self.max_depth or id(v) in self._seen or _is_leaf(v):
            return
        if isinstance(v, (torch.Tensor, torch.nn.Module, torch.optim.Optimizer, types.ModuleType, type)):
            return                 # tensors are stamped where they are referenced; modules / classes are not state
        self._seen.add(id(v))
        self.keep.append(v)
        if isinstance(v, types.MethodType):
            self._visit(v.__func__, depth)
            self._object(v.__self__, depth)
        elif isinstance(v, types.FunctionType):
            if (getattr(v, "__module__", "") or "").split(".")[0] not in _LIBRARY_ROOTS:      # library code is not user state
                self._function(v, depth)
        elif isinstance(v, functools.partial):
            self._visit(v.func, depth)
            self._add(lambda v=v: v.args, depth)
            self._add(lambda v=v: v.keywords, depth)
        elif isinstance(v, dict):
            self.entries.append((lambda v=v: len(v), ("leaf", int, len(v))))
            for i, k in enumerate(list(v)):
                if i >= self.max_items:
                    break
                if _is_leaf(k):
                    self._add(lambda v=v, k=k: v.get(k, _MISSING), depth)
        elif isinstance(v, (list, tuple)):
            self.entries.append((lambda v=v: len(v), ("leaf", int, len(v))))
            if len(v) <= self.max_items:
                for i in range(len(v)):
                    self._add(lambda v=v, i=i: v[i] if i < len(v) else _MISSING, depth)
        elif callable(v) and not hasattr(v, "__dict__"):
            return                 # builtins
        else:
            self._object(v, depth)

    def _function(self, fn, depth):
        for cell in fn.__closure__ or ():
            self._add(lambda c=cell: c.cell_contents, depth)
        names, codes = set(), [fn.__code__]
        while codes:               # the function's own code and the code of lambdas / comprehensions nested in it
            co = codes.pop()
            names.update(co.co_names)
            codes.extend(c for c in co.co_consts if isinstance(c, types.CodeType))
        g = fn.__globals__
        for name in sorted(names):
            if name in g and not isinstance(g[name], (types.ModuleType, type, types.BuiltinFunctionType)):
                self._add(lambda g=g, name=name: g.get(name, _MISSING), depth)
        if fn.__defaults__:
            self._add(lambda fn=fn: fn.__defaults__, depth)
        if fn.__kwdefaults__:
            self._add(lambda fn=fn: fn.__kwdefaults__, depth)

    def _object(self, obj, depth):
        if isinstance(obj, (torch.Tensor, torch.nn.Module, torch.optim.Optimizer, types.ModuleType, type)) or _is_leaf(obj):
            return
        d = getattr(obj, "__dict__", None)
        if not isinstance(d, dict):
            return
        if id(obj) not in self._seen:
            self._seen.add(id(obj))
            self.keep.append(obj)
        own = getattr(obj, "_own_attrs", ())     # a solver's own bookkeeping (epoch counters, histories ...) is not equation state
        for i, name in enumerate(list(d)):
            if i >= 4 * self.max_items:
                break
            if name in own or name == "_own_attrs":
                continue
            self._add(lambda obj=obj, name=name: getattr(obj, name, _MISSING), depth)

    # ------------------------------------------------------------------ checking
    def dirty(self):
        for getter, stamp in self.entries:
            try:
                v = getter()
            except Exception:   # noqa: BLE001
                return True
            if stamp[0] == "leaf":
                if type(v) is not stamp[1] or (v != stamp[2] and not (v != v and stamp[2] != stamp[2])):     # (nan stays nan)
                    return True
            elif _stamp(v) != stamp:
                return True
        return False

    def __len__(self):
        return len(self.entries)
