"""Which Python state can a traced callable see, and has it changed?

The reference re-evaluates the user's ``diff_eqs`` and the conditions every batch (``solvers.py:369-395``), so a viscosity
held in a dict, a Reynolds number ramped by a callback or a boundary value stored on a condition object take effect at the
next epoch.  The fused path executes those callables ONCE, on symbolic columns: every Python float they read becomes a
literal of the generated kernel.  ``StateWatch`` records, at trace time, the leaves of Python state the callables can
reach -- closure cells, the module globals their code names, attributes of the user modules they name, default arguments,
attributes (``__dict__`` and ``__slots__``) of bound ``self`` objects and of the condition objects, the plain class
attributes and the methods behind them, dicts / lists / tuples / deques / sets, small ndarrays by content and larger ones
by CRC, the buffers / parameters / attributes / submodules of the user's own ``torch.nn.Module`` objects, the referents of
weak references -- as (where, expected stamp) entries and compiles them into ONE checker function (a chain of ``and``-ed
comparisons, ~0.05 us per entry: the native epoch is host-bound at the headline size, a microsecond here is 4 % of the
step).  ``dirty()`` runs it: nothing for the usual stateless lambda.  A dirty watch is not yet a changed equation; the
solver then re-traces (``program.eq_probe``: the graph is hash-consed, so an unchanged system returns the same node ids)
and only a different trace makes it rebuild the kernels (the build cache is keyed by generated source).

**Fail-closed.**  The walk is a bounded heuristic, so it keeps a second answer next to ``dirty()``: ``complete``.  Whenever
it meets something it cannot stamp -- an object with neither ``__dict__`` nor ``__slots__``, an unknown container or
iterator, a dict / list beyond ``max_items``, nesting beyond ``max_depth``, an ndarray too large to hash every epoch, a
callable of a compiled non-library module, an object whose inspection raises, code that names a module outside a WHITELIST
of pure ones (the library roots and ``math``, ``operator``, ``collections`` ...: so ``time``, ``random``, ``os``, ``sys``,
``json``, any third-party package), a builtin outside a whitelist of pure ones (``open``, ``globals``, ``eval``, ``input`` ...),
a library function that hands out values from outside Python state (``torch.rand``, ``np.loadtxt`` ...), or code that names
the solver's own bookkeeping (``local_epoch``, ``global_epoch``, ``metrics_history``, ``lowest_loss`` ...) while a solver
object is reachable -- the reason is appended to ``incomplete``.
The solver treats an incomplete watch as "re-trace every time Python ran between two epochs" (one ``eq_probe`` per epoch,
93 - 685 us on the BASELINE systems) and keeps such a system off the multi-epoch native call; a complete watch keeps the
fast path.  Never a silently stale equation.
"""
import builtins
import collections
import dis
import functools
import numbers
import sys
import types
import weakref
import zlib

import torch

_LEAF_TYPES = (numbers.Number, str, bytes, type(None))
_LIBRARY_ROOTS = ("neurodiffeq_amd", "torch", "numpy", "math", "functools", "operator", "scipy")
_MISSING = object()
_OPAQUE = (torch.Tensor, torch.nn.Module, torch.optim.Optimizer, types.ModuleType, type)
_STDLIB = frozenset(getattr(sys, "stdlib_module_names", ())) | {"builtins"}
#: standard-library modules that compute from their arguments alone.  Code naming ANY other standard-library or third-party
#: module (a clock, an RNG, the environment, files, sockets, a database, another framework ...) can change what it computes
#: with nothing for a watch to see: a whitelist, because the list of ways to reach the outside world has no end
_PURE_STDLIB = frozenset({"math", "cmath", "operator", "functools", "itertools", "collections", "numbers", "fractions", "decimal",
                          "typing", "dataclasses", "enum", "abc", "copy", "string", "re", "warnings", "types", "statistics", "bisect",
                          "heapq", "contextlib", "builtins", "textwrap", "pprint", "reprlib", "array", "struct", "weakref"})
#: attribute / function names that hand out values from outside Python state when reached through a library module
#: (``np.random.rand``, ``torch.rand``, ``np.loadtxt``, ``torch.load`` ...) or through an object (``fh.read()``)
_VOLATILE_NAMES = frozenset({"random", "rand", "randn", "randint", "normal", "uniform", "rand_like", "randn_like", "randperm",
                             "multinomial", "bernoulli", "poisson", "choice", "shuffle", "permutation", "default_rng", "seed",
                             "manual_seed", "time", "perf_counter", "monotonic", "environ", "getenv", "now", "today", "load", "loadtxt",
                             "fromfile", "genfromtxt", "memmap", "read", "readline", "readlines", "recv"})
#: builtins that compute from their arguments alone; a callable's code naming any OTHER builtin function (``open``, ``input``,
#: ``globals``, ``locals``, ``vars``, ``eval``, ``exec``, ``__import__``, ``compile``, ``id``, ``hash`` ...) reads state this walk cannot see
_PURE_BUILTINS = frozenset({"abs", "all", "any", "bool", "callable", "complex", "dict", "divmod", "enumerate", "filter", "float",
                            "format", "frozenset", "getattr", "hasattr", "int", "isinstance", "issubclass", "iter", "len", "list",
                            "map", "max", "min", "next", "pow", "print", "range", "repr", "reversed", "round", "set", "slice", "sorted",
                            "str", "sum", "tuple", "type", "zip", "object", "super", "property", "staticmethod", "classmethod",
                            "bytes", "bytearray", "chr", "ord", "bin", "hex", "oct", "ascii", "setattr", "delattr", "memoryview",
                            "NotImplemented", "Ellipsis", "True", "False", "None", "__build_class__", "__debug__", "id", "hash"})
#: methods the arithmetic of an equation never runs (object housekeeping: printing, construction, pickling, copying); what
#: `dataclasses` generates for them names `_thread`, `id` ... and would make every dataclass of coefficients "incomplete"
_HOUSEKEEPING = frozenset({"__repr__", "__str__", "__init__", "__post_init__", "__new__", "__del__", "__setattr__", "__delattr__",
                           "__getstate__", "__setstate__", "__reduce__", "__reduce_ex__", "__init_subclass__", "__format__",
                           "__dir__", "__copy__", "__deepcopy__", "__sizeof__", "__class_getitem__", "__set_name__"})
_IMPURE_BUILTINS = frozenset(n for n in dir(builtins) if callable(getattr(builtins, n)) and n not in _PURE_BUILTINS
                             and not (isinstance(getattr(builtins, n), type) and issubclass(getattr(builtins, n), BaseException)))
#: attributes torch.nn.Module keeps in an instance's __dict__ for its own machinery (module.py): not equation state
_MODULE_INTERNALS = frozenset({"_backward_hooks", "_backward_pre_hooks", "_forward_hooks", "_forward_hooks_with_kwargs",
                               "_forward_hooks_always_called", "_forward_pre_hooks", "_forward_pre_hooks_with_kwargs",
                               "_is_full_backward_hook", "_non_persistent_buffers_set", "_state_dict_hooks",
                               "_state_dict_pre_hooks", "_load_state_dict_pre_hooks", "_load_state_dict_post_hooks",
                               "_compiled_call_impl", "_version", "_parameters", "_buffers", "_modules", "call_super_init"})
#: the solver's own bookkeeping (solvers.py:36-140 of the reference: counters and histories the fit loop advances by itself,
#: with no user code running): equations that read one of these through a reachable solver follow the epoch
SOLVER_BOOKKEEPING = frozenset({"local_epoch", "global_epoch", "_max_local_epoch", "metrics_history", "_history", "lowest_loss",
                                "_lowest_loss", "best_nets", "_best_nets", "_stop_training", "_phase", "_batch", "n_batches"})
#: names that read a tensor's VALUE into Python (a trainable scalar read this way is a literal of the kernel, not an argument)
_VALUE_READS = frozenset({"item", "tolist", "numpy", "float", "int", "bool"})
_CRC_BYTES = 64 << 10        # ndarrays up to this size are stamped by CRC-32 every epoch (~30 us at the limit); larger: incomplete


_BUILTIN_NAMES = frozenset(dir(builtins))


def _opnames(co, _cache={}):
    """Opcode names of a code object (cached: `dis` is slow and the same lambdas are walked again at every re-trace)."""
    r = _cache.get(co)
    if r is None:
        if len(_cache) > 512:
            _cache.clear()
        r = _cache[co] = frozenset(i.opname for i in dis.get_instructions(co))
    return r


def _root(module_name):
    return (module_name or "").split(".")[0]


def _user_class(k):
    """A class whose plain attributes and methods can be equation state: not a builtin, not library code (a user's own
    torch.nn.Module subclass counts: its ``forward`` is code the equations run)."""
    return (isinstance(k, type) and _root(getattr(k, "__module__", "")) not in _LIBRARY_ROOTS + ("builtins", "abc", "typing", "collections", "types")
            and not issubclass(k, (torch.optim.Optimizer, BaseException)))


def _third_party(mod):
    """An installed package (site-packages / dist-packages) as opposed to the user's own modules."""
    f = getattr(mod, "__file__", None) or ""
    return "site-packages" in f or "dist-packages" in f


def _is_leaf(v):
    return isinstance(v, _LEAF_TYPES) and not isinstance(v, torch.Tensor)


def _crc(a):
    import numpy as np
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).reshape(-1)) if a.dtype != object else None


class StateWatch:
    def __init__(self, roots, max_depth=6, max_items=64, skip_modules=()):
        """skip_modules: the solver's own networks -- their parameters are kernel arguments, re-read every launch."""
        self.entries = []          # (expression template over O[...] / M, kind, payload)
        self.objs = []             # objects the expressions index: they stay alive, an id() can never come back as another object
        self.incomplete = []       # reasons the walk could not stamp everything the callables can read (fail-closed, module docstring)
        self._index = {}
        self._seen = set()
        self._names = set()        # every name the walked code objects mention
        self._class_sizes = set()
        self._solver_seen = False
        self._trainable = []       # (expression, tensor) of requires_grad leaves: stamped by identity (their values are kernel arguments)
        self._optimizers = []      # optimisers in reach: their hyper-parameters are state iff the code names them (below)
        self._skip_modules = {id(m) for m in skip_modules}
        self._late_modules = []    # modules met as VALUES (a closure cell, an attribute, `import m` inside the function): walked
        #                            once every code object has been seen, with all the names the code mentions
        self.max_depth, self.max_items = max_depth, max_items
        for r in roots:
            self._visit(r, 0)
        for mod in self._late_modules:
            self._module(mod, self._names, 1)
        for opt in self._optimizers:
            # `opt.param_groups[0]['lr']` read by the equations: the hyper-parameters of every group are stamped (a scheduler
            # then makes the watch dirty every epoch: re-trace, value -> runtime constant).  The per-parameter STATE (Adam's
            # moments, step counts: tensors the optimiser rewrites every step) cannot be stamped: incomplete
            if self._names & {"param_groups", "defaults", "state_dict"}:
                r = self._ref(opt)
                self.entries.append(f"len({r}.param_groups) == {len(opt.param_groups)}")
                for i, group in enumerate(opt.param_groups[:self.max_items]):
                    for k, val in group.items():
                        if k != "params":
                            self._add(f"{r}.param_groups[{i}].get({k!r}, M)", val, 1)
            if self._names & {"state", "state_dict"}:
                self._fail("the equations name an optimiser's per-parameter state (rewritten by every step)")
        if self._solver_seen and self._names & SOLVER_BOOKKEEPING:
            self._fail("the equations name solver bookkeeping (" + ", ".join(sorted(self._names & SOLVER_BOOKKEEPING)) +
                       ") with a solver object in reach")
        for expr, t in self._trainable:
            if self._names & _VALUE_READS:
                # a trainable scalar the code may read by VALUE (.item(), float(...)): every optimiser step changes it
                self.entries.append(f"{expr}._version == {t._version}")
        self._check = self._compile()

    @property
    def complete(self):
        return not self.incomplete

    def _fail(self, why):
        if why not in self.incomplete and len(self.incomplete) < 8:
            self.incomplete.append(why)

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
            if value.requires_grad and value.is_leaf:
                # nn.Parameter coefficients (inverse problems): kernel ARGUMENTS, bumped by every optimiser step -- pinned by
                # identity only, or every epoch would re-trace for nothing
                self.entries.append(f"({expr}) is {self._ref(value)}")
                self._trainable.append((self._ref(value), value))
            else:
                self._tensor(expr, value)
            return
        try:
            import numpy as np
            if isinstance(value, np.ndarray):
                if value.size <= 64 and value.dtype != object:
                    self.entries.append(f"((v := {expr}) is {self._ref(value)} and v.tobytes() == {self._ref(value.tobytes())})")
                elif value.nbytes <= _CRC_BYTES and value.dtype != object:
                    self.entries.append(f"((v := {expr}) is {self._ref(value)} and v.shape == {self._ref(value.shape)} "
                                        f"and CRC(v) == {_crc(value)})")
                else:
                    self.entries.append(f"({expr}) is {self._ref(value)}")
                    self._fail(f"an ndarray of {value.nbytes} bytes (dtype {value.dtype}) is too large to compare every epoch")
                return
        except Exception:  # pragma: no cover
            pass
        self.entries.append(f"({expr}) is {self._ref(value)}")
        self._visit(value, depth + 1)

    def _tensor(self, expr, t):
        """A tensor that is not a trainable leaf: the equations read its VALUE (a one-element tensor is a literal of the
        kernel).  The version counter misses the usual ways a callback changes one -- ``nu.data.mul_(0.7)``, ``nu.data =
        ...``, writing through a ``.numpy()`` view or the ndarray the tensor was made from -- so small
        tensors are stamped by CONTENT (as small ndarrays are), larger host tensors by CRC, and what is too large to read
        every epoch makes the watch incomplete.  Per-point data columns ((N, 1), kernel INPUTS re-read every batch:
        symbolic.Graph.datacol) are pinned by identity alone."""
        r = self._ref(t)
        try:
            n, dev = t.numel(), t.device.type
            if t.dim() == 2 and t.shape[1] == 1 and n > 1 and not t.requires_grad:
                self.entries.append(f"(v := {expr}) is {r} and v.shape == {self._ref(t.shape)}")
                return
            if n <= 64 and t.layout == torch.strided:
                # (a device tensor costs a synchronising copy per check: the price of reading a coefficient the way the
                # reference does every batch; keep coefficients on the host or make them nn.Parameters to avoid it)
                self.entries.append(f"((v := {expr}) is {r} and v.shape == {self._ref(t.shape)} and v.tolist() == {self._ref(t.tolist())})")
                return
            if dev == "cpu" and t.layout == torch.strided and n * t.element_size() <= _CRC_BYTES and \
                    t.dtype in (torch.float32, torch.float64, torch.int64, torch.int32, torch.uint8, torch.bool, torch.int16, torch.int8):
                crc = _crc(t.detach().numpy())
                self.entries.append(f"((v := {expr}) is {r} and v.shape == {self._ref(t.shape)} and CRC(v.detach().numpy()) == {crc})")
                return
        except Exception:  # noqa: BLE001 -- a tensor subclass / layout that cannot be read: identity + version + said so
            pass
        self.entries.append(f"((v := {expr}) is {r} and v._version == {t._version})")
        self._fail(f"a {t.device.type} tensor of {t.numel()} elements (dtype {t.dtype}) cannot be compared by content every epoch")

    def _visit(self, v, depth):
        try:
            self._visit_unguarded(v, depth)
        except Exception as e:   # noqa: BLE001 -- e.g. a __getattr__ that raises something else than AttributeError
            self._fail(f"an object of type {type(v).__module__}.{type(v).__qualname__} could not be inspected ({type(e).__name__})")

    def _visit_unguarded(self, v, depth):
        if isinstance(v, torch.nn.Module) and _root(type(v).__module__) not in _LIBRARY_ROOTS:
            if id(v) not in self._seen:
                if depth > self.max_depth:
                    self._fail(f"state nested deeper than {self.max_depth} levels")
                else:
                    self._seen.add(id(v))
                    self._torch_module(v, depth)
            return
        if isinstance(v, type) and depth <= self.max_depth and id(v) not in self._seen:
            self._seen.add(id(v))
            self._class(v, self._ref(v), depth)         # `class Cfg: nu = 0.1` used as a namespace
            return
        if isinstance(v, torch.optim.Optimizer):
            if id(v) not in self._seen:
                self._seen.add(id(v))
                self._optimizers.append(v)           # stamped once the walk knows which names the code mentions (_finish)
            return
        if isinstance(v, torch.nn.Module) and id(v) not in self._seen and id(v) not in self._skip_modules:
            # a LIBRARY module (nn.BatchNorm1d, an FCNN that is not one of the solver's networks): its code is not user state,
            # its numbers are -- `bn.eps`, `bn.running_mean`, a frozen weight the equations read
            self._seen.add(id(v))
            if depth > self.max_depth:
                self._fail(f"state nested deeper than {self.max_depth} levels")
            else:
                self._torch_module(v, depth, methods=False)
            return
        if isinstance(v, types.ModuleType):
            # `import time` in the enclosing function, `self.np = numpy`, a dict of modules: the module is code, but WHICH
            # module it is decides whether the values it hands out are Python state (_module, after the walk)
            if id(v) not in self._seen:
                self._seen.add(id(v))
                self._late_modules.append(v)
            return
        if id(v) in self._seen or _is_leaf(v) or isinstance(v, _OPAQUE):
            return                 # (tensors are stamped where they are referenced; modules are not state)
        if depth > self.max_depth:
            self._fail(f"state nested deeper than {self.max_depth} levels")
            return
        self._seen.add(id(v))
        if isinstance(v, types.MethodType):
            self._visit(v.__func__, depth)
            self._object(v.__self__, depth)
        elif isinstance(v, types.FunctionType):
            if _root(getattr(v, "__module__", "")) not in _LIBRARY_ROOTS:      # library code is not user state
                self._function(v, depth)
        elif isinstance(v, functools.partial):
            self._visit(v.func, depth)
            self._add(f"{self._ref(v)}.args", v.args, depth)
            self._add(f"{self._ref(v)}.keywords", v.keywords, depth)
            for name, value in list((getattr(v, "__dict__", None) or {}).items()):
                if name.isidentifier():
                    self._add(f"getattr({self._ref(v)}, {name!r}, M)", value, depth)
        elif isinstance(v, dict):
            r = self._ref(v)
            self.entries.append(f"len({r}) == {len(v)}")
            if len(v) > self.max_items:
                self._fail(f"a dict of {len(v)} entries (more than {self.max_items})")
            for i, k in enumerate(list(v)):
                if i >= self.max_items:
                    break
                self._add(f"{r}.get({self._ref(k)}, M)", v[k], depth)       # (any hashable key: looked up through the kept object)
        elif isinstance(v, (list, collections.deque)):
            r = self._ref(v)
            self.entries.append(f"len({r}) == {len(v)}")
            if len(v) <= self.max_items:
                for i in range(len(v)):
                    self._add(f"{r}[{i}]", v[i], depth)
            else:
                self._fail(f"a {type(v).__name__} of {len(v)} items (more than {self.max_items})")
        elif isinstance(v, tuple):
            # immutable: its leaves cannot change (the reference to the tuple is pinned by identity where it was read);
            # mutable members are state
            if len(v) <= self.max_items:
                for i in range(len(v)):
                    if isinstance(v[i], (torch.Tensor, torch.nn.Module, torch.optim.Optimizer)):
                        self._add(f"{self._ref(v)}[{i}]", v[i], depth)      # (a tensor's CONTENT is state wherever it sits)
                    elif not _is_leaf(v[i]) and not isinstance(v[i], _OPAQUE):
                        self._visit(v[i], depth + 1)
            elif not all(_is_leaf(x) for x in v):
                self._fail(f"a tuple of {len(v)} items (more than {self.max_items})")
        elif isinstance(v, (set, frozenset)):
            if all(_is_leaf(x) for x in v) and len(v) <= 16 * self.max_items:
                if isinstance(v, set):
                    self.entries.append(f"{self._ref(v)} == {self._ref(frozenset(v))}")
            else:
                self._fail("a set with non-scalar members")
        elif isinstance(v, (types.GeneratorType, types.CoroutineType, types.AsyncGeneratorType)) or \
                (hasattr(v, "__next__") and not hasattr(v, "__dict__")):
            self._fail(f"an iterator ({type(v).__name__}): what it yields next cannot be compared")
        elif isinstance(v, weakref.ReferenceType):
            target = v()
            if target is not None:
                self._add(f"{self._ref(v)}()", target, depth)
        elif callable(v) and not hasattr(v, "__dict__") and not hasattr(type(v), "__slots__"):
            # builtins and compiled callables: library ones and the pure part of the standard library are not user state; any
            # other compiled function computes from state this walk cannot read
            mod = _root(getattr(v, "__module__", None) or getattr(getattr(v, "__self__", None), "__module__", None)
                        or type(v).__module__ or "builtins")
            name = getattr(v, "__name__", "")
            owner = getattr(v, "__self__", None)
            if owner is not None and not isinstance(owner, (types.ModuleType, type)):
                # `get = cfg.get`, `at = values.__getitem__`: a bound builtin method reads its owner
                self._add(f"{self._ref(v)}.__self__", owner, depth)
            if mod == "builtins" and name in _IMPURE_BUILTINS:
                self._fail(f"the builtin {name}() (reads state that is not the equations')")
            elif mod in _STDLIB and mod not in _PURE_STDLIB:
                self._fail(f"a function of module '{mod}' (values that are not Python state)")
            elif mod not in _LIBRARY_ROOTS and mod not in _STDLIB:
                self._fail(f"a compiled callable of module '{mod}'")
        else:
            self._object(v, depth)

    def _function(self, fn, depth):
        for cell in fn.__closure__ or ():
            try:
                value = cell.cell_contents
            except ValueError:     # empty cell
                continue
            self._add(f"{self._ref(cell)}.cell_contents", value, depth)
        names, codes, imports = set(), [fn.__code__], set()
        # (fn.__code__ = other.__code__; a function attribute set later)
        self.entries.append(f"((v := {self._ref(fn)}).__code__ is {self._ref(fn.__code__)} and len(v.__dict__) == {len(vars(fn))})")
        while codes:               # the function's own code and the code of lambdas / comprehensions nested in it
            co = codes.pop()
            names.update(co.co_names)
            codes.extend(c for c in co.co_consts if isinstance(c, types.CodeType))
            if "IMPORT_NAME" in _opnames(co):
                imports.update(i.argval for i in dis.get_instructions(co) if i.opname == "IMPORT_NAME" and isinstance(i.argval, str))
        self._names |= names
        for modname in sorted(imports):                      # `import cfg` INSIDE the function: sys.modules is where it comes from
            mod = sys.modules.get(modname)
            if mod is None:
                self._fail(f"the equations import module '{modname}' when they run")
            else:
                self._visit(mod, depth)
                top = sys.modules.get(_root(modname))
                if top is not None:
                    self._visit(top, depth)
        for name, value in list(vars(fn).items()):           # function attributes: `eq.nu = 0.1; def eq(u, t): return eq.nu * u`
            if name != "__wrapped__" and name.isidentifier():
                self._add(f"getattr({self._ref(fn)}, {name!r}, M)", value, depth)
        wrapped = vars(fn).get("__wrapped__")
        if wrapped is not None:
            self._visit(wrapped, depth)
        g = fn.__globals__
        hot = sorted(n for n in names & _IMPURE_BUILTINS if n not in g and n not in fn.__code__.co_varnames)
        if hot:
            self._fail(f"the equations call the builtin(s) {hot} (state that is not the equations')")
        extra = 0
        for name in sorted(names):
            if name not in g:
                # not a global NOW: an attribute / method name (most), a standard builtin -- or a name somebody put into the
                # builtins module, or a global a callback defines later (hasattr-style switches): pinned as "absent" / by value
                if name not in _BUILTIN_NAMES and extra < 32 and name.isidentifier() and not name.startswith("__"):
                    extra += 1
                    patched = vars(builtins).get(name, _MISSING)
                    if patched is not _MISSING:
                        self._add(f"vars({self._ref(builtins)}).get({name!r}, M)", patched, depth)
                    else:
                        self.entries.append(f"{self._ref(g)}.get({name!r}, M) is M and vars({self._ref(builtins)}).get({name!r}, M) is M")
                continue
            value = g[name]
            if isinstance(value, types.ModuleType):
                self._module(value, names, depth)
                continue
            if isinstance(value, types.BuiltinFunctionType):
                self._visit(value, depth)
                continue
            if isinstance(value, type) and not _user_class(value):
                continue
            if isinstance(value, types.FunctionType) and _root(getattr(value, "__module__", "")) in _LIBRARY_ROOTS:
                continue       # `diff`, `torch.sin` ...: library functions are not user state
            self._add(f"{self._ref(g)}.get({name!r}, M)", value, depth)
        if fn.__defaults__:
            self._add(f"{self._ref(fn)}.__defaults__", fn.__defaults__, depth)
        if fn.__kwdefaults__:
            self._add(f"{self._ref(fn)}.__kwdefaults__", fn.__kwdefaults__, depth)

    def _module(self, mod, names, depth):
        """``import cfg`` ... ``cfg.nu * u``: the attributes of a USER module that the code names are state (the attribute
        names sit in ``co_names`` next to the module's own); library and standard-library modules are code, except the
        ones that hand out values from outside Python state."""
        root = _root(getattr(mod, "__name__", ""))
        if root in _LIBRARY_ROOTS or root in _PURE_STDLIB:
            hot = names & _VOLATILE_NAMES
            if hot:
                self._fail(f"the equations name {sorted(hot)} of module '{root}' (values that are not Python state)")
            return
        if root in _STDLIB or _third_party(mod):
            self._fail(f"the equations name module '{root}' (values that are not Python state)")
            return
        ns, r = vars(mod), self._ref(mod)
        self.entries.append(f"len(vars({r})) == {len(ns)}")        # (`getattr(cfg, "late", 1.0)`: the name is a string constant)
        absent = [n for n in sorted(names) if n not in ns and n.isidentifier() and not n.startswith("__")][:32]
        if absent:                 # `getattr(cfg, "nu", 1.0)` / `hasattr(cfg, "nu")`: an attribute a callback sets later
            self.entries.append("(" + " and ".join(f"getattr({r}, {n!r}, M) is M" for n in absent) + ")")
        for name in sorted(names):
            if name in ns and not name.startswith("__"):
                value = ns[name]
                if isinstance(value, types.ModuleType):
                    if id(value) not in self._seen:
                        self._seen.add(id(value))
                        self._module(value, names, depth)
                    continue
                self._add(f"getattr({r}, {name!r}, M)", value, depth)

    def _torch_module(self, m, depth, methods=True):
        """A user's own torch.nn.Module the callables reach (an operator object used as ``diff_eqs``, a coefficient model
        in a closure): its buffers and parameters (tensors: identity + version, trainable leaves by identity), its plain
        attributes, its submodules, and the methods of its class."""
        r = self._ref(m)
        d = vars(m)
        for kind in ("_parameters", "_buffers"):
            store = d.get(kind) or {}
            self.entries.append(f"len(vars({r})[{kind!r}]) == {len(store)}")
            for name, t in list(store.items())[:self.max_items]:
                self._add(f"vars({r})[{kind!r}].get({name!r}, M)", t, depth)
            if len(store) > self.max_items:
                self._fail(f"a module with {len(store)} {kind[1:]}")
        for name, child in list((d.get("_modules") or {}).items())[:self.max_items]:
            self.entries.append(f"vars({r})['_modules'].get({name!r}, M) is {self._ref(child)}")
            if isinstance(child, torch.nn.Module) and id(child) not in self._seen:
                self._seen.add(id(child))
                if depth + 1 > self.max_depth:
                    self._fail(f"state nested deeper than {self.max_depth} levels")
                else:
                    self._torch_module(child, depth + 1, methods and _root(type(child).__module__) not in _LIBRARY_ROOTS)
        for name, value in list(d.items()):
            if name in _MODULE_INTERNALS or not name.isidentifier():
                continue
            self._add(f"getattr({r}, {name!r}, M)", value, depth)
        if methods:
            self._class(type(m), r, depth, skip=set(d))

    def _object(self, obj, depth):
        if isinstance(obj, (torch.nn.Module, torch.optim.Optimizer)):
            return self._visit(obj, depth)
        if isinstance(obj, _OPAQUE) or _is_leaf(obj):
            return
        d = getattr(obj, "__dict__", None)
        slots = []
        for k in type(obj).__mro__:
            ks = vars(k).get("__slots__", ())
            slots += [s for s in ((ks,) if isinstance(ks, str) else tuple(ks)) if s not in ("__dict__", "__weakref__")]
        if not isinstance(d, dict) and not slots:
            self._fail(f"an object of type {type(obj).__module__}.{type(obj).__qualname__} with neither __dict__ nor __slots__")
            return
        self._seen.add(id(obj))
        own = getattr(obj, "_own_attrs", ())     # a solver's own bookkeeping (epoch counters, histories ...) is not equation state
        if own:
            self._solver_seen = True             # ... unless the code NAMES it: checked once the walk is over (module docstring)
        r = self._ref(obj)
        if isinstance(d, dict) and not own:
            # `getattr(cfg, "nu", 1.0)` / `hasattr(cfg, "nu")` / `vars(cfg)`: an attribute that does not exist yet (a solver adds
            # attributes of its own as it runs: its bookkeeping is handled by name, above)
            self.entries.append(f"len(vars({r})) == {len(d)}")
        d = d if isinstance(d, dict) else {}
        if len(d) > 4 * self.max_items:
            self._fail(f"an object with {len(d)} attributes")
        for i, name in enumerate(list(d)):
            if i >= 4 * self.max_items:
                break
            if name in own or name == "_own_attrs" or not name.isidentifier():
                continue
            self._add(f"getattr({r}, {name!r}, M)", d[name], depth)
        for name in slots:
            if name.isidentifier() and name not in d:
                self._add(f"getattr({r}, {name!r}, M)", getattr(obj, name, _MISSING), depth)
        # plain values defined on the class and read through the instance (`class Eq: nu = 0.1`): watched THROUGH the
        # instance, so that an instance attribute set later, shadowing the class value, is seen as well
        self._class(type(obj), r, depth, skip=set(d) | set(own) | set(slots))

    def _class(self, cls, via, depth, skip=()):
        n = 0
        for k in cls.__mro__:
            if not _user_class(k):
                continue
            if id(k) not in self._class_sizes:
                self._class_sizes.add(id(k))
                self.entries.append(f"len(vars({self._ref(k)})) == {len(vars(k))}")     # (a class attribute / method added later)
            for name, value in list(vars(k).items()):
                # methods (and what property / staticmethod / classmethod wrap): code the equations may run -- their closure
                # cells and the globals they name are state like the entry function's
                fn = value.fget if isinstance(value, property) else getattr(value, "__func__", value) if isinstance(value, (staticmethod, classmethod)) else value
                if isinstance(fn, types.FunctionType):
                    if name not in _HOUSEKEEPING:
                        # (the method the class holds NOW: `Eq.__call__ = other` by a callback is a change)
                        self.entries.append(f"vars({self._ref(k)}).get({name!r}, M) is {self._ref(value)}")
                        self._visit(fn, depth)
                    continue
                if name.startswith("__") or name in skip or not name.isidentifier():
                    continue
                if hasattr(type(value), "__get__") and _user_class(type(value)):
                    # a user's descriptor object: read through the instance it computes from ITS state
                    self._add(f"vars({self._ref(k)}).get({name!r}, M)", value, depth)
                    continue
                if _is_leaf(value) or isinstance(value, (dict, list, tuple, set, collections.deque, torch.Tensor)) or type(value).__module__ == "numpy":
                    if n >= self.max_items:
                        self._fail(f"class {k.__qualname__} has more than {self.max_items} plain attributes")
                        break
                    n += 1
                    skip = set(skip) | {name}
                    self._add(f"getattr({via}, {name!r}, M)", value, depth)

    def _compile(self):
        if not self.entries:
            return None
        src = "def _check(O, M, CRC):\n    return (" + "\n            and ".join(self.entries) + ")\n"
        ns = {}
        exec(compile(src, "<neurodiffeq_amd._pystate>", "exec"), ns)     # noqa: S102 -- source built from indices into self.objs only
        return ns["_check"]

    # ------------------------------------------------------------------ checking
    def dirty(self):
        if self._check is None:
            return False
        try:
            return not self._check(self.objs, _MISSING, _crc)
        except Exception:   # noqa: BLE001 -- state that can no longer be read has changed
            return True

    def __len__(self):
        return len(self.entries)
