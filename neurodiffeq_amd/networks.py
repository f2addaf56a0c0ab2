"""Networks of the PINN hot path -- same constructor surface as the reference (neurodiffeq/networks.py:6-70,
142-209) so existing scripts keep working; ``FCNN`` is a real ``nn.Module`` (``.NN`` is the ``Sequential``; its
parameters are ordinary ``nn.Parameter`` objects, which the optimiser, ``deepcopy`` and checkpoints rely on).

What is new is :func:`describe`: it recognises the modules the gfx950 kernels can run (FCNN / Resnet with one hidden
width of up to 64 units on every layer; tanh, sin, sigmoid, Swish, APTx -- the last two with their default or with
trainable parameters; any number of output units) and :class:`FlatParams`, which re-homes the parameters as views of
one flat fp32 buffer -- linear layers in torch order, then the skip weights, then the activation parameters: the
layout ``ndq_mlp_jet_fwd/bwd`` and ``ndq_adam_step`` consume.
"""
import collections
import warnings
import weakref

import torch
import torch.nn as nn

from . import _lib

# ---- what a network IS can change between epochs (solvers.py:496-497 runs callbacks there): a layer replaced, a weight
# re-assigned, a weight-norm / parametrization registered, a forward hook added.  The kernels read a flat copy of the parameters
# describe() listed when the system was built, so every such change has to reach the solver.  Checking module trees every epoch
# would cost microseconds of a 20 us step; instead the change itself raises a flag: torch's global registration hooks (a
# parameter / submodule / buffer set on a tracked module) and hook dictionaries that count their own edits bump
# STRUCTURE[0], which is part of the solver's system key and of the custom-op seam's cache key.
STRUCTURE = [0]
_tracked = {}               # id(module) -> weak reference (by id: a user's module may define __eq__ and be unhashable)
_HOOK_DICTS = ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")


class _NotifyingHooks(collections.OrderedDict):
    """A module's hook dictionary that bumps STRUCTURE[0] whenever a hook is added or removed."""

    def __setitem__(self, k, v):
        STRUCTURE[0] += 1
        super().__setitem__(k, v)

    def __delitem__(self, k):
        STRUCTURE[0] += 1
        super().__delitem__(k)

    def pop(self, *a, **k):
        STRUCTURE[0] += 1
        return super().pop(*a, **k)

    def popitem(self, *a, **k):
        STRUCTURE[0] += 1
        return super().popitem(*a, **k)

    def clear(self):
        STRUCTURE[0] += 1
        super().clear()

    def update(self, *a, **k):
        STRUCTURE[0] += 1
        super().update(*a, **k)

    def setdefault(self, *a, **k):
        STRUCTURE[0] += 1
        return super().setdefault(*a, **k)

    def __reduce__(self):           # copies and pickles are plain dictionaries (nothing outside this process names the class)
        return (collections.OrderedDict, (), None, None, iter(list(self.items())))


def _is_tracked(module):
    ref = _tracked.get(id(module))
    return ref is not None and ref() is module


def _on_registration(module, name, value):
    if _is_tracked(module):
        STRUCTURE[0] += 1
    return None


def track_structure(net):
    """From now on a structural change of ``net`` (any module of its tree) bumps STRUCTURE[0]."""
    from torch.nn.modules import module as M
    if not getattr(M, "_ndq_registration_hooks", False):
        M.register_module_parameter_registration_hook(_on_registration)
        M.register_module_module_registration_hook(_on_registration)
        M.register_module_buffer_registration_hook(_on_registration)
        M._ndq_registration_hooks = True
    for m in net.modules():
        if not _is_tracked(m):
            key = id(m)
            _tracked[key] = weakref.ref(m, lambda _, key=key: _tracked.pop(key, None))
        for name in _HOOK_DICTS:
            d = m.__dict__.get(name)
            if d is not None and not isinstance(d, _NotifyingHooks):
                nd = _NotifyingHooks()
                collections.OrderedDict.update(nd, d)
                m.__dict__[name] = nd


def _hooked(*modules):
    for m in modules:
        d = m.__dict__
        for name in _HOOK_DICTS:
            if d.get(name):
                return True
    return False


def _forward_is(m, *owners):
    """Does ``m`` run the forward of one of ``owners`` (classes of this package, of torch, or the reference package's class of the
    same name)?  A subclass that overrides forward computes something the kernels do not."""
    f = type(m).forward
    for o in owners:
        if f is o.forward:
            return True
        if o.__module__ == __name__ and getattr(f, "__module__", "") == "neurodiffeq.networks" and \
                getattr(f, "__qualname__", "") == o.__name__ + ".forward":
            return True
    return False


class SinActv(nn.Module):
    """sin activation (reference: networks.py:142-152)."""

    def forward(self, input_):
        return torch.sin(input_)


class Swish(nn.Module):
    """x * sigmoid(beta x) (reference: networks.py:155-174)."""

    def __init__(self, beta=1.0, trainable=False):
        super().__init__()
        self.trainable = trainable
        self.beta = nn.Parameter(torch.tensor(float(beta))) if trainable else float(beta)

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x)


class APTx(nn.Module):
    """(alpha + tanh(beta x)) * gamma * x (reference: networks.py:177-209)."""

    def __init__(self, alpha=1.0, beta=1.0, gamma=0.5, trainable=False):
        super().__init__()
        self.trainable = trainable
        mk = (lambda v: nn.Parameter(torch.tensor(float(v)))) if trainable else float
        self.alpha, self.beta, self.gamma = mk(alpha), mk(beta), mk(gamma)

    def forward(self, x):
        return (self.alpha + torch.tanh(self.beta * x)) * self.gamma * x


def _traced_call(net, t):
    """``net(torch.cat([x, y], 1))`` written out by hand inside a condition's ``enforce`` (or a ``compute_func_val``) while the
    solver traces the system: the network's symbol at those coordinate columns, exactly what ``BaseCondition.enforce`` arrives
    at (conditions.py:52-55).  Anything but the solver's own network at plain coordinate columns raises TraceUnsupported."""
    from .symbolic import Sym, SymMat, current_graph
    if isinstance(t, Sym):
        return current_graph().net_symbol(net, [t])
    if isinstance(t, SymMat):
        return current_graph().net_symbol(net, list(t.cols))
    return None


class FCNN(nn.Module):
    """Fully connected network ``Linear -> actv -> ... -> Linear`` (reference: networks.py:26-70).

    Same arguments and defaults: ``hidden_units=(32, 32)``, ``actv=nn.Tanh``; the deprecated
    ``n_hidden_units`` / ``n_hidden_layers`` pair is still understood (with the reference's FutureWarning)."""

    def __init__(self, n_input_units=1, n_output_units=1, n_hidden_units=None, n_hidden_layers=None,
                 actv=nn.Tanh, hidden_units=None):
        super().__init__()
        if n_hidden_units is None and n_hidden_layers is not None:
            n_hidden_units = 32
        elif n_hidden_units is not None and n_hidden_layers is None:
            n_hidden_layers = 1
        if n_hidden_units is not None or n_hidden_layers is not None:
            if hidden_units is None:
                hidden_units = tuple(n_hidden_units for _ in range(n_hidden_layers + 1))
                warnings.warn(f"`n_hidden_units` and `n_hidden_layers` are deprecated, "
                              f"pass `hidden_units={hidden_units}` instead", FutureWarning)
            else:
                warnings.warn(f"Ignoring `n_hidden_units` and `n_hidden_layers` in favor of "
                              f"`hidden_units={hidden_units}`", FutureWarning)
        if hidden_units is None:
            hidden_units = (32, 32)
        hidden_units = tuple(hidden_units)
        widths = (n_input_units,) + hidden_units
        mods = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(fan_in, fan_out), actv()]
        mods.append(nn.Linear(widths[-1], n_output_units))
        self.NN = nn.Sequential(*mods)

    def forward(self, t):
        # CUDA fp32 inputs go through the gfx950 stream kernels (autograd_ops.MlpJet: the output comes back together
        # with its input derivatives, so diff() / operators / loss.backward() of ANY caller run on the HIP path);
        # everything else -- CPU, fp64, shapes the kernels do not cover -- is the reference's plain Sequential
        if not isinstance(t, torch.Tensor):
            sym = _traced_call(self, t)          # a custom `enforce` / `compute_func_val` calling the network while the solver traces
            if sym is not None:
                return sym
        from .autograd_ops import try_jet_forward
        out = try_jet_forward(self, t)
        return self.NN(t) if out is None else out


class Resnet(nn.Module):
    """``FCNN(x) + W x``: a fully connected residual branch plus a trainable bias-free linear skip from input to output
    (reference: networks.py:73-106; attribute names ``residual`` / ``skip_connection`` as there, the residual branch is
    built first).  On the HIP path the skip is part of the output layer (``ndq_mlp_desc.skip``)."""

    def __init__(self, n_input_units=1, n_output_units=1, n_hidden_units=None, n_hidden_layers=None, actv=nn.Tanh,
                 hidden_units=(32, 32)):
        super().__init__()
        self.residual = FCNN(n_input_units=n_input_units, n_output_units=n_output_units, n_hidden_units=n_hidden_units,
                             n_hidden_layers=n_hidden_layers, actv=actv, hidden_units=hidden_units)
        self.skip_connection = nn.Linear(n_input_units, n_output_units, bias=False)

    def forward(self, t):
        if not isinstance(t, torch.Tensor):
            sym = _traced_call(self, t)
            if sym is not None:
                return sym
        return self.skip_connection(t) + self.residual(t)


class MonomialNN(nn.Module):
    """Parameter-free feature map ``x -> [x^d for d in degrees]`` along dim 1 (reference: networks.py:109-139);
    an int ``degrees = k`` means 1..k."""

    def __init__(self, degrees):
        super().__init__()
        if isinstance(degrees, int):
            degrees = list(range(1, degrees + 1))
        self.degrees = tuple(degrees)
        if len(self.degrees) == 0:
            raise ValueError("No degrees used, check `degrees` argument again")
        if 0 in degrees:
            warnings.warn("One of the degrees is 0 which might introduce redundant features")
        if len(set(self.degrees)) < len(self.degrees):
            warnings.warn(f"Duplicate degrees found: {self.degrees}")

    def forward(self, x):
        return torch.cat([x ** d for d in self.degrees], dim=1)

    def __repr__(self):
        return f"{self.__class__.__name__}(degrees={self.degrees})"

    __str__ = __repr__


# ------------------------------------------------------------------------------------------- kernel-side description
_ACT_IDS = {nn.Tanh: _lib.NDQ_ACT_TANH, SinActv: _lib.NDQ_ACT_SIN, nn.Sigmoid: _lib.NDQ_ACT_SIGMOID,
            Swish: _lib.NDQ_ACT_SWISH, APTx: _lib.NDQ_ACT_APTX,
            # torch's own modules with their DEFAULT arguments only (checked in describe): generic sigma ... sigma'''' tables
            nn.ELU: _lib.NDQ_ACT_ELU, nn.Softplus: _lib.NDQ_ACT_SOFTPLUS, nn.GELU: _lib.NDQ_ACT_GELU}


def _default_torch_activation(a):
    """nn.ELU / nn.Softplus / nn.GELU are compiled with torch's default arguments; anything else stays on the composite path."""
    if isinstance(a, nn.ELU):
        return a.alpha == 1.0
    if isinstance(a, nn.Softplus):
        return a.beta == 1.0 and a.threshold == 20.0
    if isinstance(a, nn.GELU):
        return getattr(a, "approximate", "none") == "none"
    return True


# hidden layers the kernel templates take: any number the LDS holds when all have one width (up to MAX_LAYERS are offered),
# up to four when the widths differ (ndq_mlp_desc.widths packs 8 bits per layer)
MAX_LAYERS = 8
MAX_HIDDEN = 512       # widest hidden layer (include/ndq.h NDQ_MAX_HIDDEN; wider than 64: csrc/ndq_wide.h)


def describe(net, dtype=torch.float32):
    """Return ``dict(d, hidden, layers, act, n_out, linears)`` if ``net`` is an FCNN the HIP kernels can run,
    else ``None`` (the solver then uses the composite autograd path for the whole system)."""
    skip = None
    outer = net
    if isinstance(net, Resnet):                  # FCNN branch + bias-free linear skip: out += S x, handled in-kernel
        if not _forward_is(net, Resnet) or _hooked(net):
            return None
        skip, net = net.skip_connection, net.residual
        if type(skip) is not nn.Linear or skip.bias is not None or skip.weight.dtype != dtype or _hooked(skip):
            return None
    seq = getattr(net, "NN", net)
    if not isinstance(seq, nn.Sequential) or not _forward_is(seq, nn.Sequential):
        return None
    # the module the solver calls must compute exactly "its Sequential": an FCNN (not a subclass with a forward of its own, not
    # some other module that happens to keep a Sequential under .NN), or the Sequential itself; and nobody listens in -- a
    # forward (pre-)hook may replace inputs / outputs, a backward hook gradients, and the kernels would never call it
    if seq is not net and not _forward_is(net, FCNN):
        return None
    if _hooked(net, seq):
        return None
    # a MonomialNN feature map in front: Sequential(MonomialNN(degrees), FCNN(...)) or Sequential(MonomialNN, Linear, actv,
    # ..., Linear) -- ascending degrees 1..8 (ndq_mlp_desc.mono: the first layer evaluates the powers and their derivatives)
    mono = 0
    if len(seq) >= 2 and isinstance(seq[0], MonomialNN):
        degs = list(seq[0].degrees)
        if skip is not None or degs != sorted(set(degs)) or degs[0] < 1 or degs[-1] > 8:
            return None
        mono = sum(1 << (k - 1) for k in degs)
        if not _forward_is(seq[0], MonomialNN) or _hooked(seq[0]):
            return None
        rest = list(seq)[1:]
        if len(rest) == 1 and isinstance(rest[0], FCNN):
            if not _forward_is(rest[0], FCNN) or not _forward_is(rest[0].NN, nn.Sequential) or _hooked(rest[0], rest[0].NN):
                return None
            seq = rest[0].NN
        else:
            seq = nn.Sequential(*rest)
    mods = list(seq)
    if len(mods) < 3 or len(mods) % 2 == 0:
        return None
    linears, acts = mods[0::2], mods[1::2]
    if not all(isinstance(m, nn.Linear) and m.bias is not None for m in linears):
        return None
    # plain linear layers with their own trainable nn.Parameters: no weight_norm / spectral_norm / parametrization (the weight is
    # then COMPUTED from other parameters), no subclass forward, no hooks on layers or activations
    if not all(_forward_is(m, nn.Linear) and isinstance(m._parameters.get("weight"), nn.Parameter)
               and isinstance(m._parameters.get("bias"), nn.Parameter) for m in linears) or _hooked(*mods):
        return None
    act_types = {type(a) for a in acts}
    if len(act_types) != 1 or next(iter(act_types)) not in _ACT_IDS or not all(_default_torch_activation(a) for a in acts):
        return None
    # Swish / APTx: their default fixed parameters (constants in the kernels), trainable ones on every layer
    # (ndq_mlp_desc.actp = 1: per-layer scalars at the end of the flat parameter vector) or fixed non-default values
    # (actp = 2: the scalars follow the trainable entries in the flat buffer, no gradient slots)
    act_params, act_frozen = [], []
    if acts and isinstance(acts[0], (Swish, APTx)):
        trainable = {bool(a.trainable) for a in acts}
        if len(trainable) != 1:
            return None
        names = ("beta",) if isinstance(acts[0], Swish) else ("alpha", "beta", "gamma")
        if trainable == {True}:
            act_params = [getattr(a, k) for a in acts for k in names]
            if any(not isinstance(p, nn.Parameter) or p.dim() != 0 or p.dtype != dtype for p in act_params):
                return None
            if len({id(p) for p in act_params}) != len(act_params):
                return None      # one module object used for several layers: its parameters are shared, not per layer
        elif any(tuple(getattr(a, k) for k in names) != ((1.0,) if len(names) == 1 else (1.0, 1.0, 0.5)) for a in acts):
            act_frozen = [float(getattr(a, k)) for a in acts for k in names]
            if any(getattr(a, "beta") == 0 or getattr(a, "gamma", 1.0) == 0 for a in acts):
                return None
    # hidden widths: any (the kernels lay all layers out for the widest one, padded to a multiple of 16)
    ws = [l.out_features for l in linears[:-1]]
    if any(b.in_features != a for a, b in zip(ws, linears[1:])):
        return None
    hidden = max(ws)
    if len(set(ws)) == 1:
        if len(ws) > MAX_LAYERS or hidden > MAX_HIDDEN:
            return None
        widths = 0
    elif hidden <= 64:
        if len(ws) > 4:
            return None
        widths = sum(w << (8 * i) for i, w in enumerate(ws))          # csrc/ndq_mlp.h: 8 bits per layer, up to four
    else:
        if len(ws) > 3 or hidden > MAX_HIDDEN:
            return None
        widths = sum(w << (10 * i) for i, w in enumerate(ws))         # csrc/ndq_deep.h: 10 bits per layer, two or three
    if any(p.dtype != dtype for l in linears for p in l.parameters()):
        return None
    if skip is not None and tuple(skip.weight.shape) != (linears[-1].out_features, linears[0].in_features):
        return None
    # flat order the kernels read: W1 b1 ... Wout bout | skip weights (n_out x d, row-major) | activation parameters
    # wider than 64 units the kernels serve plain FCNNs (csrc/ndq_wide.h, csrc/ndq_deep.h): a Resnet's skip connection is then
    # handled by the TRACER -- out += S x as a symbolic term whose entries are trainable kernel arguments
    # (symbolic.Graph.register_nets) -- and stays outside the flat parameter vector of the kernels
    skip_sym = None
    if skip is not None and hidden > 64:
        if act_params or act_frozen or mono:
            return None
        skip_sym, skip = skip, None
    params = [p for l in linears for p in (l.weight, l.bias)] + ([skip.weight] if skip is not None else []) + act_params
    every = params + ([skip_sym.weight] if skip_sym is not None else [])
    if len({id(p) for p in every}) != len(every):
        return None              # one parameter in two places (tied weights): one gradient in torch, two slots in the flat vector
    if not all(p.requires_grad for p in every):
        return None              # a frozen layer: torch leaves it alone (no gradient, the optimiser skips it); the fused step would not
    n_in = linears[0].in_features
    if mono:
        n_deg = bin(mono).count("1")
        if n_in % n_deg or hidden > 48:
            return None
        n_in //= n_deg
    # plain (non-parameter) numbers of the activation modules the kernels were specialised for: `act.beta = 2.0` by a callback
    # registers nothing with torch -- FlatParams.act_unchanged() re-reads them every epoch (nothing to read for Tanh / Sigmoid / sin)
    act_state = [(a, k, getattr(a, k)) for a in acts for k in ("alpha", "beta", "gamma", "threshold", "approximate", "trainable")
                 if hasattr(a, k) and not isinstance(getattr(a, k), torch.Tensor)]
    return dict(d=n_in, mono=mono, hidden=hidden, layers=len(acts), act=_ACT_IDS[next(iter(act_types))],
                n_out=linears[-1].out_features, linears=linears, skip=int(skip is not None), params=params,
                actp=1 if act_params else (2 if act_frozen else 0), frozen=act_frozen, widths=widths, skip_sym=skip_sym,
                act_state=act_state)


class FlatParams:
    """Parameters of one network as views of a single flat fp32 device buffer (torch order: W1, b1, W2, b2, ...).

    ``sync()`` is called before every launch: if something re-homed a parameter (``net.to(...)``, ``load_state_dict``
    on a copy, ...) the flat buffer is rebuilt from the current values, so the kernels always see what torch sees."""

    def __init__(self, net, device, dtype=torch.float32):
        self.net = net
        self.device = torch.device(device)
        self.dtype = dtype
        self.esize = 8 if dtype == torch.float64 else 4
        info = describe(net, dtype=dtype)
        self.params = info["params"]
        self.frozen = info["frozen"]      # fixed non-default activation scalars: behind the trainable entries of ``flat``
        self.act_state = info["act_state"]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        # gradient buffer with one spare trailing slot: single-network systems keep the batch loss there so that
        # [gradient | loss] is ONE contiguous all-reduce message (parallel.py)
        self.grad_loss = torch.zeros(self.numel + 1, dtype=dtype, device=self.device)
        self.grad = self.grad_loss[:self.numel]
        self._offsets = []
        self._grad_views = None
        off = 0
        for p in self.params:
            self._offsets.append(off)
            off += p.numel()
        self.sync()

    def _is_flat(self):
        """Do all parameters still alias the flat buffer?  (cheap: one data_ptr() per parameter, called every epoch)"""
        if self.flat is None:
            return False
        for p, ptr in zip(self.params, self._ptrs):
            if p.data_ptr() != ptr:
                return False
        return True

    def sync(self):
        if self._is_flat():
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().to(self.device, self.dtype).reshape(-1) for p in self.params]
                             + ([torch.tensor(self.frozen, dtype=self.dtype, device=self.device)] if self.frozen else [])
                             ).contiguous()
            for p, off in zip(self.params, self._offsets):
                p.data = flat[off:off + p.numel()].view(p.shape)
        self.flat = flat
        self._ptrs = [flat.data_ptr() + self.esize * off for off in self._offsets]

    def act_unchanged(self):
        """Are the plain numbers of the activation modules (Swish.beta, ELU.alpha, ...) still what the kernels were built for?"""
        for a, k, v in self.act_state:
            if getattr(a, k, None) != v:
                return False
        return True

    def all_trainable(self):
        """Does every parameter still ask for a gradient?  (``layer.requires_grad_(False)`` by a callback: the reference stops
        updating that layer -- cheap enough to ask every epoch)"""
        if torch._C._len_torch_function_stack():
            # a global TorchFunctionMode (torch.set_default_device('cuda'), the reference's import default) intercepts every tensor
            # attribute read -- ~0.6 us each: 5 us per epoch for this loop (engine.library_code)
            with torch._C.DisableTorchFunction():
                return all([p.requires_grad for p in self.params])
        for p in self.params:
            if not p.requires_grad:
                return False
        return True

    def attach_grads(self):
        """Expose the flat gradient buffer as ``p.grad`` views (what ``loss.backward()`` leaves behind, solvers.py:393)."""
        if self._grad_views is None:
            self._grad_views = [self.grad[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self._offsets)]
        for p, g in zip(self.params, self._grad_views):
            if p.grad is not g:
                p.grad = g

    def grads_attached(self):
        if self._grad_views is None:
            return False
        for p, g in zip(self.params, self._grad_views):
            if p.grad is not g:
                return False
        return True
