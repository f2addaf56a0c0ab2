"""PyTorch-ROCm custom ops + ONE ``autograd.Function`` over the gfx950 stream kernels, so that the reference's seams
work on the HIP path from ANY caller -- not only from inside a Solver:

    u = cond.enforce(net, x, y)              # conditions.py:41-57  ->  net(cat(x, y))  ->  ndq::mlp_jet_fwd
    r = diff(u, x, order=2) + diff(u, y, order=2)     # neurodiffeq.py:21-34: autograd sweeps, served by the streams
    (r ** 2).mean().backward()               # solvers.py:393       ->  ndq::mlp_jet_bwd
    torch.optim.Adam(net.parameters()).step()

How: ``FCNN.forward`` hands a CUDA fp32 ``(N, d)`` input to :class:`MlpJet`.  Its forward is ONE launch of
``ndq_mlp_jet_fwd`` that returns the network output TOGETHER with its first and second partial derivatives w.r.t. the
inputs (the "streams" of DESIGN.md section 2) as extra outputs of the same autograd node.  The node's backward

* while a higher-order graph is being built (``diff`` calls ``autograd.grad(..., create_graph=True)``): returns the
  gradient w.r.t. the input as a differentiable expression of the node's OWN outputs -- d N/dx_a is the stream N_a,
  d N_a/dx_b is the stream N_ab -- so the reference's k reverse sweeps cost k pointwise products instead of k walks
  through the network;
* in the final ``loss.backward()``: feeds the adjoints of ALL streams to one launch of ``ndq_mlp_jet_bwd`` (+ the
  fixed-order second-stage sum) and returns the parameter gradients, which land in ``p.grad`` of the user's
  ``nn.Parameter`` objects -- any ``torch.optim`` optimiser works on them.

The two launchers are registered with the dispatcher as ``torch.ops.ndq.mlp_jet_fwd`` / ``torch.ops.ndq.mlp_jet_bwd``
(``torch.library.custom_op`` over the C-ABI entry points of include/ndq.h).  Limits: derivatives of the network output
up to second order (third order raises); inputs must be CUDA fp32 and the network one :func:`networks.describe`
recognises -- anything else silently runs the module's ordinary torch forward, exactly like the reference.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from .networks import describe

_ENABLED = os.environ.get("NDQ_NATIVE_AUTOGRAD", "1") != "0"
_MAX_ORDER = 2
_COORD_GRADS = True             # see set_native_autograd(coordinate_grads=...)
_DEVICE_TYPES = ("cuda",)       # tests add "cpu" after registering oracle-backed CPU kernels for the two ops


class JetOrderError(RuntimeError):
    """A derivative of the network output beyond what the HIP forward launch provided was requested."""


def set_native_autograd(enabled=True, max_order=2, coordinate_grads=True):
    """Switch the HIP path of plain ``net(x)`` calls on / off; ``max_order`` in {0, 1, 2, 3, 4}: highest derivative of the
    network output w.r.t. its inputs the forward launch provides (lower = fewer streams = less work per call; 3 --
    tanh / sin / sigmoid networks -- carries every third-order partial: 10 streams for two inputs, 20 for three; 4 -- the
    same activations, networks of one or two inputs -- every fourth-order partial as well: 5 / 15 streams).

    ``coordinate_grads``: ``loss.backward()`` of the reference also leaves d loss / d x in the ``.grad`` of the sampled
    coordinate tensors (nobody reads it in a training step; residual-adaptive samplers might).  For a residual with
    k-th order derivatives that gradient needs the (k + 1)-th order streams of the network: one more forward launch
    with the larger stream set (or, where no such kernel exists, a recomputation on plain torch autograd) per backward
    pass.  True (default): always produce it, like the reference.  False: skip it when the network input is made of
    plain coordinate leaves only -- their ``.grad`` then stays None; gradients that flow on into trainable tensors
    UPSTREAM of the network input (a learnable input scaling, an embedding, another network) are always exact."""
    global _ENABLED, _MAX_ORDER, _COORD_GRADS
    if max_order not in (0, 1, 2, 3, 4):
        raise ValueError("max_order must be 0, 1, 2, 3 or 4")
    _ENABLED, _MAX_ORDER, _COORD_GRADS = bool(enabled), int(max_order), bool(coordinate_grads)


class native_autograd:
    """Context manager: ``with native_autograd(False): ...`` runs plain torch forwards inside."""

    def __init__(self, enabled=True, max_order=None, coordinate_grads=None):
        """``None`` keeps a setting as it is (``enabled=None``: only the other two change)."""
        self.want = (_ENABLED if enabled is None else bool(enabled), _MAX_ORDER if max_order is None else int(max_order),
                     _COORD_GRADS if coordinate_grads is None else bool(coordinate_grads))

    def __enter__(self):
        global _ENABLED, _MAX_ORDER, _COORD_GRADS
        self.keep = (_ENABLED, _MAX_ORDER, _COORD_GRADS)
        _ENABLED, _MAX_ORDER, _COORD_GRADS = self.want

    def __exit__(self, *exc):
        global _ENABLED, _MAX_ORDER, _COORD_GRADS
        _ENABLED, _MAX_ORDER, _COORD_GRADS = self.keep


def _pairs(d):
    return [(a, b) for a in range(d) for b in range(a, d)]


def _streams(d, order):
    """Multi-indices in kernel stream order: () | (a,) | (a, b) with a <= b (csrc/ndq_mlp.h: Streams<>)."""
    s = [()]
    if order >= 1:
        s += [(a,) for a in range(d)]
    if order >= 2:
        s += _pairs(d)
    if order >= 3:
        s += [(a, b, c) for a in range(d) for b in range(a, d) for c in range(b, d)]
    if order >= 4:
        s += [(a, b, c, e) for a in range(d) for b in range(a, d) for c in range(b, d) for e in range(c, d)]
    return s


def _round_up(n, m):
    return (n + m - 1) // m * m


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _entry(dtype):
    """(library, prefix) serving ``dtype``: libndq.so / ``ndq_`` for fp32, libndq64.so / ``ndq64_`` for fp64."""
    if dtype == torch.float64:
        return _lib.lib64(), "ndq64_"
    return _lib.lib(), "ndq_"


def _desc(d, order, hidden, layers, act, n_out, skip=0):
    mask2 = (1 << (d * (d + 1) // 2)) - 1 if order >= 2 else 0
    mask3 = (1 << (d * (d + 1) * (d + 2) // 6)) - 1 if order >= 3 else 0
    mask4 = (1 << (d * (d + 1) * (d + 2) * (d + 3) // 24)) - 1 if order >= 4 else 0
    return _lib.MlpDesc(d, 1 if order >= 1 else 0, mask2, hidden, layers, act, n_out, 0, skip, mask3, 0, 0, 0, mask4)


# ------------------------------------------------------------------------------------------------ dispatcher ops
@torch.library.custom_op("ndq::mlp_jet_fwd", mutates_args=(), device_types="cuda")
def mlp_jet_fwd(coords: torch.Tensor, params: torch.Tensor, n: int, order: int, hidden: int, layers: int, act: int,
                n_out: int) -> torch.Tensor:
    """coords [d][ld] fp32 SoA, params flat [P] (torch parameter order) -> streams [n_streams * n_out][ld]
    (C-ABI: ndq_mlp_jet_fwd, include/ndq.h)."""
    d, ld = coords.shape
    desc = _desc(d, order, hidden, layers, act, n_out)
    L, pre = _entry(coords.dtype)
    ns = getattr(L, pre + "mlp_num_streams")(ctypes.byref(desc))
    if ns <= 0:
        raise _lib.NdqError(f"no gfx950 kernel for {desc.key()} ({coords.dtype})")
    jets = torch.empty(ns * n_out, ld, dtype=coords.dtype, device=coords.device)
    _lib.check(getattr(L, pre + "mlp_jet_fwd")(ctypes.byref(desc), coords.data_ptr(), ld, n, params.data_ptr(),
                                               jets.data_ptr(), ld, _stream_ptr(coords.device)), pre + "mlp_jet_fwd")
    return jets


@mlp_jet_fwd.register_fake
def _(coords, params, n, order, hidden, layers, act, n_out):
    d, ld = coords.shape
    return coords.new_empty((len(_streams(d, order)) * n_out, ld))


@torch.library.custom_op("ndq::mlp_jet_bwd", mutates_args=(), device_types="cuda")
def mlp_jet_bwd(coords: torch.Tensor, params: torch.Tensor, gbar: torch.Tensor, n: int, order: int, hidden: int,
                layers: int, act: int, n_out: int) -> torch.Tensor:
    """Adjoint of mlp_jet_fwd w.r.t. the parameters: gbar [n_streams * n_out][ld] -> flat gradient [P]
    (C-ABI: ndq_mlp_jet_bwd + ndq_reduce_partials; fixed summation order, run-to-run bit-identical)."""
    d, ld = coords.shape
    desc = _desc(d, order, hidden, layers, act, n_out)
    L, pre = _entry(coords.dtype)
    blocks = getattr(L, pre + "mlp_bwd_blocks")(ctypes.byref(desc), n)
    P = getattr(L, pre + "mlp_num_params")(ctypes.byref(desc))
    if blocks <= 0 or P != params.numel():
        raise _lib.NdqError(f"no gfx950 kernel for {desc.key()} / parameter count mismatch ({P} vs {params.numel()})")
    stream = _stream_ptr(coords.device)
    partials = torch.empty(blocks, P, dtype=coords.dtype, device=coords.device)
    _lib.check(getattr(L, pre + "mlp_jet_bwd")(ctypes.byref(desc), coords.data_ptr(), ld, n, params.data_ptr(),
                                               gbar.data_ptr(), ld, partials.data_ptr(), stream), pre + "mlp_jet_bwd")
    grad = torch.empty(P, dtype=coords.dtype, device=coords.device)
    _lib.check(getattr(L, pre + "reduce_partials")(partials.data_ptr(), blocks, P, grad.data_ptr(), 0, 1.0, stream),
               pre + "reduce_partials")
    return grad


@mlp_jet_bwd.register_fake
def _(coords, params, gbar, n, order, hidden, layers, act, n_out):
    return params.new_empty(params.shape)


# ------------------------------------------------------------------------------------------------ autograd node
def _will_run(node):
    """Will the autograd engine execute ``node`` in the backward pass that is running?  (False for the inputs a
    ``torch.autograd.grad(..., inputs=[x])`` sweep does not ask for.)  Unknown -> assume yes."""
    if node is None:
        return False
    try:
        return bool(torch._C._will_engine_execute_node(node))
    except Exception:           # noqa: BLE001 -- not inside a backward pass / private API moved: be conservative
        return True


def _only_coordinate_leaves(node, limit=256):
    """True if everything upstream of ``node`` ends in leaf tensors that are NOT ``nn.Parameter``s (sampled coordinates
    and constants); False as soon as a parameter shows up or the graph is larger than ``limit`` nodes."""
    seen, todo = set(), [node]
    while todo:
        n = todo.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        if len(seen) > limit:
            return False
        var = getattr(n, "variable", None)              # AccumulateGrad
        if var is not None:
            if isinstance(var, torch.nn.Parameter):
                return False
            continue
        todo.extend(fn for fn, _ in n.next_functions)
    return True


class MlpJet(torch.autograd.Function):
    """inputs: X (N, d), the network's flat parameter vector (``torch.cat`` of its parameters: ONE tensor input whose
    producer node the engine can be asked about, see ``_will_run``); outputs: one (N, n_out) tensor per stream, in
    kernel order; output 0 is the network output itself.

    ``backward`` never drops a gradient silently: what the streams of the forward launch cannot express --
    the input gradient of a top-order stream, parameter gradients that have to be differentiable themselves -- is
    obtained from a forward launch with the next-higher stream set or recomputed on plain torch autograd."""

    @staticmethod
    def forward(ctx, X, flat_in, spec, net, params):
        d, order, hidden, layers, act, n_out = spec
        n = X.shape[0]
        ld = _round_up(n, 64)
        coords = torch.zeros(d, ld, dtype=X.dtype, device=X.device)
        coords[:, :n] = X.detach().t()
        flat = flat_in.detach()
        jets = torch.ops.ndq.mlp_jet_fwd(coords, flat, n, order, hidden, layers, act, n_out)
        outs = tuple(jets[s * n_out:(s + 1) * n_out, :n].t() for s in range(len(_streams(d, order))))
        ctx.spec, ctx.n, ctx.ld, ctx.net, ctx.params = spec, n, ld, net, params
        ctx.coord_grads = _COORD_GRADS
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(coords, flat, X, *outs)
        return outs

    @staticmethod
    def _plain_vjp(ctx, X, gouts, want_x, want_p, create_graph):
        """The vector-Jacobian product of this node recomputed on plain torch autograd (the reference's own cost model:
        one nested sweep per stream): exact for every stream order, differentiable when ``create_graph``.  Returns
        (gX or None, flat parameter gradient or None)."""
        d, order, hidden, layers, act, n_out = ctx.spec
        streams = _streams(d, order)
        params = list(ctx.params)
        with native_autograd(False), torch.enable_grad():
            Xr = X if (create_graph and X.requires_grad) else X.detach().requires_grad_(True)
            y = ctx.net(Xr)
            vals = {(): [y[:, j:j + 1] for j in range(n_out)]}
            for mi in streams[1:]:
                parent = vals[mi[:-1]]
                vals[mi] = [torch.autograd.grad(c, Xr, torch.ones_like(c), create_graph=True)[0][:, mi[-1]:mi[-1] + 1]
                            for c in parent]
            outs, gos = [], []
            for s, g in enumerate(gouts):
                if g is not None:
                    outs.append(torch.cat(vals[streams[s]], dim=1))
                    gos.append(g)
            wrt = ([Xr] if want_x else []) + (params if want_p else [])
            got = list(torch.autograd.grad(outs, wrt, gos, create_graph=create_graph, allow_unused=True)) if wrt else []
        gX = got.pop(0) if want_x else None
        gflat = None
        if want_p:
            gflat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(got, params)])
        return gX, gflat

    @staticmethod
    def backward(ctx, *gouts):
        d, order, hidden, layers, act, n_out = ctx.spec
        coords, flat, X, *outs = ctx.saved_tensors
        streams = _streams(d, order)
        index = {mi: k for k, mi in enumerate(streams)}
        n, ld = ctx.n, ctx.ld
        # which gradients does THIS backward pass need?  (a diff() sweep asks for the input only, autograd.grad(loss,
        # params) for the parameters only, loss.backward() for everything that requires grad)
        nf = ctx.next_functions                    # one entry per TENSOR input: X, the flat parameter vector
        want_x = bool(ctx.needs_input_grad[0]) and _will_run(nf[0][0])
        want_p = bool(ctx.needs_input_grad[1]) and _will_run(nf[1][0])
        top = any(g is not None and len(streams[s]) == order for s, g in enumerate(gouts))    # adjoint on a top-order stream

        def input_grad(values, idx):
            """sum_s g_s * d(stream s)/dx_a for every input a, from stream values ``values`` indexed by ``idx``"""
            cols = []
            for a in range(d):
                tot = None
                for s, g in enumerate(gouts):
                    if g is None:
                        continue
                    mi = tuple(sorted(streams[s] + (a,)))
                    if mi not in idx:
                        raise JetOrderError(
                            f"derivative of order {len(mi)} of a network output w.r.t. its inputs requested, but the HIP "
                            f"forward provides orders <= {order}: call neurodiffeq_amd.set_native_autograd(max_order=3) "
                            "for third order (4: fourth, networks of one or two inputs), or set_native_autograd(False) for "
                            "the plain torch forward")
                    t = g * values[idx[mi]]
                    if n_out > 1:
                        t = t.sum(dim=1, keepdim=True)
                    tot = t if tot is None else tot + t
                cols.append(tot if tot is not None else torch.zeros(n, 1, dtype=coords.dtype, device=coords.device))
            return torch.cat(cols, dim=1)

        if torch.is_grad_enabled():
            # a higher-order graph is being built.  diff() sweeps (create_graph=True, inputs = coordinates) need the
            # input gradient as a differentiable expression of this node's own outputs; parameter gradients that must be
            # differentiable (loss.backward(create_graph=True)) cannot come from the adjoint kernel: plain torch
            if want_p:
                gX, gflat = MlpJet._plain_vjp(ctx, X, gouts, want_x, True, True)
                return gX, gflat, None, None, None
            return (input_grad(outs, index) if want_x else None), None, None, None, None
        # the final backward: parameter gradients from ONE adjoint launch over all streams
        gflat = None
        if want_p:
            gbar = torch.zeros(len(streams) * n_out, ld, dtype=coords.dtype, device=coords.device)
            for s, g in enumerate(gouts):
                if g is not None:
                    gbar[s * n_out:(s + 1) * n_out, :n] = g.t()
            gflat = torch.ops.ndq.mlp_jet_bwd(coords, flat, gbar, n, order, hidden, layers, act, n_out)
        # d loss / d inputs: from the node's own streams unless a top-order stream carries an adjoint -- then it needs
        # the streams one order up (the reference computes this gradient in every step and nobody reads it, SURVEY.md
        # App. A.2; it matters when trainable tensors sit upstream of the network input)
        gX = None
        if want_x and not top:
            gX = input_grad(outs, index)
        elif want_x and (ctx.coord_grads or not _only_coordinate_leaves(nf[0][0])):
            up = _spec_for(ctx.net, order + 1, coords.dtype) if order + 1 <= 4 else None
            if up is not None:
                jets = torch.ops.ndq.mlp_jet_fwd(coords, flat, n, order + 1, hidden, layers, act, n_out)
                hi = _streams(d, order + 1)
                vals = [jets[s * n_out:(s + 1) * n_out, :n].t() for s in range(len(hi))]
                gX = input_grad(vals, {mi: k for k, mi in enumerate(hi)})
            else:
                gX, _ = MlpJet._plain_vjp(ctx, X, gouts, True, False, False)
        return gX, gflat, None, None, None


# ------------------------------------------------------------------------------------------------ the seam
_SPECS = weakref.WeakKeyDictionary()


def _spec_for(net, order, dtype=torch.float32):
    """(d, order, hidden, layers, act, n_out) + parameter list if the gfx950 kernels can run ``net`` in ``dtype``, else
    None."""
    cache = _SPECS.setdefault(net, {})
    from .networks import STRUCTURE, track_structure
    if cache.get("structure") != STRUCTURE[0]:           # a layer / parameter / hook of some tracked network changed: ask again
        cache.clear()
        cache["structure"] = STRUCTURE[0]
    state = cache.get("act_state")
    if state:                                            # plain numbers of the activation modules (ELU.alpha, Swish.beta ...)
        for a, k, v in state:
            if getattr(a, k, None) != v:
                cache.clear()
                cache["structure"] = STRUCTURE[0]
                break
    hit = cache.get((order, dtype))
    if hit is not None:
        return hit if hit else None
    track_structure(net)
    cache["structure"] = STRUCTURE[0]
    info = describe(net, dtype=dtype)
    if info is not None:
        cache["act_state"] = info["act_state"]
    ok = info is not None and info["skip"] == 0 and info.get("skip_sym") is None and info["actp"] == 0 and info["widths"] == 0 and info["mono"] == 0 and 1 <= info["d"] <= 3 \
        and (order < 4 or info["d"] <= 2)          # (every fourth-order partial of three inputs would be 35 streams)
    if ok:
        from . import codegen
        desc = _desc(info["d"], order, info["hidden"], info["layers"], info["act"], info["n_out"])
        try:
            ok = bool(codegen.ensure_mlp_kernels(desc, f64=(dtype == torch.float64)))
        except _lib.NdqError:
            ok = False
    cache[(order, dtype)] = ((info["d"], order, info["hidden"], info["layers"], info["act"], info["n_out"]),
                             info["params"]) if ok else False
    return cache[(order, dtype)] or None


def try_jet_forward(net, t):
    """``net(t)`` through the HIP stream kernels if possible, else None (the caller runs its torch forward)."""
    if not _ENABLED or not isinstance(t, torch.Tensor) or t.device.type not in _DEVICE_TYPES \
            or t.dtype not in (torch.float32, torch.float64) or t.dim() != 2:
        return None
    if t.shape[0] == 0 or torch.jit.is_tracing() or torch.is_autocast_enabled():
        return None                      # (under autocast the plain forward runs its layers in half precision, as the user asked)
    want_grad = torch.is_grad_enabled() and t.requires_grad
    order = _MAX_ORDER if want_grad else 0
    spec = _spec_for(net, order, t.dtype)
    if spec is None or t.shape[1] != spec[0][0]:
        return None
    params = spec[1]
    if any(p.device != t.device or p.dtype != t.dtype for p in params):
        return None
    # the parameters enter as ONE flat vector (torch.cat hands its gradient back to every p.grad)
    flat = torch.cat([p.reshape(-1) for p in params])
    return MlpJet.apply(t, flat, spec[0], net, params)[0]
