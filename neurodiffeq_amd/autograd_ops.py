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
_DEVICE_TYPES = ("cuda",)       # tests add "cpu" after registering oracle-backed CPU kernels for the two ops


class JetOrderError(RuntimeError):
    """A derivative of the network output beyond what the HIP forward launch provided was requested."""


def set_native_autograd(enabled=True, max_order=2):
    """Switch the HIP path of plain ``net(x)`` calls on / off; ``max_order`` in {0, 1, 2, 3}: highest derivative of the
    network output w.r.t. its inputs the forward launch provides (lower = fewer streams = less work per call; 3 --
    tanh / sin / sigmoid networks -- carries every third-order partial: 10 streams for two inputs, 20 for three)."""
    global _ENABLED, _MAX_ORDER
    if max_order not in (0, 1, 2, 3):
        raise ValueError("max_order must be 0, 1, 2 or 3")
    _ENABLED, _MAX_ORDER = bool(enabled), int(max_order)


class native_autograd:
    """Context manager: ``with native_autograd(False): ...`` runs plain torch forwards inside."""

    def __init__(self, enabled=True, max_order=None):
        self.want = (bool(enabled), _MAX_ORDER if max_order is None else int(max_order))

    def __enter__(self):
        global _ENABLED, _MAX_ORDER
        self.keep = (_ENABLED, _MAX_ORDER)
        _ENABLED, _MAX_ORDER = self.want

    def __exit__(self, *exc):
        global _ENABLED, _MAX_ORDER
        _ENABLED, _MAX_ORDER = self.keep


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
    return _lib.MlpDesc(d, 1 if order >= 1 else 0, mask2, hidden, layers, act, n_out, 0, skip, mask3)


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
class MlpJet(torch.autograd.Function):
    """outputs: one (N, n_out) tensor per stream, in kernel order; output 0 is the network output itself."""

    @staticmethod
    def forward(ctx, X, spec, *params):
        d, order, hidden, layers, act, n_out = spec
        n = X.shape[0]
        ld = _round_up(n, 64)
        coords = torch.zeros(d, ld, dtype=X.dtype, device=X.device)
        coords[:, :n] = X.detach().t()
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        jets = torch.ops.ndq.mlp_jet_fwd(coords, flat, n, order, hidden, layers, act, n_out)
        outs = tuple(jets[s * n_out:(s + 1) * n_out, :n].t() for s in range(len(_streams(d, order))))
        ctx.spec, ctx.n, ctx.ld = spec, n, ld
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(coords, flat, *outs)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        d, order, hidden, layers, act, n_out = ctx.spec
        coords, flat, *outs = ctx.saved_tensors
        streams = _streams(d, order)
        index = {mi: k for k, mi in enumerate(streams)}
        n, ld = ctx.n, ctx.ld
        second = any(g is not None and len(streams[s]) == order and order >= 2 for s, g in enumerate(gouts))

        def input_grad():
            """sum_s g_s * d(stream s)/dx_a for every input a, from the node's own outputs"""
            cols = []
            for a in range(d):
                tot = None
                for s, g in enumerate(gouts):
                    if g is None:
                        continue
                    mi = tuple(sorted(streams[s] + (a,)))
                    if mi not in index:
                        raise JetOrderError(
                            f"derivative of order {len(mi)} of a network output w.r.t. its inputs requested, but the HIP "
                            f"forward provides orders <= {order}: call neurodiffeq_amd.set_native_autograd(max_order=3) "
                            "for third order, or set_native_autograd(False) for the plain torch forward")
                    t = g * outs[index[mi]]
                    if n_out > 1:
                        t = t.sum(dim=1, keepdim=True)
                    tot = t if tot is None else tot + t
                cols.append(tot if tot is not None else torch.zeros(n, 1, dtype=coords.dtype, device=coords.device))
            return torch.cat(cols, dim=1)

        if torch.is_grad_enabled():
            # a sweep of diff() (create_graph=True): only the input gradient matters, as a differentiable expression
            gX = input_grad() if ctx.needs_input_grad[0] else None
            return (gX, None) + (None,) * len(ctx.shapes)
        # the final backward: parameter gradients from ONE adjoint launch over all streams
        grads = [None] * len(ctx.shapes)
        if any(ctx.needs_input_grad[2:]):
            gbar = torch.zeros(len(streams) * n_out, ld, dtype=coords.dtype, device=coords.device)
            for s, g in enumerate(gouts):
                if g is not None:
                    gbar[s * n_out:(s + 1) * n_out, :n] = g.t()
            gflat = torch.ops.ndq.mlp_jet_bwd(coords, flat, gbar, n, order, hidden, layers, act, n_out)
            off = 0
            for k, shape in enumerate(ctx.shapes):
                cnt = 1
                for v in shape:
                    cnt *= v
                if ctx.needs_input_grad[2 + k]:
                    grads[k] = gflat[off:off + cnt].view(shape)
                off += cnt
        # d loss / d inputs of the network part: available unless it needs third-order streams (a training step never
        # reads it; the reference computes it and throws it away)
        gX = input_grad() if (ctx.needs_input_grad[0] and not second) else None
        return (gX, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ the seam
_SPECS = weakref.WeakKeyDictionary()


def _spec_for(net, order, dtype=torch.float32):
    """(d, order, hidden, layers, act, n_out) + parameter list if the gfx950 kernels can run ``net`` in ``dtype``, else
    None."""
    cache = _SPECS.setdefault(net, {})
    hit = cache.get((order, dtype))
    if hit is not None:
        return hit if hit else None
    info = describe(net, dtype=dtype)
    ok = info is not None and info["skip"] == 0 and info["actp"] == 0 and info["widths"] == 0 and info["mono"] == 0 and 1 <= info["d"] <= 3
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
    if t.shape[0] == 0 or torch.jit.is_tracing():
        return None
    want_grad = torch.is_grad_enabled() and t.requires_grad
    order = _MAX_ORDER if want_grad else 0
    spec = _spec_for(net, order, t.dtype)
    if spec is None or t.shape[1] != spec[0][0]:
        return None
    params = spec[1]
    if any(p.device != t.device or p.dtype != t.dtype for p in params):
        return None
    return MlpJet.apply(t, spec[0], *params)[0]
