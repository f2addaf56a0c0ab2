"""HIP code generation for the traced pointwise stage (see :mod:`neurodiffeq_amd.symbolic`).

One generated kernel per traced PDE system does, per collocation point, everything the reference does between the
network forward and the parameter backward: condition re-parameterisation (conditions.py ``parameterize``), the
user's residuals (``diff_eqs``, solvers.py:380, with every ``diff`` already resolved symbolically), the squared
residual sum for the loss (solvers.py:218) and the adjoint of that loss w.r.t. every network output stream -- the
seed of ``loss.backward()`` (solvers.py:393).  It is the HBM-bound kernel of the path: it reads the coordinates and
the streams, writes the adjoint streams (+ optionally function values / residuals) and one partial sum per block.

The launcher it exports has the ``ndq_pointwise_fn`` signature of include/ndq.h.
"""
import ctypes
import hashlib
import os
import re

from .symbolic import Graph, TraceUnsupported

HERE = os.path.dirname(os.path.abspath(__file__))
JIT_DIR = os.path.join(HERE, "_jit")
from . import _hipcc        # noqa: E402  (hipcc + the device-assembly fix-up pass; see _hipcc.py)


def _extra_flags():
    """Tuning knobs for experiments: NDQ_JIT_FLAGS="-DNDQ_X=1 ..." is appended to hipcc and hashed into the cache key."""
    return os.environ.get("NDQ_JIT_FLAGS", "").split()


# ----------------------------------------------------------------------------------------------- stream layout
def pair_list(d):
    return [(a, b) for a in range(d) for b in range(a, d)]


def triple_list(d):
    return [(a, b, c) for a in range(d) for b in range(a, d) for c in range(b, d)]


#: bf16 plane products of the single-launch closure kernels' GEMMs (csrc/ndq_mlp.h: NDQ_FWD_NPROD / NDQ_HBAR_NPROD / NDQ_WG_NPROD):
#: forward 6 (the derivative streams keep fp32 class: 5 puts u_xx of the reference's trained C3 state at 3.2e-5), reverse 4,
#: weight gradients 3 -- measured against the unmodified reference's fp64 numbers in profiles/r06_headline_ab.md.  The generic
#: adjoint kernels behind the C-ABI keep all six (arbitrary seeds).  NDQ_JIT_FLAGS may override (A/B runs).
CLOSURE_PRODUCTS = ("#ifndef NDQ_HBAR_NPROD\n#define NDQ_HBAR_NPROD 4\n#endif\n"
                    "#ifndef NDQ_WG_NPROD\n#define NDQ_WG_NPROD 3\n#endif\n")


def quad_list(d):
    return [(a, b, c, e) for a in range(d) for b in range(a, d) for c in range(b, d) for e in range(c, d)]


def _m4_arg(desc):
    """Trailing template argument of ndq::Cfg for fourth-order streams -- appended only when there are any, so that the
    generated source (and with it the build cache key) of every other module stays what it was."""
    return f", {desc.mask4}u" if getattr(desc, "mask4", 0) else ""


class NetStreams:
    """Which derivative streams of net ``k`` the residual needs, closed to a set the MLP kernels provide.

    deps: global coordinate indices fed to the net (in order).  Stream slots follow include/ndq.h."""

    def __init__(self, deps, n_out):
        self.deps = tuple(deps)
        self.d = len(deps)
        self.n_out = n_out
        self.first = 0
        self.mask2 = 0
        self.mask3 = 0          # third-order triples (include/ndq.h: ndq_mlp_desc.mask3)
        self.mask4 = 0          # fourth-order quadruples (ndq_mlp_desc.mask4; round 6)
        self.lap = 0            # 1: the diagonal pairs of mask2 travel as ONE stream holding their sum

    def need(self, mi):
        """mi: multi-index of GLOBAL coordinate indices, or ("L", a, b, ..) for the Laplacian stream."""
        if mi and mi[0] == "L":
            self.first, self.lap = 1, 1
            for c in mi[1:]:
                a = self.deps.index(c)
                self.mask2 |= 1 << pair_list(self.d).index((a, a))
            return
        loc = tuple(sorted(self.deps.index(c) for c in mi))
        if len(loc) >= 1:
            self.first = 1
        if len(loc) == 2:
            self.mask2 |= 1 << pair_list(self.d).index(loc)
        if len(loc) == 3:               # a triple travels with its three pairs (the recurrence needs them)
            self.mask3 |= 1 << triple_list(self.d).index(loc)
            for pair in ((loc[0], loc[1]), (loc[0], loc[2]), (loc[1], loc[2])):
                self.mask2 |= 1 << pair_list(self.d).index(pair)
        if len(loc) == 4:               # a quadruple travels with its four triples and six pairs (Faa di Bruno over the positions)
            if self.d > 3:
                raise TraceUnsupported("fourth-order derivatives of a network with more than three inputs are outside the fused path")
            self.mask4 |= 1 << quad_list(self.d).index(loc)
            for i in range(4):
                tri = tuple(loc[j] for j in range(4) if j != i)
                self.mask3 |= 1 << triple_list(self.d).index(tri)
                for j in range(i + 1, 4):
                    self.mask2 |= 1 << pair_list(self.d).index((loc[i], loc[j]))
        if len(loc) > 4:
            raise TraceUnsupported("derivatives of network outputs beyond fourth order are outside the fused path")

    @property
    def n_streams(self):
        return (1 + self.first * self.d + (1 if self.lap else bin(self.mask2).count("1")) + bin(self.mask3).count("1")
                + bin(self.mask4).count("1"))

    def slot(self, mi):
        if mi and mi[0] == "L":
            return 1 + self.d
        assert not (self.lap and len(mi) == 2)
        loc = tuple(sorted(self.deps.index(c) for c in mi))
        if len(loc) == 0:
            return 0
        if len(loc) == 1:
            return 1 + loc[0]
        if len(loc) == 4:
            k = quad_list(self.d).index(loc)
            assert (self.mask4 >> k) & 1 and not self.lap
            return 1 + self.d + bin(self.mask2).count("1") + bin(self.mask3).count("1") + bin(self.mask4 & ((1 << k) - 1)).count("1")
        if len(loc) == 3:
            k = triple_list(self.d).index(loc)
            assert (self.mask3 >> k) & 1 and not self.lap
            return 1 + self.d + bin(self.mask2).count("1") + bin(self.mask3 & ((1 << k) - 1)).count("1")
        k = pair_list(self.d).index(loc)
        assert (self.mask2 >> k) & 1
        return 1 + self.d + bin(self.mask2 & ((1 << k) - 1)).count("1")


# ----------------------------------------------------------------------------------------------- expression emission
def _lit(v):
    if v != v or v in (float("inf"), float("-inf")):
        raise ValueError("non-finite constant in traced expression")
    s = repr(float(v))          # shortest decimal that round-trips the double: exact for the fp32 and the fp64 build alike
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s + "f"


def _powi_expr(x, n):
    if n == 0:
        return "1.0f"
    if n == 1:
        return x
    if n <= 8:
        return "(" + "*".join([x] * n) + ")"
    return f"powf({x}, {float(n)}f)"


_UN_FWD = {
    "neg": "-{a}", "sin": "sinf({a})", "cos": "cosf({a})", "tan": "tanf({a})", "exp": "expf({a})",
    "log": "logf({a})", "tanh": "tanhf({a})", "sqrt": "sqrtf({a})", "abs": "fabsf({a})", "sinh": "sinhf({a})",
    "cosh": "coshf({a})", "sigmoid": "(1.0f/(1.0f+expf(-{a})))", "recip": "(1.0f/{a})",
    "sign": "(({a}>0.0f)-({a}<0.0f))",
    "log1p": "log1pf({a})", "expm1": "expm1f({a})", "erf": "erff({a})", "atan": "atanf({a})",
    "floor": "floorf({a})", "ceil": "ceilf({a})", "round": "rintf({a})", "trunc": "truncf({a})",       # (rint: half to even, like torch.round)
    "detach": "{a}",
}
# adjoint factor of the single child: child_adj += b * factor ; {a} child value, {v} node value
_UN_ADJ = {
    "neg": "-{b}", "sin": "{b}*cosf({a})", "cos": "-{b}*sinf({a})", "tan": "{b}*(1.0f+{v}*{v})",
    "exp": "{b}*{v}", "log": "{b}/{a}", "tanh": "{b}*(1.0f-{v}*{v})", "sqrt": "{b}/(2.0f*{v})",
    "abs": "{b}*(({a}>0.0f)-({a}<0.0f))", "sinh": "{b}*coshf({a})", "cosh": "{b}*sinhf({a})",
    "sigmoid": "{b}*{v}*(1.0f-{v})", "recip": "-{b}*{v}*{v}", "sign": None,
    "log1p": "{b}/(1.0f+{a})", "expm1": "{b}*({v}+1.0f)", "erf": "{b}*1.1283791670955126f*expf(-{a}*{a})",
    "atan": "{b}/(1.0f+{a}*{a})",
    "floor": None, "ceil": None, "round": None, "trunc": None, "detach": None,      # (no adjoint flows into the child)
}


def _tv_launcher(args_t, fill, kern_tv, lds_train, lds_eval, threads="CFG::BWD_THREADS", kern_loop=None, lds_loop=None):
    """Host launcher of the train + validation closure kernel (csrc/ndq_mlp.h: fused_*_closure_tv_kernel), emitted into
    the anonymous namespace of every generated closure module: workgroups [0, blocks(n)) run the training closure on the
    training batch, the next blocks(vn) the forward-only closure on the validation batch; n = 0 / vn = 0 drops a half."""
    return f"""
int launch_tv(const float* coords, int ldc, int n, const float* const* params, float* const* partials, float* loss_partials,
              float seed, const float* vcoords, int vldc, int vn, float* vloss_partials, const void* pull, void* stream) {{
  if (!params || n < 0 || vn < 0 || (n == 0 && vn == 0)) return -2;
  if (n > 0 && (!coords || !partials || !loss_partials || ldc < n)) return -2;
  if (vn > 0 && (!vcoords || !vloss_partials || vldc < vn)) return -2;
  {args_t} t{{}}, v{{}};
  {{
    {args_t}& a = t;
    a.coords = coords; a.loss_partials = loss_partials; a.n = n; a.ldc = ldc; a.ldj = ldc; a.seed = seed;
    a.theta = g_theta; a.theta_partials = g_theta_partials;
    {fill}
  }}
  {{
    {args_t}& a = v;
    float* const* partials = nullptr;
    a.coords = vcoords; a.loss_partials = vloss_partials; a.n = vn; a.ldc = vldc; a.ldj = vldc; a.seed = 0.f;
    a.theta = g_theta;
    {fill}
  }}
  // a training epoch on its own is the plain training kernel (the same device code as the training half of the combined
  // kernel -- engine.verify_fused compares the two bit for bit -- without the second body's registers: C2 -1 us, C3 -13 us)
  static const bool always_tv = getenv("NDQ_TV_ALWAYS") != nullptr;        // measurement knob
  if (vn == 0 && !pull && !always_tv)
    return launch(coords, ldc, n, params, partials, loss_partials, nullptr, nullptr, ldc, seed, 1, stream);
  const int tb = n > 0 ? fused_blocks(n) : 0, vb = vn > 0 ? fused_blocks(vn) : 0;
  static bool attr = false;
  if (!attr) {{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern_tv}),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds_train});
    if (e != hipSuccess) return (int)e;
    attr = true;
  }}
  ndq::PullArgs pa{{}};        // pull prologue (csrc/ndq_tail.h): the launch finishes the previous epoch itself
  if (pull) {{
    if (!ndq::pull_supported<CFG>()) return -2;
    pa = *static_cast<const ndq::PullArgs*>(pull);
  }}
  hipLaunchKernelGGL(({kern_tv}), dim3(tb + vb), dim3({threads}), tb > 0 ? {lds_train} : {lds_eval},
                     static_cast<hipStream_t>(stream), t, v, tb, pa);
  return (int)hipGetLastError();
}}""" + (f"""
// loop mode (csrc/ndq_tail.h: LoopArgs): ONE workgroup runs a run of fit()'s launches back to back, state in LDS
int launch_loop(const float* coords, int ldc, int n, float seed, const float* vcoords, int vldc, int vn, const void* loop,
                void* stream) {{
  if (!loop || !ndq::pull_supported<CFG>() || n < 0 || vn < 0 || (n > 0 && (!coords || ldc < n || fused_blocks(n) != 1)) ||
      (vn > 0 && (!vcoords || vldc < vn || fused_blocks(vn) != 1)))
    return -2;
  {args_t} t{{}}, v{{}};
  t.coords = coords; t.n = n; t.ldc = ldc; t.ldj = ldc; t.seed = seed; t.theta = g_theta;
  v.coords = vcoords; v.n = vn; v.ldc = vldc; v.ldj = vldc; v.seed = 0.f; v.theta = g_theta;
  static bool attr = false;
  if (!attr) {{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern_loop}),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds_loop});
    if (e != hipSuccess) return (int)e;
    attr = true;
  }}
  hipLaunchKernelGGL(({kern_loop}), dim3(1), dim3({threads}), {lds_loop}, static_cast<hipStream_t>(stream), t, v,
                     *static_cast<const ndq::LoopArgs*>(loop));
  return (int)hipGetLastError();
}}
int loop_ok() {{ return ndq::pull_supported<CFG>() && {lds_loop} <= 160 * 1024 ? 1 : 0; }}
""" if kern_loop else """
int launch_loop(const float*, int, int, float, const float*, int, int, const void*, void*) { return -2; }
int loop_ok() { return 0; }
""")


# the ndq_fused_launch_tv_fn of include/ndq.h
_TV_EXPORT = """
extern "C" int ndq_fused_launch_tv(const float* coords, int ldc, int n, const float* const* params, float* const* partials,
                                   float* loss_partials, float seed, const float* vcoords, int vldc, int vn,
                                   float* vloss_partials, const void* pull, void* stream) {
  return launch_tv(coords, ldc, n, params, partials, loss_partials, seed, vcoords, vldc, vn, vloss_partials, pull, stream);
}
extern "C" int ndq_fused_pull_ok() { return ndq::pull_supported<CFG>() ? 1 : 0; }
// the ndq_fused_launch_loop_fn of include/ndq.h
extern "C" int ndq_fused_launch_loop(const float* coords, int ldc, int n, float seed, const float* vcoords, int vldc, int vn,
                                     const void* loop, void* stream) {
  return launch_loop(coords, ldc, n, seed, vcoords, vldc, vn, loop, stream);
}
extern "C" int ndq_fused_loop_ok() { return loop_ok(); }
"""


class PointwiseProgram:
    """A traced system lowered to straight-line fp32 code.

    graph      : symbolic.Graph
    residuals  : node ids, one per equation (columns of the reference's (N, n_eq) residual, solvers.py:381)
    funcs      : node ids of the re-parameterised function values (one per condition, solvers.py:373-375)
    streams    : {net_idx: NetStreams}
    """

    LOSS_KINDS = ("l2", "l1", "infinity", "custom")

    def __init__(self, graph: Graph, residuals, funcs, n_nets, widen=None, allow_lap=None, unify=None, loss="l2",
                 loss_term=None):
        """widen(net_idx, NetStreams): optional hook that may enlarge ``first`` / ``mask2`` of a net to the nearest
        stream set libndq.so has kernels for (slots are assigned after it ran).
        unify(streams dict): optional hook run before ``widen`` that may give several networks one common stream set
        (so that one multi-network closure kernel can serve them).
        allow_lap(net_idx, coords) -> bool: may the second derivatives of net k w.r.t. ``coords`` be merged into one
        Laplacian stream (asked only after the merge has been proven valid symbolically)."""
        assert loss in self.LOSS_KINDS and (loss == "custom") == (loss_term is not None)
        # per-point loss term (losses.py:4-12): sum r^2 | sum |r| | max |r|, averaged by the host; "custom": the traced
        # per-point term of a user loss_fn (+ additional_loss), node ``loss_term`` -- loss = mean over points of it
        self.loss = loss
        self.g = graph
        self.residuals = list(residuals)
        self.funcs = list(funcs)
        self.loss_term = loss_term
        if allow_lap is not None:
            # the loss term is held to the same standard as the residuals: it may see pure second derivatives of a
            # network only through their sum if those are to travel as one Laplacian stream
            extra = [loss_term] if loss_term is not None else []
            merged = self._merge_laplacian(self.residuals + extra, self.funcs, n_nets, allow_lap)
            self.residuals = merged[:len(self.residuals)]
            if extra:
                self.loss_term = merged[-1]
        self.n_nets = n_nets                    # parameter sets
        self.site_net = list(getattr(graph, "site_net", None) or range(n_nets))
        self.n_sites = len(self.site_net)       # (network, coordinate tuple) pairs: stream arrays are per site
        self.n_coords = graph.n_coords
        self.n_data = len(getattr(graph, "data", ()))       # per-point data columns: input rows behind the coordinates
        self.n_theta = len(getattr(graph, "params", ()))    # trainable scalars of the equations: kernel arguments
        self.order = graph.reachable(self.residuals + self.funcs + ([self.loss_term] if self.loss_term is not None else []))
        self.streams = {}
        for k in range(self.n_sites):
            deps = graph.net_deps.get(k)
            if deps is None:
                continue
            self.streams[k] = NetStreams(deps, graph.net_nout[k])
        self.symbols = []          # net nodes in use, in order
        for i in self.order:
            n = graph.nodes[i]
            if n[0] == "net":
                self.streams[n[1]].need(n[3])
                self.symbols.append(i)
        if unify is not None:
            unify(self.streams)
        if widen is not None:
            for k, st in self.streams.items():
                widen(k, st)
        self.source = self._emit()
        self.key = hashlib.sha1((self.source + _build_tag()).encode()).hexdigest()[:16]

    # ---- Laplacian-stream rewrite
    def _merge_laplacian(self, residuals, funcs, n_nets, allow_lap):
        """If every residual depends on the pure second derivatives N_aa of a network output only through their SUM
        (d r / d N_aa is the same expression for all a, and free of the N_bb), replace them by one symbol
        L = sum_a N_aa: r(N_xx, N_yy, ..) = r0 + c (N_xx + N_yy + ..) = r(L, 0, ..).  Proven on the DAG, per network."""
        g = self.g
        order = g.reachable(list(residuals) + list(funcs))
        second = {}
        third = set()
        for i in order:
            n = g.nodes[i]
            if n[0] == "net" and len(n[3]) == 2 and n[3][0] != "L":
                second.setdefault(n[1], []).append(i)
            if n[0] == "net" and len(n[3]) >= 3 and n[3][0] != "L":
                third.add(n[1])          # third-order streams need every pair on its own: no Laplacian stream
        for k in third:
            second.pop(k, None)
        func_set = set(g.reachable(list(funcs)))
        out = list(residuals)
        for k, leaves in second.items():
            mis = [g.nodes[i][3] for i in leaves]
            if any(a != b for a, b in mis) or len(leaves) < 2:
                continue                                     # mixed partials present, or nothing to merge
            if any(i in func_set for i in leaves):
                continue
            by_out = {}
            for i in leaves:
                by_out.setdefault(g.nodes[i][2], []).append(i)
            coords = sorted({mi[0] for mi in mis})
            if any(sorted(g.nodes[i][3][0] for i in ls) != coords for ls in by_out.values()):
                continue                                     # every output must use the same coordinate set
            leafset = set(leaves)
            ok = True
            for r in out:
                for o, ls in by_out.items():
                    ds = [g.diff(r, ("n", i)) for i in ls]
                    if any(d != ds[0] for d in ds) or any(j in leafset for j in g.reachable([ds[0]])):
                        ok = False
            if not ok or not allow_lap(k, tuple(coords)):
                continue
            mapping = {}
            for o, ls in by_out.items():
                lap = g.net(k, o, ("L",) + tuple(coords))
                for j, i in enumerate(sorted(ls, key=lambda i: g.nodes[i][3])):
                    mapping[i] = lap if j == 0 else g.const(0.0)
            memo = {}
            out = [g.subst(r, mapping, memo) for r in out]
        return out

    # ---- helpers
    def _val(self, i):
        n = self.g.nodes[i]
        if n[0] == "const":
            return _lit(n[1])
        if n[0] == "coord":
            return f"c{n[1]}"
        if n[0] == "data":
            return f"d{n[1]}"
        if n[0] == "param":
            return f"t{n[1]}"
        return f"v{i}"

    def sym_location(self, i):
        _, k, o, mi = self.g.nodes[i]
        st = self.streams[k]
        return k, st.slot(mi) * st.n_out + o

    def _emit_point_fn(self):
        g = self.g
        L = []
        # which nodes depend on a network symbol (only those carry adjoints)
        dep = {}
        for i in self.order:
            n = g.nodes[i]
            # (runtime constants -- frozen 'param' leaves, symbolic.Graph.external -- carry no adjoint)
            dep[i] = n[0] == "net" or (n[0] == "param" and n[1] not in getattr(g, "frozen", ())) or any(dep[c] for c in g.children(i))
        res_order = g.reachable(self.residuals)
        res_set = set(res_order)
        L.append("// ---- forward")
        for i in self.order:
            n = g.nodes[i]
            op = n[0]
            if op in ("const", "coord", "data", "param"):
                continue
            if op == "net":
                L.append(f"  const float v{i} = s[{self.symbols.index(i)}];")
                continue
            if op in ("add", "sub", "mul", "div"):
                sym = {"add": "+", "sub": "-", "mul": "*", "div": "/"}[op]
                e = f"{self._val(n[1])} {sym} {self._val(n[2])}"
            elif op == "powi":
                e = _powi_expr(self._val(n[1]), n[2])
            elif op == "powc":
                e = f"powf({self._val(n[1])}, {_lit(n[2])})"
            elif op == "atan2":
                e = f"atan2f({self._val(n[1])}, {self._val(n[2])})"
            elif op in ("gt", "ge"):          # masks: 1.0f / 0.0f
                e = f"(({self._val(n[1])} {'>' if op == 'gt' else '>='} {self._val(n[2])}) ? 1.0f : 0.0f)"
            elif op == "where":               # a select, not a blend (symbolic.BINARY)
                e = f"(({self._val(n[1])} != 0.0f) ? {self._val(n[2])} : {self._val(n[3])})"
            else:
                e = _UN_FWD[op].format(a=self._val(n[1]))
            L.append(f"  const float v{i} = {e};")
        for e, i in enumerate(self.residuals):
            L.append(f"  r[{e}] = {self._val(i)};")
        if self.loss == "custom":
            L.append(f"  r[{len(self.residuals)}] = {self._val(self.loss_term)};      // per-point loss term")
        for m, i in enumerate(self.funcs):
            L.append(f"  f[{m}] = {self._val(i)};")
        L.append("  if (!want_adj) return;")
        L.append(f"// ---- adjoint of the per-point loss term ({self.loss}, scaled by seed) w.r.t. the network streams")
        terms = {}
        neq = len(self.residuals)
        if self.loss == "custom":
            res_order = g.reachable([self.loss_term])
            res_set = set(res_order)
            if dep.get(self.loss_term):
                terms[self.loss_term] = ["seed"]
        if self.loss == "infinity" and neq > 1:      # d max_e |r_e|: the first maximal entry carries the seed
            L.append("  int amax = 0; float vmax = fabsf(r[0]);")
            L.append(f"  for (int e = 1; e < {neq}; ++e) if (fabsf(r[e]) > vmax) {{ vmax = fabsf(r[e]); amax = e; }}")
        for e, i in enumerate(self.residuals if self.loss != "custom" else []):
            if dep.get(i):
                v = self._val(i)
                sign = f"(({v}) > 0.0f ? seed : (({v}) < 0.0f ? -seed : 0.0f))"
                if self.loss == "l2":
                    t = f"(2.0f*seed)*{v}"
                elif self.loss == "l1" or neq == 1:
                    t = sign
                else:
                    t = f"(amax == {e} ? {sign} : 0.0f)"
                terms.setdefault(i, []).append(t)
        for i in reversed(res_order):
            if i not in terms or not dep[i]:
                continue
            n = g.nodes[i]
            op = n[0]
            L.append(f"  const float b{i} = {' + '.join(terms[i])};")
            b = f"b{i}"
            if op in ("net", "param"):
                continue

            def push(child, expr):
                if dep[child]:
                    terms.setdefault(child, []).append(expr)

            if op == "add":
                push(n[1], b); push(n[2], b)
            elif op == "sub":
                push(n[1], b); push(n[2], f"-{b}")
            elif op == "mul":
                push(n[1], f"{b}*{self._val(n[2])}"); push(n[2], f"{b}*{self._val(n[1])}")
            elif op == "div":
                push(n[1], f"{b}/{self._val(n[2])}")
                push(n[2], f"-{b}*v{i}/{self._val(n[2])}")
            elif op == "atan2":
                L.append(f"  const float q{i} = {b}/({self._val(n[1])}*{self._val(n[1])} + {self._val(n[2])}*{self._val(n[2])});")
                push(n[1], f"q{i}*{self._val(n[2])}"); push(n[2], f"-q{i}*{self._val(n[1])}")
            elif op in ("gt", "ge"):
                pass                          # piecewise constant: nothing flows through a mask
            elif op == "where":
                push(n[2], f"(({self._val(n[1])} != 0.0f) ? {b} : 0.0f)")
                push(n[3], f"(({self._val(n[1])} != 0.0f) ? 0.0f : {b})")
            elif op == "powi":
                push(n[1], f"{b}*{float(n[2])}f*{_powi_expr(self._val(n[1]), n[2] - 1)}")
            elif op == "powc":
                push(n[1], f"{b}*{_lit(n[2])}*powf({self._val(n[1])}, {_lit(n[2] - 1.0)})")
            else:
                t = _UN_ADJ[op]
                if t is not None:
                    push(n[1], t.format(b=b, a=self._val(n[1]), v=f"v{i}"))
        for idx, i in enumerate(self.symbols):
            L.append(f"  g[{idx}] = {'b%d' % i if (i in terms and i in res_set) else '0.0f'};")
        # per-point adjoints of the trainable scalars: behind the stream adjoints in g (summed over the points by the kernels)
        nsym = len(self.symbols)
        for j in range(self.n_theta):
            i = g._ids.get(("param", j))
            L.append(f"  g[{nsym + j}] = {'b%d' % i if (i is not None and i in terms and i in res_set) else '0.0f'};")
        return "\n".join(L)

    def point_fn_source(self):
        nc = self.n_coords
        body = self._emit_point_fn()
        neq = len(self.residuals)
        if self.loss == "custom":
            term = f"r[{neq}]"
        elif self.loss == "l2":
            term = " + ".join(f"r[{e}]*r[{e}]" for e in range(neq)) or "0.0f"
        elif self.loss == "l1":
            term = " + ".join(f"fabsf(r[{e}])" for e in range(neq)) or "0.0f"
        else:
            term = "0.0f"
            for e in range(neq):
                term = f"fmaxf({term}, fabsf(r[{e}]))"
        nd, nt = self.n_data, self.n_theta
        return f"""// c: {nc} coordinate(s) | {nd} per-point data value(s) | {nt} trainable scalar(s) of the equations;  g: adjoints of the
// {len(self.symbols)} stream symbol(s) | of the {nt} trainable scalar(s)
NDQ_PW_INLINE void ndq_pw_point(const float* c, const float* s, float seed, int want_adj, float* r, float* f, float* g) {{
{chr(10).join(f"  const float c{i} = c[{i}];" for i in range(nc))}
{chr(10).join(f"  const float d{j} = c[{nc + j}];" for j in range(nd))}
{chr(10).join(f"  const float t{j} = c[{nc + nd + j}];" for j in range(nt))}
{chr(10).join(f"  const float c{i} = {_lit(v)};   // virtual coordinate (boundary value)" for i, v in sorted(self.g.vcoords.items()))}
{body}
}}
// per-point loss term ({self.loss}); the host averages it: seed = 1 / (N * n_eq) for l2 / l1, 1 / N for infinity
NDQ_PW_INLINE float ndq_pw_loss(const float* r) {{ return {term}; }}
"""

    @property
    def loss_norm(self):
        """what the sum over points of the per-point loss term is divided by, per point"""
        return 1 if self.loss in ("infinity", "custom") else max(len(self.residuals), 1)

    @property
    def n_r(self):
        """length of the per-point ``r`` array: residuals (+ the loss term of a traced custom loss)"""
        return max(len(self.residuals) + (1 if self.loss == "custom" else 0), 1)

    def fused_source(self, desc, f64=False):
        """Source of the single-launch closure kernel (csrc/ndq_mlp.h: fused_closure_kernel for one network,
        fused_multi_closure_kernel for 2..4 networks of one shape and stream set) specialised with this program's
        per-point function.  desc: the ndq_mlp_desc shared by all networks.
        f64: the same module in double (csrc/ndq_mlp.h under NDQ_F64: per-point GEMMs on the f64 MFMA, no bf16 planes, no
        pull / loop mode) -- one network, plain closure only (``can_fuse_f64``)."""
        if f64:
            return "#define NDQ_F64 1\n" + source_f64(self._fused_source(desc, True))
        return self._fused_source(desc, False)

    def _fused_source(self, desc, f64):
        K = self.n_nets
        mode = fuse_mode(self, {k: desc for k in range(K)})
        assert mode is not None and not (f64 and (mode != "tile" or K != 1))
        if mode in ("group", "wide"):
            return self._group_source(desc, wide=(mode == "wide"))
        ns = self.streams[0].n_streams
        nsym = max(len(self.symbols), 1)
        jet = (lambda k, loc: f"jets[{loc}]") if K == 1 else (lambda k, loc: f"jets[{k}][{loc}]")
        gj = (lambda k, loc: f"gj[{loc}]") if K == 1 else (lambda k, loc: f"gj[{k}][{loc}]")
        loads, stores = [], []
        used = {}
        for idx, i in enumerate(self.symbols):
            k, loc = self.sym_location(i)
            used[(k, loc)] = idx
            loads.append(f"    s[{idx}] = {jet(k, loc)};")
        for k in range(K):
            for loc in range(ns):
                stores.append(f"    {gj(k, loc)} = {'g[%d]' % used[(k, loc)] if (k, loc) in used else '0.0f'};")
        header = "ndq_mlp.h"                 # found through -I csrc (_hipcc.BASE_FLAGS): sources and cache keys do not depend on the checkout path
        neq, nf = len(self.residuals), len(self.funcs)
        jets_t = "const float (&jets)[CFG::NS]" if K == 1 else f"const float (&jets)[{K}][CFG::NS]"
        gj_t = "float (&gj)[CFG::NS]" if K == 1 else f"float (&gj)[{K}][CFG::NS]"
        kern = (lambda train: f"ndq::fused_closure_kernel<CFG, PW, {train}>") if K == 1 else \
            (lambda train: f"ndq::fused_multi_closure_kernel<CFG, {K}, PW, {train}>")
        lds = (lambda train: f"ndq::fused_lds_bytes<CFG>({train})") if K == 1 else \
            (lambda train: f"(ndq::fused_multi_lds_bytes<CFG, {K}>({train}))")
        # workgroup shape: waves x tiles per round (multi-network closure: K x G waves, G tile slots -- csrc/ndq_mlp.h)
        threads = "CFG::BWD_THREADS" if K == 1 else f"ndq::multi_threads<{K}>()"
        tiles_per_block = "CFG::BWD_THREADS / 64" if K == 1 else f"ndq::multi_group<{K}>()"
        if K == 1:
            args_t = "ndq::FusedArgs"
            fill = "a.params = params[0]; a.partials = partials ? partials[0] : nullptr;"
            kern_tv = "ndq::fused_closure_tv_kernel<CFG, PW>"
        else:
            args_t = "ndq::FusedMultiArgs"
            fill = f"for (int k = 0; k < {K}; ++k) {{ a.params[k] = params[k]; a.partials[k] = partials ? partials[k] : nullptr; }}"
            kern_tv = f"ndq::fused_multi_closure_tv_kernel<CFG, {K}, PW>"
        if f64:
            kern_loop = lds_loop = None          # (loop / pull mode: fp32 only, csrc/ndq_tail.h)
        elif K == 1:
            kern_loop, lds_loop = "ndq::fused_closure_loop_kernel<CFG, PW>", "ndq::fused_loop_lds_bytes<CFG>()"
        elif K == 2:
            kern_loop, lds_loop = f"ndq::fused_multi_closure_loop_kernel<CFG, {K}, PW>", f"(ndq::fused_multi_loop_lds_bytes<CFG, {K}>())"
        else:
            kern_loop = lds_loop = None
        tv = _tv_launcher(args_t, fill, kern_tv, lds('true'), lds('false'), threads, kern_loop, lds_loop)
        # hidden-layer weight gradients from the bf16x3 planes through transposing LDS reads (csrc/ndq_mlp.h Cfg::WG_TR;
        # shapes it does not cover, or whose K images would not fit the LDS, ignore the switch)
        wg_tr = f"#ifndef NDQ_WG_TR\n#define NDQ_WG_TR 1\n#endif\n#define NDQ_WG_TR_K {K}\n" + CLOSURE_PRODUCTS
        return f"""// GENERATED by neurodiffeq_amd/codegen.py -- single-launch closure kernel (forward streams + pointwise stage +
// reverse pass) of one PDE system with {K} network(s), gfx950.
#include <cstdlib>
{wg_tr}#include "{header}"
#define NDQ_PW_INLINE __device__ __forceinline__
#ifndef NDQ_MAX_BLOCKS
#define NDQ_MAX_BLOCKS 256     // closure workgroups per launch (one per CU; experiments: 512 = two 8-wave workgroups per CU)
#endif
{self.point_fn_source()}
namespace {{
using CFG = ndq::Cfg<{desc.d}, {desc.first}, {desc.mask2}u, {(desc.hidden + 15) // 16}, {desc.layers}, {desc.act}, 1, {desc.lap}, {desc.skip}, {desc.mask3}u, {desc.actp}, {desc.hidden if (desc.hidden % 16 or desc.widths) else 0}, {desc.widths}u, {desc.mono}u{_m4_arg(desc)}>;
struct PW {{
  // ND per-point data columns (rows D .. D + ND of the coordinate block), NT trainable scalars of the equations
  static constexpr int NEQ = {neq}, NF = {nf}, NR = {self.n_r}, ND = {self.n_data}, NT = {self.n_theta};
  static __device__ __forceinline__ float loss(const float* r) {{ return ndq_pw_loss(r); }}
  // xe: the point's ND data values, then the NT scalars;  gth (want_adj): per-point adjoints of the scalars
  static __device__ __forceinline__ void apply(const float (&x)[CFG::D], const float* xe, {jets_t}, float seed,
                                               int want_adj, float (&r)[{self.n_r}], float (&f)[{max(nf, 1)}],
                                               {gj_t}, float* gth) {{
    float c[CFG::D + ND + NT], s[{nsym}], g[{nsym} + NT];
#pragma unroll
    for (int d = 0; d < CFG::D; ++d) c[d] = x[d];
#pragma unroll
    for (int j = 0; j < ND + NT; ++j) c[CFG::D + j] = xe[j];
{chr(10).join(loads)}
    ndq_pw_point(c, s, seed, want_adj, r, f, g);
{chr(10).join(stores)}
#pragma unroll
    for (int j = 0; j < NT; ++j) gth[j] = g[{len(self.symbols)} + j];
  }}
}};
constexpr int kWaves = {tiles_per_block};        // tiles a workgroup handles per round
int fused_blocks(int n) {{
  const int tiles = (n + 15) / 16;
  int b = (tiles + kWaves - 1) / kWaves;
  return b > NDQ_MAX_BLOCKS ? NDQ_MAX_BLOCKS : (b < 1 ? 1 : b);
}}

// trainable scalars of the equations (PW::NT of them): values read by every launch, block sums of their adjoints written by
// training launches -- bound by the engine before it launches (ndq_fused_bind_theta)
const float* g_theta = nullptr;
float* g_theta_partials = nullptr;

// params / partials: host arrays of {K} device pointers (one per network)
int launch(const float* coords, int ldc, int n, const float* const* params, float* const* partials, float* loss_partials,
           float* funcs, float* resid, int ldj, float seed, int train, void* stream) {{
  if (!coords || !params || !loss_partials || n <= 0 || ldc < n || (train && !partials)) return -2;
  if (PW::NT > 0 && !g_theta) return -2;
  {args_t} a{{}};
  a.coords = coords; a.loss_partials = loss_partials;
  {fill}
  a.funcs = funcs; a.resid = resid; a.n = n; a.ldc = ldc; a.ldj = ldj; a.seed = seed;
  a.theta = g_theta; a.theta_partials = train ? g_theta_partials : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern('true')}),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds('true')});
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern('false')}),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds('false')});
    if (e != hipSuccess) return (int)e;
    attr = true;
  }}
  if (train)
    hipLaunchKernelGGL(({kern('true')}), dim3(fused_blocks(n)), dim3({threads}), {lds('true')}, s, a);
  else
    hipLaunchKernelGGL(({kern('false')}), dim3(fused_blocks(n)), dim3({threads}), {lds('false')}, s, a);
  return (int)hipGetLastError();
}}
{tv}
}}  // namespace

extern "C" int ndq_fused_blocks(int n) {{ return fused_blocks(n); }}
extern "C" int ndq_fused_num_params() {{ return CFG::P; }}
extern "C" int ndq_fused_num_theta() {{ return PW::NT; }}
extern "C" void ndq_fused_bind_theta(const float* theta, float* theta_partials) {{ g_theta = theta; g_theta_partials = theta_partials; }}
extern "C" int ndq_fused_num_nets() {{ return {K}; }}
extern "C" int ndq_fused_threads() {{ return {threads}; }}
extern "C" unsigned long ndq_fused_lds_bytes() {{ return (unsigned long){lds('true')}; }}

#ifdef NDQ_PHASE_TS
extern "C" int ndq_fused_phase_ts(unsigned long long* out) {{    // experiments: scripts/phase_ts.py
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndq::ndq_phase_ts), sizeof(unsigned long long) * 256 * 8);
}}
extern "C" int ndq_fused_pull_ts(unsigned long long* out) {{    // experiments: scripts/pull_ts.py
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndq::ndq_pull_ts), sizeof(unsigned long long) * 8);
}}
extern "C" int ndq_fused_tile_ts(unsigned long long* out) {{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndq::ndq_tile_ts), sizeof(unsigned long long) * 48);
}}
#endif

// one network: the ndq_fused_launch_fn of include/ndq.h
extern "C" int ndq_fused_launch(const float* coords, int ldc, int n, const float* params, float* partials,
                                float* loss_partials, float* funcs, float* resid, int ldj, float seed, int train,
                                void* stream) {{
  if ({K} != 1) return -2;
  const float* pp[1] = {{params}};
  float* qq[1] = {{partials}};
  return launch(coords, ldc, n, pp, partials ? qq : nullptr, loss_partials, funcs, resid, ldj, seed, train, stream);
}}

// any number of networks: params / partials are host arrays of device pointers
extern "C" int ndq_fused_launch_multi(const float* coords, int ldc, int n, const float* const* params,
                                      float* const* partials, float* loss_partials, float* funcs, float* resid, int ldj,
                                      float seed, int train, void* stream) {{
  return launch(coords, ldc, n, params, partials, loss_partials, funcs, resid, ldj, seed, train, stream);
}}
{_TV_EXPORT}"""

    def _group_source(self, desc, wide=False):
        """Source of the grouped single-launch closure kernel (csrc/ndq_mlp.h: fused_group_closure_kernel): one network
        with any number of outputs, reading any subset of the batch coordinates; the per-point function runs on one
        point per lane, stream values and adjoint seeds are exchanged through an LDS tile.
        ``wide``: the closure kernel of csrc/ndq_wide.h (one hidden layer of 65 .. 512 units) -- the same per-point
        interface, stream rows [n_streams][n_out] without padding."""
        st = self.streams[0]
        width = st.n_streams * st.n_out
        gw = 1 if st.n_out == 1 else (st.n_out + 15) // 16 * 16      # row layout [n_streams][gw] (ndq::group_w)
        if wide:
            gw = st.n_out
        row = lambda loc: (loc // st.n_out) * gw + loc % st.n_out
        nsym = max(len(self.symbols), 1)
        used = {}
        loads = []
        for idx, i in enumerate(self.symbols):
            k, loc = self.sym_location(i)
            used[loc] = idx
            loads.append(f"    s[{idx}] = srow[{row(loc)}];")
        stores = [f"    grow[{row(loc)}] = {'g[%d]' % used[loc] if loc in used else '0.0f'};" for loc in range(width)]
        deps = list(st.deps)
        dep_fn = " : ".join(f"d == {d} ? {c}" for d, c in enumerate(deps)) + " : 0"
        header = "ndq_mlp.h"                 # found through -I csrc (_hipcc.BASE_FLAGS): sources and cache keys do not depend on the checkout path
        neq, nf = len(self.residuals), len(self.funcs)
        kern = lambda train: f"ndq::fused_group_closure_kernel<CFG, PW, {train}>"
        lds = lambda train: f"ndq::group_lds_bytes<CFG>({train})"
        kern_tv = "ndq::fused_group_closure_tv_kernel<CFG, PW>"
        cfg_t = (f"ndq::Cfg<{desc.d}, {desc.first}, {desc.mask2}u, {(desc.hidden + 15) // 16}, {desc.layers}, {desc.act}, "
                 f"{desc.n_out}, {desc.lap}, {desc.skip}, {desc.mask3}u, {desc.actp}, "
                 f"{desc.hidden if (desc.hidden % 16 or desc.widths) else 0}, {desc.widths}u, {desc.mono}u{_m4_arg(desc)}>")
        blocks_body = """  constexpr int gp = 16 * ndq::group_tiles<CFG>();
  const int groups = (n + gp - 1) / gp;
  int b = (groups + kWaves - 1) / kWaves;"""
        if wide:
            header = "ndq_wide.h"
            kern = lambda train: f"ndq::wide_closure_kernel<CFG, PW, {train}>"
            lds = lambda train: "(ndq::wide_closure_lds_bytes<CFG, PW>())"
            kern_tv = "ndq::wide_closure_tv_kernel<CFG, PW>"
            cfg_t = wide_cfg(desc)
            blocks_body = """  const int tiles = (n + 15) / 16;
  int b = (tiles + kWaves - 1) / kWaves;"""
        tv = _tv_launcher("ndq::FusedArgs", "a.params = params[0]; a.partials = partials ? partials[0] : nullptr;",
                          kern_tv, lds('true'), lds('false'))
        return f"""// GENERATED by neurodiffeq_amd/codegen.py -- grouped single-launch closure kernel (forward streams -> LDS exchange ->
// per-point stage, one point per lane -> reverse pass) of one PDE system, gfx950.
#include <cstdlib>
{CLOSURE_PRODUCTS}#include "{header}"
#define NDQ_PW_INLINE __device__ __forceinline__
#ifndef NDQ_MAX_BLOCKS
#define NDQ_MAX_BLOCKS 256     // closure workgroups per launch (one per CU; experiments: 512 = two 8-wave workgroups per CU)
#endif
{self.point_fn_source()}
namespace {{
using CFG = {cfg_t};
static_assert(CFG::NS * CFG::NOUT == {width}, "stream layout of the traced program and of the kernel disagree");
struct PW {{
  // NC rows of the coordinate block per point: the batch coordinates, then ND data columns; NT trainable scalars
  static constexpr int NEQ = {neq}, NF = {nf}, ND = {self.n_data}, NT = {self.n_theta}, NC = {self.n_coords} + ND, NR = {self.n_r};
  static constexpr int dep(int d) {{ return {dep_fn}; }}     // batch coordinate fed to network input d
  static __device__ __forceinline__ float loss(const float* r) {{ return ndq_pw_loss(r); }}
  static __device__ __forceinline__ void apply(const float (&cc)[NC], const float* th, const float* srow, float seed, int want_adj,
                                               float (&r)[{self.n_r}], float (&f)[{max(nf, 1)}], float* grow, float* gth) {{
    float c[NC + NT], s[{nsym}], g[{nsym} + NT];
#pragma unroll
    for (int d = 0; d < NC; ++d) c[d] = cc[d];
#pragma unroll
    for (int j = 0; j < NT; ++j) c[NC + j] = th[j];
{chr(10).join(loads)}
    ndq_pw_point(c, s, seed, want_adj, r, f, g);
    if (!want_adj) return;
{chr(10).join(stores)}
#pragma unroll
    for (int j = 0; j < NT; ++j) gth[j] = g[{len(self.symbols)} + j];
  }}
}};
constexpr int kWaves = CFG::BWD_THREADS / 64;
int fused_blocks(int n) {{
{blocks_body}
  return b > NDQ_MAX_BLOCKS ? NDQ_MAX_BLOCKS : (b < 1 ? 1 : b);
}}

const float* g_theta = nullptr;          // see the tile-closure module: trainable scalars of the equations
float* g_theta_partials = nullptr;

int launch(const float* coords, int ldc, int n, const float* const* params, float* const* partials, float* loss_partials,
           float* funcs, float* resid, int ldj, float seed, int train, void* stream) {{
  if (!coords || !params || !loss_partials || n <= 0 || ldc < n || (train && !partials)) return -2;
  if (PW::NT > 0 && !g_theta) return -2;
  ndq::FusedArgs a{{}};
  a.coords = coords; a.loss_partials = loss_partials;
  a.params = params[0]; a.partials = partials ? partials[0] : nullptr;
  a.funcs = funcs; a.resid = resid; a.n = n; a.ldc = ldc; a.ldj = ldj; a.seed = seed;
  a.theta = g_theta; a.theta_partials = train ? g_theta_partials : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr = false;
  if (!attr) {{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern('true')}),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds('true')});
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kern('false')}),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int){lds('false')});
    if (e != hipSuccess) return (int)e;
    attr = true;
  }}
  if (train)
    hipLaunchKernelGGL(({kern('true')}), dim3(fused_blocks(n)), dim3(CFG::BWD_THREADS), {lds('true')}, s, a);
  else
    hipLaunchKernelGGL(({kern('false')}), dim3(fused_blocks(n)), dim3(CFG::BWD_THREADS), {lds('false')}, s, a);
  return (int)hipGetLastError();
}}
{tv}
}}  // namespace

extern "C" int ndq_fused_blocks(int n) {{ return fused_blocks(n); }}
extern "C" int ndq_fused_num_params() {{ return CFG::P; }}
extern "C" int ndq_fused_num_theta() {{ return PW::NT; }}
extern "C" void ndq_fused_bind_theta(const float* theta, float* theta_partials) {{ g_theta = theta; g_theta_partials = theta_partials; }}
extern "C" int ndq_fused_num_nets() {{ return 1; }}
extern "C" int ndq_fused_threads() {{ return CFG::BWD_THREADS; }}
extern "C" unsigned long ndq_fused_lds_bytes() {{ return (unsigned long){lds('true')}; }}

#ifdef NDQ_PHASE_TS
extern "C" int ndq_fused_phase_ts(unsigned long long* out) {{    // experiments: scripts/phase_ts_group.py
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndq::ndq_phase_ts), sizeof(unsigned long long) * 256 * 8);
}}
extern "C" int ndq_fused_tile_ts(unsigned long long* out) {{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ndq::ndq_tile_ts), sizeof(unsigned long long) * 48);
}}
#endif

extern "C" int ndq_fused_launch(const float* coords, int ldc, int n, const float* params, float* partials,
                                float* loss_partials, float* funcs, float* resid, int ldj, float seed, int train,
                                void* stream) {{
  const float* pp[1] = {{params}};
  float* qq[1] = {{partials}};
  return launch(coords, ldc, n, pp, partials ? qq : nullptr, loss_partials, funcs, resid, ldj, seed, train, stream);
}}

extern "C" int ndq_fused_launch_multi(const float* coords, int ldc, int n, const float* const* params,
                                      float* const* partials, float* loss_partials, float* funcs, float* resid, int ldj,
                                      float seed, int train, void* stream) {{
  return launch(coords, ldc, n, params, partials, loss_partials, funcs, resid, ldj, seed, train, stream);
}}
{_TV_EXPORT}"""

    def _emit(self):
        nsym = max(len(self.symbols), 1)
        neq, nf, nc, nn = len(self.residuals), len(self.funcs), self.n_coords, self.n_sites
        loads, stores = [], []
        for idx, i in enumerate(self.symbols):
            k, loc = self.sym_location(i)
            loads.append(f"    s[{idx}] = a.jets[{k}][(size_t){loc} * a.ldj + n];")
        # adjoint stores: every slot of every net is written (unused streams get 0 so the backward kernel can read them)
        for k, st in sorted(self.streams.items()):
            used = {}
            for idx, i in enumerate(self.symbols):
                kk, loc = self.sym_location(i)
                if kk == k:
                    used[loc] = idx
            for loc in range(st.n_streams * st.n_out):
                val = f"g[{used[loc]}]" if loc in used else "0.0f"
                stores.append(f"      a.gbar[{k}][(size_t){loc} * a.ldj + n] = {val};")
        src = f"""// GENERATED by neurodiffeq_amd/codegen.py -- fused pointwise stage of one traced PDE system (gfx950).
// per point: {nc} coords + {len(self.symbols)} stream symbols in, {neq} residual(s), {nf} function value(s), adjoints out.
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define NDQ_PW_INLINE __device__ __forceinline__
#else
#include <math.h>
#include <stddef.h>
#define NDQ_PW_INLINE static inline
#endif

#define NDQ_PW_NC {nc}
#define NDQ_PW_ND {self.n_data}
#define NDQ_PW_NT {self.n_theta}
#define NDQ_PW_NSYM {len(self.symbols)}
#define NDQ_PW_NEQ {neq}
#define NDQ_PW_NR {self.n_r}
#define NDQ_PW_NF {nf}
#define NDQ_PW_NNETS {nn}

{self.point_fn_source()}
#ifdef __HIPCC__
struct PwArgs {{
  const float* coords;
  const float* jets[NDQ_PW_NNETS];
  float* gbar[NDQ_PW_NNETS];
  float* funcs;
  float* resid;
  float* loss_partials;
  int n, ldc, ldj, want_adj;
  float seed;
  const float* theta;          // [NDQ_PW_NT] trainable scalars of the equations
  float* theta_partials;       // [gridDim.x][NDQ_PW_NT] block sums of their per-point adjoints
}};

extern "C" __global__ __launch_bounds__(256) void ndq_pw_kernel(PwArgs a) {{
  const int n = blockIdx.x * 256 + threadIdx.x;
  float sq = 0.f;
  float gt[NDQ_PW_NT > 0 ? NDQ_PW_NT : 1] = {{0.f}};
  if (n < a.n) {{
    float c[NDQ_PW_NC + NDQ_PW_ND + NDQ_PW_NT], s[{nsym}], r[{self.n_r}], f[{max(nf, 1)}], g[{nsym} + NDQ_PW_NT];
#pragma unroll
    for (int i = 0; i < NDQ_PW_NC + NDQ_PW_ND; ++i) c[i] = a.coords[(size_t)i * a.ldc + n];   // data rows follow the coordinates
#pragma unroll
    for (int j = 0; j < NDQ_PW_NT; ++j) c[NDQ_PW_NC + NDQ_PW_ND + j] = a.theta[j];
{chr(10).join(loads)}
    ndq_pw_point(c, s, a.seed, a.want_adj, r, f, g);
    if (a.want_adj) {{
#pragma unroll
      for (int j = 0; j < NDQ_PW_NT; ++j) gt[j] = g[NDQ_PW_NSYM + j];
    }}
    sq += ndq_pw_loss(r);
    if (a.resid) {{
#pragma unroll
      for (int e = 0; e < NDQ_PW_NEQ; ++e) a.resid[(size_t)e * a.ldj + n] = r[e];
    }}
    if (a.funcs) {{
#pragma unroll
      for (int m = 0; m < NDQ_PW_NF; ++m) a.funcs[(size_t)m * a.ldj + n] = f[m];
    }}
    if (a.want_adj) {{
{chr(10).join(stores)}
    }}
  }}
  // fixed-order block reduction of the squared residuals: wave shuffle tree, then the 4 waves in order
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
#if NDQ_PW_NT > 0
  // the same fixed-order block sums for the adjoints of the trainable scalars
  __shared__ float tsum[4][NDQ_PW_NT];
#pragma unroll
  for (int j = 0; j < NDQ_PW_NT; ++j) {{
    float v = gt[j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) tsum[threadIdx.x >> 6][j] = v;
  }}
  __syncthreads();
  if (threadIdx.x < NDQ_PW_NT && a.want_adj && a.theta_partials)
    a.theta_partials[(size_t)blockIdx.x * NDQ_PW_NT + threadIdx.x] =
        (tsum[0][threadIdx.x] + tsum[1][threadIdx.x]) + (tsum[2][threadIdx.x] + tsum[3][threadIdx.x]);
#endif
}}

extern "C" int ndq_pw_blocks(int n) {{ return (n + 255) / 256; }}
extern "C" int ndq_pw_num_theta() {{ return NDQ_PW_NT; }}
extern "C" int ndq_pw_num_data() {{ return NDQ_PW_ND; }}

// trainable scalars of the equations: values read by every launch, block sums of their adjoints written by training launches
static const float* g_theta = nullptr;
static float* g_theta_partials = nullptr;
extern "C" void ndq_pw_bind_theta(const float* theta, float* theta_partials) {{ g_theta = theta; g_theta_partials = theta_partials; }}

extern "C" int ndq_pw_launch(const float* coords, int ldc, int n, const float* const* jets, float* const* gbar, int ldj,
                             float* funcs, float* resid, float* loss_partials, float seed_scale, void* stream) {{
  if (!coords || !jets || !loss_partials || n <= 0) return -2;
  if (NDQ_PW_NT > 0 && !g_theta) return -2;
  PwArgs a;
  a.theta = g_theta; a.theta_partials = g_theta_partials;
  a.coords = coords;
  for (int k = 0; k < NDQ_PW_NNETS; ++k) {{
    a.jets[k] = jets[k];
    a.gbar[k] = gbar ? gbar[k] : nullptr;
  }}
  a.funcs = funcs; a.resid = resid; a.loss_partials = loss_partials;
  a.n = n; a.ldc = ldc; a.ldj = ldj; a.want_adj = gbar ? 1 : 0; a.seed = seed_scale;
  hipLaunchKernelGGL(ndq_pw_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}}
#endif
"""
        return src

    # ---- algorithmic HBM traffic per point (bytes), for the roofline report
    def bytes_per_point(self, train=True, write_resid=False, write_funcs=False):
        b = 4 * self.n_coords + 4 * len(self.symbols)
        if train:
            b += 4 * sum(st.n_streams * st.n_out for st in self.streams.values())
        if write_resid:
            b += 4 * len(self.residuals)
        if write_funcs:
            b += 4 * len(self.funcs)
        return b


# ----------------------------------------------------------------------------------------------- build / load
class PointwiseKernel:
    def __init__(self, so_path, f64=False):
        self.path = so_path
        self.lib = ctypes.CDLL(so_path)
        self.lib.ndq_pw_launch.restype = ctypes.c_int
        self.lib.ndq_pw_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_double if f64 else ctypes.c_float, ctypes.c_void_p]
        self.lib.ndq_pw_blocks.restype = ctypes.c_int
        self.lib.ndq_pw_blocks.argtypes = [ctypes.c_int]

    def blocks(self, n):
        return self.lib.ndq_pw_blocks(n)


def so_path_for(program: PointwiseProgram, f64=False):
    return os.path.join(JIT_DIR, f"pw{'64' if f64 else ''}_{program.key}.so")


_F64_FUNCS = re.compile(r"\b(pow|sin|cos|tan|exp|log|tanh|sqrt|fabs|sinh|cosh|fmax|log1p|expm1|erf|atan|atan2|floor|ceil|rint|trunc)f\(")
_F64_LITERAL = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?)f\b")


def source_f64(source):
    """The generated pointwise source in double precision (the reference's default dtype, neurodiffeq/__init__.py:22):
    the emitter writes ``float``, ``expf(..)``-style calls and ``1.0f``-style literals only, so the fp64 build is a
    textual rewrite of the same program -- types, math calls, literal suffixes."""
    out = re.sub(r"\bfloat\b", "double", source)
    out = _F64_FUNCS.sub(lambda m: m.group(1) + "(", out)
    return _F64_LITERAL.sub(lambda m: m.group(1), out)


def build(program: PointwiseProgram, force=False, f64=False):
    """Compile the generated source for gfx950 (in-tree cache keyed by the source hash) and return the .so path."""
    os.makedirs(JIT_DIR, exist_ok=True)
    so = so_path_for(program, f64)
    src = so[:-3] + ".hip"
    if os.path.exists(so) and not force:
        return so
    with open(src, "w") as fh:
        fh.write(source_f64(program.source) if f64 else program.source)
    try:
        _hipcc.compile_shared(src, so, defer=True)
    except RuntimeError as e:
        raise RuntimeError(f"hipcc failed for generated pointwise kernel {src}:\n{str(e)[-4000:]}") from e
    return so


def load(program: PointwiseProgram, f64=False):
    return PointwiseKernel(build(program, f64=f64), f64)


# ----------------------------------------------------------------------------------------------- fused closure kernel
class FusedKernel:
    def __init__(self, so_path, f64=False):
        self.path = so_path
        self.lib = ctypes.CDLL(so_path)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, (ctypes.c_double if f64 else ctypes.c_float)
        self.lib.ndq_fused_launch.restype = ci
        self.lib.ndq_fused_launch.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, cf, ci, vp]
        self.lib.ndq_fused_launch_multi.restype = ci
        self.lib.ndq_fused_launch_multi.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, cf, ci, vp]
        self.lib.ndq_fused_launch_tv.restype = ci
        self.lib.ndq_fused_launch_tv.argtypes = [vp, ci, ci, vp, vp, vp, cf, vp, ci, ci, vp, vp, vp]
        self.lib.ndq_fused_pull_ok.restype = ci
        self.lib.ndq_fused_launch_loop.restype = ci
        self.lib.ndq_fused_launch_loop.argtypes = [vp, ci, ci, cf, vp, ci, ci, vp, vp]
        self.lib.ndq_fused_loop_ok.restype = ci
        self.lib.ndq_fused_bind_theta.restype = None
        self.lib.ndq_fused_bind_theta.argtypes = [vp, vp]
        self.lib.ndq_fused_blocks.restype = ci
        self.lib.ndq_fused_blocks.argtypes = [ci]
        self.lib.ndq_fused_num_params.restype = ci
        self.lib.ndq_fused_num_nets.restype = ci
        self.lib.ndq_fused_threads.restype = ci
        self.threads = self.lib.ndq_fused_threads()
        self.lib.ndq_fused_lds_bytes.restype = ctypes.c_ulong
        self.n_nets = self.lib.ndq_fused_num_nets()

    def blocks(self, n):
        return self.lib.ndq_fused_blocks(n)


def _cache_key(source):
    """Cache key of a generated module: its source (with the package location factored out, so a relocated checkout
    keeps its cache), the kernel headers it instantiates and the extra compile flags."""
    text = source.replace(HERE, "<neurodiffeq_amd>")
    return hashlib.sha1((text + _header_digest() + " ".join(_extra_flags()) + _build_tag()).encode()).hexdigest()[:16]


def _build_tag():
    """What the build pipeline itself contributes to a cache key (the assembly fix-up pass and its version)."""
    return _hipcc.FIXUP_VERSION if _hipcc.fixup_enabled() else "no-fixup"


def _header_digest():
    h = hashlib.sha1()
    for name in ("csrc/ndq_mlp.h", "csrc/ndq_tail.h", "csrc/ndq_launch.h", "csrc/ndq_wide.h", "csrc/ndq_deep.h", "../include/ndq.h"):
        with open(os.path.join(HERE, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# ----------------------------------------------------------------------------------------------- MLP kernel extensions
# libndq.so carries a table of (shape, stream set) kernel pairs; anything else the templates can express is compiled
# on first use as a tiny extension module (one Cfg: forward + adjoint kernel) and registered with libndq.so, after
# which the ordinary C-ABI entry points serve it.  Cached in-tree like the generated pointwise kernels.
_MLP_EXT = {}


MAX_INPUTS = 6          # network inputs the templates are instantiated for (Streams<D, ...>: D (D + 1) / 2 pair bits)
MAX_LAYERS = 8          # hidden layers of one width (networks.describe); the LDS footprint decides at registration


def mlp_ext_allowed(desc):
    """Can ndq_mlp.h express this descriptor?  (H = 1..64, 1..8 hidden layers of one width or 1..4 of different widths,
    d <= 6; the LDS footprint is checked by ndq_mlp_register once the module is built.)"""
    if os.environ.get("NDQ_JIT_MLP", "1") == "0":
        return False
    npair = desc.d * (desc.d + 1) // 2
    diag = 0
    for a in range(desc.d):
        diag |= 1 << pair_list(desc.d).index((a, a))
    if desc.mask3:              # third order: tanh / sin / sigmoid, every triple with its three pairs, no Laplacian stream,
        #                         d <= 4 (five inputs have more triples than the 32 mask bits)
        if desc.lap or desc.d > 4 or desc.act not in (0, 1, 2, 5, 6, 7) or desc.mask3 >> len(triple_list(desc.d)):
            return False
        for k, (a, b, c) in enumerate(triple_list(desc.d)):
            if (desc.mask3 >> k) & 1 and not all((desc.mask2 >> pair_list(desc.d).index(p)) & 1
                                                   for p in ((a, b), (a, c), (b, c))):
                return False
    if desc.mask4:              # fourth order (round 6): tanh / sin / sigmoid, d <= 3, every quadruple with its pairs and triples
        if (desc.lap or desc.d > 3 or desc.act not in (0, 1, 2) or desc.actp or desc.mono or is_wide(desc)
                or desc.mask4 >> len(quad_list(desc.d))):
            return False
        for k, q in enumerate(quad_list(desc.d)):
            if (desc.mask4 >> k) & 1:
                for i in range(4):
                    tri = tuple(q[j] for j in range(4) if j != i)
                    if not (desc.mask3 >> triple_list(desc.d).index(tri)) & 1:
                        return False
                    for j in range(i + 1, 4):
                        if not (desc.mask2 >> pair_list(desc.d).index((q[i], q[j]))) & 1:
                            return False
    if desc.widths:           # per-layer widths: 1..hidden each, the widest equal to `hidden`, nothing beyond `layers`
        bits, most = (10, 3) if is_wide(desc) else (8, 4)      # (csrc/ndq_deep.h: 10 bits x 2 .. 3 layers; csrc/ndq_mlp.h: 8 x 4)
        ws = [(desc.widths >> (bits * l)) & ((1 << bits) - 1) for l in range(most)]
        if (desc.widths >> (bits * most)) or desc.layers > most or any(w for w in ws[desc.layers:]) or min(ws[:desc.layers]) < 1 \
                or max(ws[:desc.layers]) != desc.hidden or (is_wide(desc) and desc.layers < 2):
            return False
    if desc.mono and (not 0 < desc.mono < 256 or desc.mask3 or desc.skip or desc.hidden > 48):
        return False          # monomial features: degrees 1..8, up to second order, H <= 48, no skip connection
    if is_wide(desc) and (desc.skip or desc.actp or desc.mono or desc.n_out > 16):
        return False          # wider than 64 units (csrc/ndq_wide.h: one hidden layer; csrc/ndq_deep.h: 2 .. 8): plain FCNN
    return (1 <= desc.d <= MAX_INPUTS and 1 <= desc.hidden <= MAX_HIDDEN and 1 <= desc.layers <= (4 if desc.widths else MAX_LAYERS)
            and desc.act in (0, 1, 2, 3, 4, 5, 6, 7) and 1 <= desc.n_out <= 64 and desc.first in (0, 1)
            and 0 <= desc.mask2 < (1 << npair) and (desc.first == 1 or desc.mask2 == 0)
            and (desc.lap == 0 or (desc.n_out == 1 and desc.mask2 != 0 and (desc.mask2 & ~diag) == 0))
            and desc.skip in (0, 1) and desc.actp in (0, 1, 2) and (desc.actp == 0 or desc.act in (3, 4)))


MAX_HIDDEN = 512        # include/ndq.h NDQ_MAX_HIDDEN


def is_wide(desc):
    """Shapes served by csrc/ndq_wide.h (units over lanes, weights in registers) instead of csrc/ndq_mlp.h (16-point MFMA
    fragments, weights resident in LDS): hidden layers wider than 64 units."""
    return desc.hidden > 64


def deep_cfg(desc):
    """The ndq::DeepCfg instantiation of a descriptor (2 .. 8 hidden layers of 65 .. 512 units, csrc/ndq_deep.h)."""
    return (f"ndq::DeepCfg<{desc.d}, {desc.first}, {desc.mask2}u, {desc.lap}, {desc.mask3}u, {desc.hidden}, {desc.layers}, "
            f"{desc.act}, {desc.n_out}" + (f", {desc.widths}u>" if desc.widths else ">"))


def wide_cfg(desc):
    """The ndq::WideCfg instantiation of a descriptor (one hidden layer of 65 .. 512 units)."""
    return (f"ndq::WideCfg<{desc.d}, {desc.first}, {desc.mask2}u, {desc.lap}, {desc.mask3}u, {desc.hidden}, {desc.act}, "
            f"{desc.n_out}>")


#: extra hipcc flags of modules built from csrc/ndq_wide.h: its tile code is 16 rounds x U units of independent scalar
#: chains; the SLP vectoriser pairs them ACROSS rounds into v_pk_* (no faster on gfx950: a packed fp32 op issues like two),
#: which drags every round's live values to the end of the tile -- 512 VGPRs + 200 .. 700 spilled against 227 without
WIDE_FLAGS = ["-fno-slp-vectorize"]


def padded_width(hidden):
    """Width the kernels lay a hidden layer out for: the next multiple of 16 (csrc/ndq_mlp.h: Cfg::H vs Cfg::HR)."""
    return (hidden + 15) // 16 * 16


def mlp_ext_source(desc, f64=False):
    header = "ndq_launch.h"              # -I csrc, as above
    record = "ndq64_mlp_kernels" if f64 else "ndq_mlp_kernels"
    if is_wide(desc) and desc.layers >= 2:
        return f"""// GENERATED by neurodiffeq_amd/codegen.py -- layer-by-layer forward / adjoint kernels of one deep FCNN wider than 64
// units (csrc/ndq_deep.h)
#include "{header}"
using CFG = {deep_cfg(desc)};
extern "C" const {record}* ndq_ext_kernels(void) {{
  static const {record} k = ndq::make_deep_kernels<CFG>();
  return &k;
}}
// 1: the next adjoint call may use the activations the last forward call left in the workspace (same parameters, same batch)
extern "C" void ndq_ext_reuse_forward(int on) {{ ndq::deep_last().reuse = on != 0; }}
"""
    if is_wide(desc):
        return f"""// GENERATED by neurodiffeq_amd/codegen.py -- forward-stream and adjoint kernels of one single-hidden-layer FCNN wider
// than 64 units (csrc/ndq_wide.h)
{"#define NDQ_F64 1" if f64 else ""}
#include "{header}"
using CFG = {wide_cfg(desc)};
extern "C" const {record}* ndq_ext_kernels(void) {{
  static const {record} k = ndq::make_wide_kernels<CFG>();
  return &k;
}}
"""
    return f"""// GENERATED by neurodiffeq_amd/codegen.py -- forward-stream and adjoint kernels of one FCNN shape / stream set
{"#define NDQ_F64 1" if f64 else ""}
#include "{header}"
using CFG = ndq::Cfg<{desc.d}, {desc.first}, {desc.mask2}u, {(desc.hidden + 15) // 16}, {desc.layers}, {desc.act}, {desc.n_out}, {desc.lap}, {desc.skip}, {desc.mask3}u, {desc.actp}, {desc.hidden if (desc.hidden % 16 or desc.widths) else 0}, {desc.widths}u, {desc.mono}u{_m4_arg(desc)}>;
extern "C" const {record}* ndq_ext_kernels(void) {{
  static const {record} k = ndq::make_kernels<CFG>();
  return &k;
}}
"""


def build_mlp_ext(desc, force=False, f64=False):
    os.makedirs(JIT_DIR, exist_ok=True)
    source = mlp_ext_source(desc, f64)
    key = _cache_key(source)
    so = os.path.join(JIT_DIR, f"mlp{'64' if f64 else ''}_{key}.so")
    src = os.path.join(JIT_DIR, f"mlp{'64' if f64 else ''}_{key}.hip")
    if os.path.exists(so) and not force:
        return so
    with open(src, "w") as fh:
        fh.write(source)
    try:
        _hipcc.compile_shared(src, so, _extra_flags() + (WIDE_FLAGS if is_wide(desc) and desc.layers == 1 else []))
    except RuntimeError as e:
        raise RuntimeError(f"hipcc failed for MLP kernel extension {src}:\n{str(e)[-4000:]}") from e
    return so


def ensure_mlp_kernels(desc, f64=False):
    """True if libndq.so (``f64``: libndq64.so) can serve ``desc`` -- from its table, or after building + registering an
    extension module."""
    from . import _lib
    L = _lib.lib64() if f64 else _lib.lib()
    supported = L.ndq64_mlp_supported if f64 else L.ndq_mlp_supported
    register = L.ndq64_mlp_register if f64 else L.ndq_mlp_register
    if supported(ctypes.byref(desc)):
        return True
    # fp64: twice the LDS per weight -- ndq64_mlp_register turns down what does not fit; wider than 64 units only ONE hidden
    # layer (csrc/ndq_wide.h compiled in double: forward-stream and adjoint kernels; csrc/ndq_deep.h is fp32 only)
    if not mlp_ext_allowed(desc) or (f64 and desc.hidden > 64 and desc.layers != 1):
        return False
    key = desc.key() + (("f64",) if f64 else ())
    if key in _MLP_EXT:
        return _MLP_EXT[key] is not None
    _MLP_EXT[key] = None
    try:
        ext = ctypes.CDLL(build_mlp_ext(desc, f64=f64))
    except RuntimeError as e:                     # the templates reject this combination: not a supported shape
        import warnings
        warnings.warn(f"could not build an MLP kernel extension for {key}: {str(e)[:400]}")
        return False
    except OSError as e:                          # no hipcc / unreadable module: the native path is broken, say so
        raise _lib.NdqError(f"cannot build or load the MLP kernel extension for {key}: {e}") from e
    ext.ndq_ext_kernels.restype = ctypes.c_void_p
    if register(ctypes.c_void_p(ext.ndq_ext_kernels())) != 0:
        return False                              # e.g. the shape needs more LDS than a workgroup has
    _MLP_EXT[key] = ext                           # keep the module (and its record) alive
    return bool(supported(ctypes.byref(desc)))


def mlp_ext_module(desc, f64=False):
    """The loaded extension module serving ``desc`` (None: the descriptor is served by libndq.so's own table)."""
    return _MLP_EXT.get(desc.key() + (("f64",) if f64 else ()))


def can_fuse_f64(program, descs):
    """fp64 systems: the single-launch closure exists for ONE network on the plain closure kernel (no grouped / wide /
    multi-network variant in double)."""
    return program.n_nets == 1 and can_fuse(program, descs) and fuse_mode(program, descs) == "tile" and descs[0].hidden <= 64


def build_fused(program: PointwiseProgram, desc, force=False, threads=None, f64=False):
    """Compile the fused closure kernel of a single-network system for gfx950 (in-tree cache keyed by the generated
    source AND the kernel header it instantiates).  ``threads=512``: the 8-wave build (two waves per SIMD where the
    per-wave state fits 256 registers; csrc/ndq_mlp.h NDQ_BWD_THREADS) the engine uses for large batches."""
    os.makedirs(JIT_DIR, exist_ok=True)
    source = program.fused_source(desc, f64=f64)
    flags = _extra_flags() + ([f"-DNDQ_BWD_THREADS={int(threads)}"] if threads else []) + (WIDE_FLAGS if is_wide(desc) else [])
    key = _cache_key(source + (f"|threads={int(threads)}" if threads else ""))
    so = os.path.join(JIT_DIR, f"fused_{key}.so")
    src = os.path.join(JIT_DIR, f"fused_{key}.hip")
    if os.path.exists(so) and not force:
        return so
    with open(src, "w") as fh:
        fh.write(source)
    try:
        _hipcc.compile_shared(src, so, flags, defer=True)
    except RuntimeError as e:
        raise RuntimeError(f"hipcc failed for generated fused kernel {src}:\n{str(e)[-4000:]}") from e
    return so


def fuse_mode(program: PointwiseProgram, descs=None):
    """Which single-launch closure kernel serves the system, if any:
    "tile"   one single-output network reading all coordinates (fused_closure_kernel),
    "multi"  2..4 such networks of ONE shape and stream set, H <= 48 (fused_multi_closure_kernel),
    "group"  one network with several outputs and / or reading only some of the batch coordinates, H <= 48
             (fused_group_closure_kernel: per-point stage on one point per lane through an LDS exchange tile);
             NDQ_FUSE_GROUP=1 sends every eligible single-network system there.
    None     three-kernel pipeline."""
    K = program.n_nets
    if K < 1 or K > 4 or program.n_sites != K or any(k not in program.streams for k in range(K)):
        return None
    plain = all(tuple(program.streams[k].deps) == tuple(range(program.n_coords)) and program.streams[k].n_out == 1
                for k in range(K))
    if K == 1:
        st = program.streams[0]
        d0 = descs[0] if descs is not None and 0 in descs else None
        if d0 is not None and is_wide(d0):
            # one hidden layer wider than 64 units: csrc/ndq_wide.h's closure kernel (any number of outputs, any subset of
            # the batch coordinates as inputs)
            return "wide" if (d0.layers == 1 and all(c < program.n_coords for c in st.deps)
                              and not os.environ.get("NDQ_NO_WIDE_FUSE")) else None
        group_ok = (d0 is not None and padded_width(d0.hidden) <= 48 and all(c < program.n_coords for c in st.deps)
                    and not os.environ.get("NDQ_NO_GROUP_FUSE"))
        if plain and not (group_ok and os.environ.get("NDQ_FUSE_GROUP") == "1"):
            return "tile"
        return "group" if group_ok else None
    if not plain:
        return None
    if descs is None or len({descs[k].key() for k in range(K)}) != 1 or padded_width(descs[0].hidden) > 48:
        return None
    if os.environ.get("NDQ_NO_MULTI_FUSE"):
        return None
    return "multi"


def can_fuse(program: PointwiseProgram, descs=None):
    return fuse_mode(program, descs) is not None
