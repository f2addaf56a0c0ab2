"""Test infrastructure: compile the per-point function of a generated pointwise kernel (neurodiffeq_amd/codegen.py)
with gcc and run it over a batch on the host, so tracer + symbolic differentiation + adjoint code generation can be
checked against the oracle without a GPU.  The product never loads this."""
import ctypes
import hashlib
import os
import subprocess
import tempfile

import numpy as np

_DRIVER = r"""
/* coords: the NDQ_PW_NC coordinate rows followed by the NDQ_PW_ND data rows; theta: the NDQ_PW_NT trainable scalars;
   gtheta [NT][n]: their per-point adjoints (the kernels sum these over the points) */
void pw_cpu_run_ex(const float* coords, const float* syms, const float* theta, int n, float seed, int want_adj,
                   float* resid, float* funcs, float* gbar, float* lossterm, float* gtheta) {
  for (int i = 0; i < n; ++i) {
    float c[NDQ_PW_NC + NDQ_PW_ND + NDQ_PW_NT + 1], s[NDQ_PW_NSYM > 0 ? NDQ_PW_NSYM : 1];
    float r[NDQ_PW_NR], f[NDQ_PW_NF > 0 ? NDQ_PW_NF : 1], g[NDQ_PW_NSYM + NDQ_PW_NT + 1];
    for (int k = 0; k < NDQ_PW_NC + NDQ_PW_ND; ++k) c[k] = coords[(size_t)k * n + i];
    for (int k = 0; k < NDQ_PW_NT; ++k) c[NDQ_PW_NC + NDQ_PW_ND + k] = theta[k];
    for (int k = 0; k < NDQ_PW_NSYM; ++k) s[k] = syms[(size_t)k * n + i];
    ndq_pw_point(c, s, seed, want_adj, r, f, g);
    if (want_adj && gtheta) for (int k = 0; k < NDQ_PW_NT; ++k) gtheta[(size_t)k * n + i] = g[NDQ_PW_NSYM + k];
    for (int k = 0; k < NDQ_PW_NEQ; ++k) resid[(size_t)k * n + i] = r[k];
    lossterm[i] = ndq_pw_loss(r);
    for (int k = 0; k < NDQ_PW_NF; ++k) funcs[(size_t)k * n + i] = f[k];
    if (want_adj) for (int k = 0; k < NDQ_PW_NSYM; ++k) gbar[(size_t)k * n + i] = g[k];
  }
}
void pw_cpu_run(const float* coords, const float* syms, int n, float seed, int want_adj,
                float* resid, float* funcs, float* gbar, float* lossterm) {
  pw_cpu_run_ex(coords, syms, 0, n, seed, want_adj, resid, funcs, gbar, lossterm, 0);
}
"""


def compile_cpu(program, f64=False):
    src = program.source + _DRIVER
    if f64:                     # the fp64 build of the same program (codegen.source_f64: what FusedSystem(dtype=float64) compiles)
        from neurodiffeq_amd.codegen import source_f64
        src = source_f64(src)
    key = hashlib.sha1(src.encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "ndq_pw_cpu")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, f"pw_{key}.so")
    if not os.path.exists(so):
        c = os.path.join(d, f"pw_{key}.c")
        with open(c, "w") as fh:
            fh.write(src)
        subprocess.run(["gcc", "-O2", "-std=c99", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", c, "-o", so,
                        "-lm"], check=True)
    lib = ctypes.CDLL(so)
    return lib


def run_cpu(program, coords, syms, seed, want_adj=True, return_loss=False, f64=False, data=None, theta=None):
    """coords [nc][n], syms [nsym][n] (order = program.symbols; fp32, or fp64 with ``f64``) -> resid [neq][n],
    funcs [nf][n], gbar [nsym][n].  Systems with per-point data columns / trainable scalars: ``data`` [n_data][n],
    ``theta`` [n_theta]; the per-point adjoints of the scalars [n_theta][n] are appended to the result."""
    lib = compile_cpu(program, f64)
    dt = np.float64 if f64 else np.float32
    coords = np.ascontiguousarray(coords, dtype=dt)
    if program.n_data or program.n_theta:
        if program.n_data:
            coords = np.ascontiguousarray(np.concatenate([coords, np.asarray(data, dt).reshape(program.n_data, -1)]), dtype=dt)
        theta = np.ascontiguousarray(theta, dtype=dt)
        syms = np.ascontiguousarray(syms, dtype=dt)
        n = coords.shape[1]
        resid, funcs = np.zeros((len(program.residuals), n), dt), np.zeros((len(program.funcs), n), dt)
        gbar, lossterm = np.zeros((max(len(program.symbols), 1), n), dt), np.zeros(n, dt)
        gth = np.zeros((max(program.n_theta, 1), n), dt)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib.pw_cpu_run_ex.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double if f64 else ctypes.c_float, ctypes.c_int] \
            + [ctypes.c_void_p] * 5
        lib.pw_cpu_run_ex(p(coords), p(syms), p(theta), n, seed, int(want_adj), p(resid), p(funcs), p(gbar), p(lossterm), p(gth))
        return (resid, funcs, gbar, lossterm, gth) if return_loss else (resid, funcs, gbar, gth)
    syms = np.ascontiguousarray(syms, dtype=dt)
    n = coords.shape[1]
    resid = np.zeros((len(program.residuals), n), dt)
    funcs = np.zeros((len(program.funcs), n), dt)
    gbar = np.zeros((max(len(program.symbols), 1), n), dt)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lossterm = np.zeros(n, dt)
    lib.pw_cpu_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double if f64 else ctypes.c_float, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.pw_cpu_run(p(coords), p(syms), n, seed, int(want_adj), p(resid), p(funcs), p(gbar), p(lossterm))
    return (resid, funcs, gbar, lossterm) if return_loss else (resid, funcs, gbar)
