"""ORACLE (test infrastructure, never shipped): closed-form "jet" restatement of the fused HIP kernels, numpy fp64.

The reference obtains d^k u / dx^k by k autograd graph walks (neurodiffeq.py:21-34) through
``FCNN.forward`` (networks.py:59-70).  The HIP path instead propagates the value and the first/second
partial derivatives ("streams") of every hidden unit forward through the MLP and reverses those recurrences
by hand.  This file states exactly that algorithm, one numpy statement per formula,
so the kernels can be checked stage by stage:

* ``mlp_jets``      -- forward streams of the raw network output,
* ``mlp_jets_vjp``  -- parameter gradient given the adjoint of every output stream.

Parity status: PINNED via ``tests/test_oracle_golden.py`` -- streams are compared with ``ref_diff`` of
``oracle/autograd_ref.py`` (itself pinned to the reference's golden vectors) and the VJP with torch autograd.

A stream is a tuple of coordinate indices: ``()`` value, ``(a,)`` d/dx_a, ``(a, b)`` d2/dx_a dx_b (a <= b), up to four
indices (round 6: ``diff(u, x, order=4)`` -- beam / biharmonic equations; neurodiffeq.py:21-34 has no order limit).  Orders 3
and 4 are stated through the general Faa di Bruno sum over the set partitions of the index POSITIONS (``_partitions``), which
handles repeated indices by itself; orders 1 and 2 keep their written-out formulas.
"""
import numpy as np


def act_deriv4(name, z):
    """Fourth derivative of the activation (the adjoint of a third-order stream needs it)."""
    if name == "tanh":
        t = np.tanh(z)
        s1 = 1 - t * t
        return 4 * t * s1 * (1 - 3 * t * t) + 12 * t * s1 * s1
    if name == "sin":
        return np.sin(z)
    if name == "sigmoid":
        s = 1.0 / (1.0 + np.exp(-z))
        d1 = s * (1 - s)
        return d1 * (1 - 2 * s) * (1 - 12 * d1)
    if name == "elu":
        return np.where(z > 0, 0.0, np.exp(np.minimum(z, 0.0)))
    if name == "softplus":
        s = 1.0 / (1.0 + np.exp(-z))
        d = s * (1 - s)
        return d * (1 - 6 * d)
    if name == "gelu":
        phi = np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi)
        return phi * (-z ** 4 + 7 * z ** 2 - 4)
    raise KeyError(f"no fourth derivative stated for {name}")


def act_deriv5(name, z):
    """Fifth derivative of the activation (the adjoint of a fourth-order stream needs it)."""
    if name == "tanh":       # d/dz = (1 - t^2) d/dt:  P5 = (1 - t^2)(16 - 120 t^2 + 120 t^4)
        t = np.tanh(z)
        return (1 - t * t) * (16 - 120 * t ** 2 + 120 * t ** 4)
    if name == "sin":
        return np.cos(z)
    if name == "sigmoid":    # d/dz = (s - s^2) d/ds applied to Q4 = (s - s^2)(1 - 14 s + 36 s^2 - 24 s^3)
        s = 1.0 / (1.0 + np.exp(-z))
        d1 = s * (1 - s)
        return d1 * ((1 - 2 * s) * (1 - 14 * s + 36 * s ** 2 - 24 * s ** 3) + d1 * (-14 + 72 * s - 72 * s ** 2))
    raise KeyError(f"no fifth derivative stated for {name}")


def _partitions(k):
    """Set partitions of the positions 0 .. k-1 (1, 2, 5, 15 of them for k = 1 .. 4), each a tuple of sorted blocks."""
    if k == 0:
        return [()]
    out = []
    for part in _partitions(k - 1):
        out.append(part + ((k - 1,),))
        for i in range(len(part)):
            out.append(part[:i] + (part[i] + (k - 1,),) + part[i + 1:])
    return out


def _sub(m, block):
    return tuple(sorted(m[i] for i in block))


def act_derivs(name, z, theta=None):
    """sigma(z) and its first three derivatives.  ``theta``: the layer's activation parameters
    -- (beta,) for swish, (alpha, beta, gamma) for aptx (networks.py:155-209); None = their defaults."""
    if theta is not None and name == "swish":            # z s(beta z): Leibniz with S^(k)(z) = beta^k s^(k)(beta z)
        (beta,) = theta
        s = 1.0 / (1.0 + np.exp(-beta * z))
        d1 = s * (1 - s)
        S0, S1, S2, S3 = s, beta * d1, beta ** 2 * d1 * (1 - 2 * s), beta ** 3 * d1 * (1 - 6 * d1)
        return z * S0, S0 + z * S1, 2 * S1 + z * S2, 3 * S2 + z * S3
    if theta is not None and name == "aptx":             # gamma z (alpha + tanh(beta z))
        alpha, beta, gamma = theta
        t = np.tanh(beta * z)
        u1 = 1 - t * t
        U0, U1, U2, U3 = alpha + t, beta * u1, beta ** 2 * (-2 * t * u1), beta ** 3 * (-2 * u1 * (1 - 3 * t * t))
        return gamma * z * U0, gamma * (U0 + z * U1), gamma * (2 * U1 + z * U2), gamma * (3 * U2 + z * U3)
    if name == "tanh":
        t = np.tanh(z)
        s1 = 1 - t * t
        return t, s1, -2 * t * s1, -2 * s1 * (1 - 3 * t * t)
    if name == "sin":
        s, c = np.sin(z), np.cos(z)
        return s, c, -s, -c
    if name in ("sigmoid", "swish"):
        s = 1.0 / (1.0 + np.exp(-z))
        d1 = s * (1 - s)
        d2 = d1 * (1 - 2 * s)
        d3 = d1 * (1 - 6 * d1)
        if name == "sigmoid":
            return s, d1, d2, d3
        # Leibniz on z * s(z)  (Swish, beta = 1: networks.py:155-175)
        return z * s, s + z * d1, 2 * d1 + z * d2, 3 * d2 + z * d3
    if name == "elu":         # torch.nn.ELU, alpha = 1: z (z > 0), e^z - 1 (z <= 0)
        e = np.exp(np.minimum(z, 0.0))
        pos = z > 0
        return np.where(pos, z, e - 1), np.where(pos, 1.0, e), np.where(pos, 0.0, e), np.where(pos, 0.0, e)
    if name == "softplus":    # torch.nn.Softplus, beta = 1 (the threshold = 20 branch is the same function to 2e-9)
        s = 1.0 / (1.0 + np.exp(-z))
        d = s * (1 - s)
        return np.logaddexp(0.0, z), s, d, d * (1 - 2 * s)
    if name == "gelu":        # torch.nn.GELU, exact: z Phi(z)
        from math import sqrt, pi
        try:
            from scipy.special import erf
        except Exception:     # pragma: no cover
            erf = np.vectorize(__import__("math").erf)
        Phi = 0.5 * (1 + erf(z / sqrt(2.0)))
        phi = np.exp(-0.5 * z * z) / sqrt(2 * pi)
        return z * Phi, Phi + z * phi, phi * (2 - z * z), phi * (z ** 3 - 4 * z)
    if name == "aptx":        # z (1 + tanh z) / 2: APTx with alpha = 1, beta = 1, gamma = 1/2 (networks.py:177-209); Leibniz
        t = np.tanh(z)
        u, u1 = 1 + t, 1 - t * t
        u2, u3 = -2 * t * u1, -2 * u1 * (1 - 3 * t * t)
        return 0.5 * z * u, 0.5 * (u + z * u1), 0.5 * (2 * u1 + z * u2), 0.5 * (3 * u2 + z * u3)
    raise KeyError(name)


def split_params(flat, dims):
    """flat -> [(W (out,in), b (out,)), ...] in torch order."""
    out, off = [], 0
    for a, b in zip(dims[:-1], dims[1:]):
        w = flat[off:off + a * b].reshape(b, a); off += a * b
        bias = flat[off:off + b]; off += b
        out.append((w, bias))
    assert off == flat.size
    return out


def _pairs_of(m):
    return [(m[0], m[1]), (m[0], m[2]), (m[1], m[2])]


def close_streams(streams):
    """A second-order stream needs both of its first-order streams, a third-order one also its three second-order
    sub-streams, a fourth-order one its six pairs and four triples (Faa di Bruno: every sub-multi-index); the value stream
    is always present."""
    s = {()}
    for m in streams:
        m = tuple(sorted(m))
        assert len(m) <= 4, "order > 4 is outside the fused path"
        for part in _partitions(len(m)):
            for block in part:
                s.add(_sub(m, block))
    return sorted(s, key=lambda m: (len(m), m))


def _input_streams(coords, streams, mono=None):
    """Streams of what the first linear layer sees.  Plain network: value = x, d/dx_a = e_a, higher orders 0.  ``mono``
    (degrees of a networks.MonomialNN in front, networks.py:109-139): the features x_a^deg, degree after degree, and
    their derivatives -- d^k/dx_a^k x_a^deg = deg!/(deg-k)! x_a^(deg-k), every mixed derivative 0."""
    x = np.stack([np.asarray(c, dtype=np.float64).reshape(-1) for c in coords], axis=1)   # (N, d)
    n, d = x.shape
    if mono is None:
        h = {m: np.zeros((n, d)) for m in streams}
        h[()] = x
        for m in streams:
            if len(m) == 1:
                h[m][:, m[0]] = 1.0
        return h
    h = {m: np.zeros((n, d * len(mono))) for m in streams}
    for m in streams:
        for i, deg in enumerate(mono):
            for a in range(d):
                if all(idx == a for idx in m) and len(m) <= deg:
                    coef = 1.0
                    for k in range(len(m)):
                        coef *= deg - k
                    h[m][:, i * d + a] = coef * x[:, a] ** (deg - len(m))
    return h


def _mono_dims(dims, mono):
    return dims if mono is None else (dims[0] * len(mono),) + tuple(dims[1:])


def _forward(flat, dims, act, coords, streams, thetas=None, mono=None):
    layers = split_params(np.asarray(flat, dtype=np.float64), _mono_dims(dims, mono))
    h = _input_streams(coords, streams, mono)
    saved = []
    for li, (w, b) in enumerate(layers):
        z = {m: h[m] @ w.T for m in streams}
        z[()] = z[()] + b
        if li == len(layers) - 1:
            return z, saved, layers
        s0, s1, s2, s3 = act_derivs(act, z[()], None if thetas is None else thetas[li])
        hn = {(): s0}
        for m in streams:
            if len(m) == 1:
                hn[m] = s1 * z[m]
            elif len(m) == 2:
                hn[m] = s2 * z[(m[0],)] * z[(m[1],)] + s1 * z[m]
            elif len(m) >= 3:
                # h_abc = s3 z_a z_b z_c + s2 (z_ab z_c + z_ac z_b + z_bc z_a) + s1 z_abc;  h_abcd = s4 z_a z_b z_c z_d
                # + s3 (6 terms z_pair z z) + s2 (3 terms z_pair z_pair + 4 terms z_triple z) + s1 z_abcd: one term per set
                # partition of the positions, sigma^(number of blocks) times the product of the blocks' streams
                sig = {1: s1, 2: s2, 3: s3, 4: act_deriv4(act, z[()]) if len(m) == 4 else None}
                acc = 0.0
                for part in _partitions(len(m)):
                    term = sig[len(part)]
                    for block in part:
                        term = term * z[_sub(m, block)]
                    acc = acc + term
                hn[m] = acc
        saved.append((h, z, (s1, s2, s3)))
        h = hn
    raise AssertionError


def _n_fcnn_params(dims):
    return sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))


def _act_thetas(flat, dims, act, skip, actp):
    """Per-layer activation parameters from the tail of the flat vector (include/ndq.h: ndq_mlp_desc.actp)."""
    if not actp:
        return None
    k = {"swish": 1, "aptx": 3}[act]
    off = _n_fcnn_params(dims) + (dims[-1] * dims[0] if skip else 0)
    n_layers = len(dims) - 2
    assert flat.size == off + k * n_layers
    return [tuple(flat[off + k * l: off + k * (l + 1)]) for l in range(n_layers)]


def mlp_jets(flat, dims, act, coords, streams, skip=False, actp=False, mono=None):
    """Streams of the raw network output: dict stream -> (N, n_out).  ``skip``: Resnet (networks.py:73-106) -- the
    bias-free skip matrix S (n_out, d) follows the FCNN parameters in the flat vector and the output gains S x (value) /
    S[:, a] (d/dx_a).  ``actp``: trainable Swish / APTx parameters, one set per hidden layer, at the end of the vector."""
    streams = close_streams(streams)
    flat = np.asarray(flat, dtype=np.float64)
    if mono is not None:
        assert not skip and not actp
        return _forward(flat, dims, act, coords, streams, None, mono)[0]
    z, _, _ = _forward(flat[:_n_fcnn_params(dims)], dims, act, coords, streams, _act_thetas(flat, dims, act, skip, actp))
    if skip:
        S = flat[_n_fcnn_params(dims):_n_fcnn_params(dims) + dims[-1] * dims[0]].reshape(dims[-1], dims[0])
        x = np.stack([np.asarray(c, dtype=np.float64).reshape(-1) for c in coords], axis=1)
        z = dict(z)
        z[()] = z[()] + x @ S.T
        for m in streams:
            if len(m) == 1:
                z[m] = z[m] + S[:, m[0]]
    return z


def mlp_jets_vjp(flat, dims, act, coords, gbar, skip=False, actp=False, thetas=None, mono=None):
    """Parameter gradient sum_n sum_streams <gbar[stream][n], d out_stream[n] / d params>  (flat, torch order; with
    ``skip`` the gradient of the skip matrix follows, with ``actp`` that of the activation parameters after it).

    ``gbar``: dict stream -> (N, n_out) adjoints."""
    if actp:
        # weights: the same recurrences with the parameterised derivative table; the activation parameters themselves:
        # central differences of the scalar <gbar, streams> in fp64 (a handful of scalars; step 1e-6 -> ~1e-10 relative)
        flat = np.asarray(flat, dtype=np.float64)
        th = _act_thetas(flat, dims, act, skip, True)
        n_lin = _n_fcnn_params(dims) + (dims[-1] * dims[0] if skip else 0)
        base = mlp_jets_vjp(flat[:n_lin], dims, act, coords, gbar, skip=skip, thetas=th)

        def scalar(v):
            out = mlp_jets(v, dims, act, coords, list(gbar.keys()), skip=skip, actp=True)
            return sum(float(np.sum(np.asarray(g, dtype=np.float64) * out[tuple(sorted(m))])) for m, g in gbar.items())
        dth = np.zeros(flat.size - n_lin)
        for i in range(dth.size):
            e = np.zeros_like(flat); e[n_lin + i] = 1e-6
            dth[i] = (scalar(flat + e) - scalar(flat - e)) / 2e-6
        return np.concatenate([base, dth])
    if skip:
        flat = np.asarray(flat, dtype=np.float64)
        base = mlp_jets_vjp(flat[:_n_fcnn_params(dims)], dims, act, coords, gbar, thetas=thetas)
        x = np.stack([np.asarray(c, dtype=np.float64).reshape(-1) for c in coords], axis=1)
        dS = np.asarray(gbar.get((), np.zeros((x.shape[0], dims[-1]))), dtype=np.float64).T @ x      # (n_out, d)
        for m, g in gbar.items():
            if len(m) == 1:
                dS[:, m[0]] += np.asarray(g, dtype=np.float64).sum(axis=0)
        return np.concatenate([base, dS.reshape(-1)])
    streams = close_streams(list(gbar.keys()))
    zlast, saved, layers = _forward(flat, dims, act, coords, streams, thetas, mono)
    n = zlast[()].shape[0]
    zb = {m: np.asarray(gbar.get(m, np.zeros_like(zlast[()])), dtype=np.float64) for m in streams}
    grads = [None] * len(layers)
    # h feeding the last layer is recomputed from the last saved z
    for li in range(len(layers) - 1, -1, -1):
        w, _ = layers[li]
        if li == 0:
            hin = _input_streams(coords, streams, mono)
        else:
            _, zprev, (s1, s2, s3) = saved[li - 1]
            s0 = act_derivs(act, zprev[()], None if thetas is None else thetas[li - 1])[0]
            hin = {(): s0}
            for m in streams:
                if len(m) == 1:
                    hin[m] = s1 * zprev[m]
                elif len(m) == 2:
                    hin[m] = s2 * zprev[(m[0],)] * zprev[(m[1],)] + s1 * zprev[m]
                elif len(m) >= 3:
                    sig = {1: s1, 2: s2, 3: s3, 4: act_deriv4(act, zprev[()]) if len(m) == 4 else None}
                    acc = 0.0
                    for part in _partitions(len(m)):
                        term = sig[len(part)]
                        for block in part:
                            term = term * zprev[_sub(m, block)]
                        acc = acc + term
                    hin[m] = acc
        dw = sum(zb[m].T @ hin[m] for m in streams)
        db = zb[()].sum(axis=0)
        grads[li] = (dw, db)
        if li == 0:
            break
        hb = {m: zb[m] @ w for m in streams}
        _, zprev, (s1, s2, s3) = saved[li - 1]
        nzb = {m: np.zeros_like(hb[()]) for m in streams}
        nzb[()] = s1 * hb[()]
        for m in streams:
            if len(m) == 1:
                nzb[()] += s2 * zprev[m] * hb[m]
                nzb[m] += s1 * hb[m]
        for m in streams:
            if len(m) == 2:
                a, b = (m[0],), (m[1],)
                nzb[()] += (s3 * zprev[a] * zprev[b] + s2 * zprev[m]) * hb[m]
                nzb[a] += s2 * zprev[b] * hb[m]
                nzb[b] += s2 * zprev[a] * hb[m]
                nzb[m] += s1 * hb[m]
        if any(len(m) >= 3 for m in streams):
            s4 = act_deriv4(act, zprev[()])
        for m in streams:
            if len(m) == 4:
                # adjoint of the fourth-order recurrence, partition by partition: a term sigma^(r) prod_B z_B gives
                # sigma^(r+1) prod_B z_B to the value stream's adjoint and sigma^(r) prod_{B' != B} z_B' to block B's
                sig = {1: s1, 2: s2, 3: s3, 4: s4, 5: act_deriv5(act, zprev[()])}
                for part in _partitions(4):
                    r = len(part)
                    zs = [zprev[_sub(m, block)] for block in part]
                    prod = 1.0
                    for v in zs:
                        prod = prod * v
                    nzb[()] += sig[r + 1] * prod * hb[m]
                    for i, block in enumerate(part):
                        rest = sig[r]
                        for j, v in enumerate(zs):
                            if j != i:
                                rest = rest * v
                        nzb[_sub(m, block)] += rest * hb[m]
            if len(m) == 3:         # adjoint of the third-order recurrence, position by position
                a, b, c = (m[0],), (m[1],), (m[2],)
                pab, pac, pbc = _pairs_of(m)
                za, zb_, zc = zprev[a], zprev[b], zprev[c]
                mix = zprev[pab] * zc + zprev[pac] * zb_ + zprev[pbc] * za
                nzb[()] += (s4 * za * zb_ * zc + s3 * mix + s2 * zprev[m]) * hb[m]
                nzb[a] += (s3 * zb_ * zc + s2 * zprev[pbc]) * hb[m]
                nzb[b] += (s3 * za * zc + s2 * zprev[pac]) * hb[m]
                nzb[c] += (s3 * za * zb_ + s2 * zprev[pab]) * hb[m]
                nzb[pab] += s2 * zc * hb[m]
                nzb[pac] += s2 * zb_ * hb[m]
                nzb[pbc] += s2 * za * hb[m]
                nzb[m] += s1 * hb[m]
        zb = nzb
    return np.concatenate([np.concatenate([dw.reshape(-1), db]) for dw, db in grads])
