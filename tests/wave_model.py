"""Test infrastructure: a numpy, lane-by-lane model of ONE wave of the gfx950 kernels in
neurodiffeq_amd/csrc/ndq_mlp.h (same fragment layout, same permuted contraction order, same LDS staging indices,
MFMA 16x16x4 semantics from /opt/skills/guides/cdna_hip_programming.md section 3).

It exists so the index algebra of the kernels can be checked on a machine without a GPU (tests/test_wave_model.py
compares it with oracle/jet_ref.py).  Never imported by the product."""
import numpy as np

LANES = np.arange(64)
P_ = LANES & 15
Q_ = LANES >> 4


def mfma16x16x4(a, b, c):
    """D = A.B + C with A[i][k] at lane i+16k, B[k][j] at lane j+16k, C/D[row=4*(lane>>4)+r][col=lane&15] at (lane, r)."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[P_, Q_] = a
    B[Q_, P_] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * Q_ + r, P_]
    return out


def act_derivs(act, z):
    if act == "tanh":
        t = np.tanh(z); s1 = 1 - t * t
        return t, s1, -2 * t * s1, -2 * s1 * (1 - 3 * t * t)
    if act in ("sigmoid", "swish"):        # the kernel's closed forms in (t, c), csrc/ndq_mlp.h: Act<ACT_SIGMOID / ACT_SWISH>
        c = 1.0 / (1.0 + np.exp(-z))
        if act == "sigmoid":
            s1 = c * (1 - c)
            return c, s1, s1 * (1 - 2 * c), s1 * (1 - 6 * s1)
        t = z * c
        return t, c + t * (1 - c), (1 - c) * (2 * c + t * (1 - 2 * c)), (1 - c) * (3 * c * (1 - 2 * c) + t * (1 - 6 * c + 6 * c * c))
    if act == "aptx":                      # the kernel's closed forms in (z, T = tanh z), Act<ACT_APTX>
        T = np.tanh(z)
        return 0.5 * z * (1 + T), 0.5 * ((1 + T) + z * (1 - T * T)), (1 - T * T) * (1 - z * T), (1 - T * T) * (3 * z * T * T - 3 * T - z)
    s, c = np.sin(z), np.cos(z)
    return s, c, -s, -c


class Model:
    def __init__(self, flat, d, hidden, layers, act, streams2):
        """streams2: list of (a,b) second-order pairs in kernel order."""
        self.d, self.H, self.L, self.act = d, hidden, layers, act
        self.NB = hidden // 16
        self.pairs = list(streams2)
        self.first = True
        self.NS = 1 + d + len(self.pairs)
        H, D = hidden, d
        off = 0
        self.W1 = flat[off:off + H * D].reshape(H, D); off += H * D
        self.b1 = flat[off:off + H]; off += H
        self.W, self.b = {}, {}
        for l in range(2, layers + 1):
            self.W[l] = flat[off:off + H * H].reshape(H, H); off += H * H
            self.b[l] = flat[off:off + H]; off += H
        self.Wout = flat[off:off + H]; off += H
        self.bout = flat[off]; off += 1
        self.P = off
        assert off == flat.size
        # fragment-ordered LDS images (stage_weights)
        self.Wf, self.Wt = {}, {}
        NB = self.NB
        for l in range(2, layers + 1):
            Wl = self.W[l]
            wf = np.zeros(H * H); wt = np.zeros(H * H)
            for i in range(H * H):
                lane, t, blk = i & 63, (i >> 6) & 3, i >> 8
                b0, b1 = blk // NB, blk % NB
                wf[i] = Wl[16 * b0 + (lane & 15), 16 * b1 + 4 * (lane >> 4) + t]
                wt[i] = Wl[16 * b1 + 4 * (lane >> 4) + t, 16 * b0 + (lane & 15)]
            self.Wf[l], self.Wt[l] = wf, wt

    # fragments: arrays [NS][NB][64 lanes][4]
    def frag_units(self, b):
        """unit index held by (lane, r) in block b"""
        return 16 * b + 4 * Q_[:, None] + np.arange(4)[None, :]

    def gemm_frag(self, w, h):
        NB, NS = self.NB, self.NS
        z = np.zeros((NS, NB, 64, 4))
        for ib in range(NB):
            for kb in range(NB):
                for t in range(4):
                    a = w[((ib * NB + kb) * 4 + t) * 64 + LANES]
                    for s in range(NS):
                        z[s, ib] = mfma16x16x4(a, h[s, kb][:, t], z[s, ib])
        return z

    def act_forward(self, st):
        t, s1, s2, _ = st["d"]
        z = st["z"]
        h = np.zeros_like(z)
        h[0] = t
        for a in range(self.d):
            h[1 + a] = s1 * z[1 + a]
        for k, (a, b) in enumerate(self.pairs):
            s = 1 + self.d + k
            h[s] = s2 * z[1 + a] * z[1 + b] + s1 * z[s]
        return h

    def act_backward(self, st, g):
        t, s1, s2, s3 = st["d"]
        z = st["z"]
        out = np.zeros_like(g)
        z0 = s1 * g[0]
        za = [s1 * g[1 + a] for a in range(self.d)]
        for a in range(self.d):
            z0 = z0 + s2 * z[1 + a] * g[1 + a]
        for k, (a, b) in enumerate(self.pairs):
            s = 1 + self.d + k
            hb = g[s]
            z0 = z0 + (s3 * z[1 + a] * z[1 + b] + s2 * z[s]) * hb
            za[a] = za[a] + s2 * z[1 + b] * hb
            za[b] = za[b] + s2 * z[1 + a] * hb
            out[s] = s1 * hb
        out[0] = z0
        for a in range(self.d):
            out[1 + a] = za[a]
        return out

    def forward_states(self, x):
        """x: [d][16] coordinates of the tile.  Returns list of layer states."""
        NB, NS, d = self.NB, self.NS, self.d
        xs = x[:, P_]                                  # per lane
        states = []
        z = np.zeros((NS, NB, 64, 4)); z0 = np.zeros((NB, 64, 4))
        for b in range(NB):
            j = self.frag_units(b)
            z0[b] = self.b1[j] + sum(self.W1[j, a] * xs[a][:, None] for a in range(d))
            for a in range(d):
                z[1 + a, b] = self.W1[j, a]
        states.append(dict(d=act_derivs(self.act, z0), z=z))
        for l in range(2, self.L + 1):
            h = self.act_forward(states[-1])
            zz = self.gemm_frag(self.Wf[l], h)
            for b in range(NB):
                zz[0, b] += self.b[l][self.frag_units(b)]
            z0 = zz[0].copy()
            states.append(dict(d=act_derivs(self.act, z0), z=zz))
        return states

    def forward(self, x):
        st = self.forward_states(x)
        h = self.act_forward(st[-1])
        out = np.zeros((self.NS, 64))
        for b in range(self.NB):
            wo = self.Wout[self.frag_units(b)]
            for s in range(self.NS):
                out[s] += (wo * h[s, b]).sum(axis=1)
        # quad_sum over q
        res = np.zeros((self.NS, 16))
        for s in range(self.NS):
            for p in range(16):
                res[s, p] = out[s][P_ == p].sum()
        res[0] += self.bout
        return res

    def weight_grad(self, zb, h):
        NB, H = self.NB, self.H
        HP = H + 4
        acc = np.zeros((NB, NB, 64, 4))
        for s in range(self.NS):
            Zt = np.zeros(16 * HP); Ht = np.zeros(16 * HP)
            for b in range(NB):
                for r in range(4):
                    Zt[P_ * HP + 16 * b + 4 * Q_ + r] = zb[s, b][:, r]
                    Ht[P_ * HP + 16 * b + 4 * Q_ + r] = h[s, b][:, r]
            for st in range(4):
                av = [Zt[(4 * Q_ + st) * HP + 16 * b + P_] for b in range(NB)]
                bv = [Ht[(4 * Q_ + st) * HP + 16 * b + P_] for b in range(NB)]
                for jb in range(NB):
                    for kb in range(NB):
                        acc[jb, kb] = mfma16x16x4(av[jb], bv[kb], acc[jb, kb])
        return acc

    def backward(self, x, gout):
        """x [d][16], gout [NS][16] -> flat gradient of sum_s,p gout*jets (one tile)."""
        NB, NS, d, H, L = self.NB, self.NS, self.d, self.H, self.L
        st = self.forward_states(x)
        h = self.act_forward(st[-1])
        go = gout[:, P_]                               # [NS][64]
        grad = np.zeros(self.P)
        offW1, offb1 = 0, H * d
        offW = lambda l: H * d + H + (l - 2) * (H * H + H)
        offb = lambda l: offW(l) + H * H
        offWout = H * d + H + (L - 1) * (H * H + H)
        offbout = offWout + H
        g = np.zeros((NS, NB, 64, 4))
        for b in range(NB):
            j = self.frag_units(b)
            wo = self.Wout[j]
            dw = np.zeros((64, 4))
            for s in range(NS):
                g[s, b] = wo * go[s][:, None]
                dw += go[s][:, None] * h[s, b]
            for q in range(4):                         # point_sum, lanes with p == 0 write
                for r in range(4):
                    grad[offWout + 16 * b + 4 * q + r] += dw[Q_ == q, r].sum()
        grad[offbout] += go[0][Q_ == 0].sum()
        for l in range(L, 1, -1):
            li = l - 1
            g = self.act_backward(st[li], g)
            for b in range(NB):
                for q in range(4):
                    for r in range(4):
                        grad[offb(l) + 16 * b + 4 * q + r] += g[0, b][Q_ == q, r].sum()
            h = self.act_forward(st[li - 1])
            acc = self.weight_grad(g, h)
            for jb in range(NB):
                for kb in range(NB):
                    for r in range(4):
                        grad[offW(l) + (16 * jb + 4 * Q_ + r) * H + 16 * kb + P_] += acc[jb, kb][:, r]
            g = self.gemm_frag(self.Wt[l], g)
        g = self.act_backward(st[0], g)
        xs = x[:, P_]
        for b in range(NB):
            for q in range(4):
                for r in range(4):
                    j = 16 * b + 4 * q + r
                    m = Q_ == q
                    grad[offb1 + j] += g[0, b][m, r].sum()
                    for a in range(d):
                        grad[offW1 + j * d + a] += (g[0, b][m, r] * xs[a][m] + g[1 + a, b][m, r]).sum()
        return grad
