"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the device sampler (neurodiffeq_amd/csrc/ndq_sample.h).

Philox4x32-10 is restated from its publication (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11; the Random123 library) and pinned to that library's known-answer vectors in
tests/test_sampler.py.  The transforms on top restate the reference's generators (neurodiffeq/generators.py):
'uniform' 150-152, noisy ij-meshgrid 253-266 (Generator2D; 1-D 158, 3-D 388-399), GeneratorSpherical 622-646.
Integer work (the Philox words, grid indices, sign bits) is bit-exact with the kernel; the float transforms agree to
a few ulp (libm vs the device's fast log/sin/cos)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (4, n) uint32, key: (k0, k1) python ints -> (4, n) uint32"""
    c = [np.asarray(x, dtype=np.uint64) for x in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & MASK, p1 & MASK,
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c).astype(np.uint32)


def words(n, seed, draw, stream_id):
    i = np.arange(n, dtype=np.uint64)
    ctr = [i, np.full(n, draw & 0xFFFFFFFF, np.uint64), np.full(n, (draw >> 32) & 0xFFFFFFFF, np.uint64),
           np.full(n, stream_id, np.uint64)]
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def u01(w):
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def u01_open(w):
    return ((w >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * np.float32(2.0 ** -24)


def linspace(lo, hi, n):
    """torch.linspace in fp32: stepped from the nearer end with ONE fused multiply-add per element (emulated through
    fp64: the product of two fp32 numbers is exact there)"""
    if n <= 1:
        return np.full(n, lo, np.float32)
    lo, hi = np.float32(lo), np.float32(hi)
    step = np.float32((hi - lo) / np.float32(n - 1))
    i = np.arange(n)
    lo64, hi64, step64 = np.float64(lo), np.float64(hi), np.float64(step)
    return np.where(i < n // 2, lo64 + step64 * i, hi64 - step64 * (n - 1 - i)).astype(np.float32)


def sample_uniform(n, lo, hi, seed, draw, stream_id=0):
    w = words(n, seed, draw, stream_id)
    return np.stack([np.float32(l) + (np.float32(h) - np.float32(l)) * u01(w[c]) for c, (l, h) in enumerate(zip(lo, hi))])


def sample_grid(grid, lo, hi, std, seed, draw, stream_id=0):
    n = int(np.prod(grid))
    w = words(n, seed, draw, stream_id)
    two_pi = np.float32(6.283185307179586)
    r0, t0 = np.sqrt(np.float32(-2.0) * np.log(u01_open(w[0]))), two_pi * u01(w[1])
    r1, t1 = np.sqrt(np.float32(-2.0) * np.log(u01_open(w[2]))), two_pi * u01(w[3])
    z = [r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1)]
    idx = np.unravel_index(np.arange(n), grid)                     # ij order, last axis fastest
    out = []
    for c in range(len(grid)):
        v = linspace(lo[c], hi[c], grid[c])[idx[c]]
        if std[c] != 0:
            v = v + np.float32(std[c]) * z[c].astype(np.float32)
        out.append(v.astype(np.float32))
    return np.stack(out)


def sample_spherical(n, r_min, r_max, radial, seed, draw, stream_id=0):
    w = words(n, seed, draw, stream_id)
    p, q, t = u01_open(w[0]), u01_open(w[1]), u01_open(w[2])
    inv = np.float32(1.0) / (p + q + t)
    eps = np.float32(1e-6)
    x, y = np.sqrt(p * inv) + eps, np.sqrt(q * inv) + eps
    z = np.minimum(np.sqrt(t * inv) + eps, np.float32(1.0))
    x = np.where(w[0] & 1, -x, x); y = np.where(w[1] & 1, -y, y); z = np.where(w[2] & 1, -z, z)
    u = u01(w[3])
    lo, hi = np.float32(r_min), np.float32(r_max)
    rad = lo + (hi - lo) * u if radial else np.sqrt((hi * hi - lo * lo) * u + lo * lo)
    return np.stack([rad, np.arccos(z), np.float32(np.pi) - np.arctan2(y, x)]).astype(np.float32)
