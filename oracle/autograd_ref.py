"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's training hot path -- ``BaseSolver._run_epoch`` and everything it calls
per batch -- in terms of the same third-party primitive the reference uses (PyTorch CPU autograd):

* ``ref_diff``            <- neurodiffeq/neurodiffeq.py:7-34  (k reverse sweeps, ``create_graph=True``)
* ``make_fcnn``           <- neurodiffeq/networks.py:59-70    (Sequential(Linear, actv, ..., Linear))
* ``ivp / dirichlet_bvp2d / ibvp1d_dd / no_condition``
                          <- neurodiffeq/conditions.py:247-267, 501-509, 677-681, 212-222
* ``sample_1d / sample_2d`` <- neurodiffeq/generators.py:152-158, 253-266 (noisy equally-spaced grids)
* ``closure``             <- neurodiffeq/solvers.py:369-395   (funcs -> residuals -> cat -> mean(r^2) -> backward)
* ``TrainLoop.epoch``     <- neurodiffeq/solvers.py:343-424   (zero_grad, accumulate over batches, Adam step)

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against the fixtures in
``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` produced by running the unmodified reference
(generator draws bit-exact; funcs/residuals/loss/grad in fp64 and fp32; a 3-epoch Adam trajectory).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
It is what the HIP path is checked against and what is timed as the CPU baseline (kind = "port": same ATen
CPU kernels and the same number of autograd graph walks as the reference, which cannot travel to the GPU box).
"""
import math
from itertools import chain

import torch
from torch import nn

PI = math.pi


# ------------------------------------------------------------------------------------------- diff
def ref_diff(u, t, order=1):
    """Per-sample derivative d^order u / dt^order by repeated reverse sweeps (sum trick).

    Follows neurodiffeq.py:21-34: ``grad_outputs=ones``, ``create_graph=True``, ``allow_unused=True``;
    an unused ``t`` yields zeros (not an error).  Shapes must both be (N, 1) (neurodiffeq.py:52-59)."""
    if u.dim() != 2 or t.dim() != 2 or u.shape[1] != 1 or t.shape[1] != 1 or u.shape != t.shape:
        raise ValueError(f"diff needs (N,1) tensors of equal shape; got {tuple(u.shape)}, {tuple(t.shape)}")
    cur = u
    for _ in range(order):
        (cur,) = torch.autograd.grad(cur, t, grad_outputs=torch.ones_like(cur), create_graph=True,
                                     allow_unused=True)
        if cur is None:
            return torch.zeros_like(t, requires_grad=True)
        if not cur.requires_grad:      # constant derivative (e.g. d2(t^2)/dt2): the next sweep must yield
            cur.requires_grad_()       # "unused" -> zeros, not an autograd error (neurodiffeq.py:25-26,32-33)
    return cur


# ------------------------------------------------------------------------------------------- network
class Sin(nn.Module):
    def forward(self, z):
        return torch.sin(z)


class SwishRef(nn.Module):
    """x * sigmoid(beta x) with the default beta = 1 (networks.py:155-175)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


class APTxRef(nn.Module):
    """(alpha + tanh(beta x)) * gamma * x with the defaults alpha = 1, beta = 1, gamma = 0.5 (networks.py:177-209)."""

    def forward(self, x):
        return (1.0 + torch.tanh(x)) * 0.5 * x


class SwishTrainableRef(nn.Module):
    """x * sigmoid(beta x) with a trainable scalar beta (networks.py:166-169)."""

    def __init__(self):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x)


class APTxTrainableRef(nn.Module):
    """(alpha + tanh(beta x)) * gamma * x with trainable scalars, registered in that order (networks.py:196-203)."""

    def __init__(self):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor(1.0))
        self.beta = nn.Parameter(torch.tensor(1.0))
        self.gamma = nn.Parameter(torch.tensor(0.5))

    def forward(self, x):
        return (self.alpha + torch.tanh(self.beta * x)) * self.gamma * x


class MonomialRef(nn.Module):
    """x -> [x^d for d in degrees] along dim 1 (networks.py:109-139)."""

    def __init__(self, degrees):
        super().__init__()
        self.degrees = tuple(degrees)

    def forward(self, x):
        return torch.cat([x ** d for d in self.degrees], dim=1)


class SwishFixedRef(nn.Module):
    """x * sigmoid(beta x) with a fixed non-default beta = 1.7 (Swish(beta=1.7), networks.py:161-169)."""

    def forward(self, x):
        return x * torch.sigmoid(1.7 * x)


class APTxFixedRef(nn.Module):
    """(alpha + tanh(beta x)) * gamma * x with fixed alpha = 0.8, beta = 1.3, gamma = 0.6 (networks.py:193-209)."""

    def forward(self, x):
        return (0.8 + torch.tanh(1.3 * x)) * 0.6 * x


ACTIVATIONS = {"tanh": nn.Tanh, "sin": Sin, "sigmoid": nn.Sigmoid, "swish": SwishRef, "aptx": APTxRef,
               "swish-tr": SwishTrainableRef, "aptx-tr": APTxTrainableRef, "swish-fixed": SwishFixedRef,
               "aptx-fixed": APTxFixedRef, "elu": nn.ELU, "softplus": nn.Softplus, "gelu": nn.GELU}


def make_fcnn(n_in, n_out, hidden, act="tanh", dtype=torch.float32):
    """Linear -> act -> ... -> Linear, PyTorch default init (networks.py:59-66)."""
    dims = (n_in,) + tuple(hidden)
    mods = []
    for a, b in zip(dims[:-1], dims[1:]):
        mods += [nn.Linear(a, b), ACTIVATIONS[act]()]
    mods.append(nn.Linear(dims[-1], n_out))
    return nn.Sequential(*mods).to(dtype)


class ResnetRef(nn.Module):
    """FCNN branch + trainable bias-free linear skip from input to output (networks.py:73-106); the branch is built
    first, so ``parameters()`` lists its weights before the skip matrix."""

    def __init__(self, n_in, n_out, hidden, act="tanh", dtype=torch.float32):
        super().__init__()
        self.residual = make_fcnn(n_in, n_out, hidden, act, dtype)
        self.skip_connection = nn.Linear(n_in, n_out, bias=False).to(dtype)

    def forward(self, t):
        return self.skip_connection(t) + self.residual(t)


def get_flat(nets):
    return torch.cat([p.detach().reshape(-1) for p in chain.from_iterable(n.parameters() for n in nets)])


def set_flat(nets, flat):
    """Load a flat parameter vector in torch's parameter order (W1,b1,W2,b2,... per net, nets in order)."""
    flat = torch.as_tensor(flat)
    off = 0
    with torch.no_grad():
        for p in chain.from_iterable(n.parameters() for n in nets):
            k = p.numel()
            p.copy_(flat[off:off + k].reshape(p.shape).to(p.dtype))
            off += k
    assert off == flat.numel()


def get_flat_grad(nets):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in chain.from_iterable(n.parameters() for n in nets)])


# ------------------------------------------------------------------------------------------- conditions
# Each factory returns ``enforce(net, *coords) -> (N,1)``: evaluate the net on cat(coords, 1) (conditions.py:52)
# and re-parameterise so the condition holds exactly.

def _apply(fn, s):
    """Boundary callables may return python scalars (``lambda y: 0``), as in the README examples."""
    return fn(s)


def no_condition():
    return lambda net, *c: net(torch.cat(c, dim=1))


def ivp(t0, u0):
    """u = u0 + (1 - exp(-(t - t0))) * N   (conditions.py:264-265, Dirichlet form)."""
    return lambda net, t: u0 + (1 - torch.exp(-t + t0)) * net(t)


def dirichlet_bvp2d(x0, f0, x1, f1, y0, g0, y1, g1):
    """conditions.py:501-509.  A(x,y) interpolates the four edge functions; B = xt(1-xt)yt(1-yt)."""
    def enforce(net, x, y):
        out = net(torch.cat([x, y], dim=1))
        xt, yt = (x - x0) / (x1 - x0), (y - y0) / (y1 - y0)
        cx0, cx1 = torch.full_like(x, x0), torch.full_like(x, x1)
        a = (1 - xt) * _apply(f0, y) + xt * _apply(f1, y) \
            + (1 - yt) * (_apply(g0, x) - ((1 - xt) * _apply(g0, cx0) + xt * _apply(g0, cx1))) \
            + yt * (_apply(g1, x) - ((1 - xt) * _apply(g1, cx0) + xt * _apply(g1, cx1)))
        return a + xt * (1 - xt) * yt * (1 - yt) * out
    return enforce


def ibvp1d_dd(x0, x1, t0, u_init, g_left, h_right):
    """Dirichlet-Dirichlet IBVP (conditions.py:669-681):
    u = u_init(x) + xt (h(t) - h(t0)) + (1 - xt)(g(t) - g(t0)) + xt (1 - xt)(1 - exp(-(t - t0))) N."""
    def enforce(net, x, t):
        out = net(torch.cat([x, t], dim=1))
        xt, tt = (x - x0) / (x1 - x0), t - t0
        ct0 = torch.full_like(t, t0)
        a = _apply(u_init, x) + xt * (_apply(h_right, t) - _apply(h_right, ct0)) \
            + (1 - xt) * (_apply(g_left, t) - _apply(g_left, ct0))
        return a + xt * (1 - xt) * (1 - torch.exp(-tt)) * out
    return enforce


def dirichlet_bvp_spherical_basis(r0, R0, r1, R1):
    """Coefficient-vector Dirichlet condition on a spherical shell (conditions.py:1089-1096):
    R = R0 (1 - rt) + R1 rt + (1 - exp((1 - rt) rt)) N(r),  rt = (r - r0)/(r1 - r0); the net sees r only."""
    def enforce(net, r):
        rt = (r - r0) / (r1 - r0)
        return R0 * (1 - rt) + R1 * rt + (1. - torch.exp((1 - rt) * rt)) * net(r)
    return enforce


def real_spherical_harmonics(theta, phi, max_degree=4):
    """(N,1),(N,1) -> (N,(max_degree+1)^2): real harmonics l <= 4 in the reference's normalisation
    (function_basis.py:200-229; textbook closed forms)."""
    s, c, sp, cp, c2p = torch.sin(theta), torch.cos(theta), torch.sin(phi), torch.cos(phi), torch.cos(2 * phi)
    bands = [
        [torch.ones_like(theta) * 0.5],
        [s * sp * 0.866025404, c * 0.866025404, s * cp * 0.866025404],
        [s ** 2 * sp * cp * 1.936491673, s * c * sp * 1.936491673, (2 * c ** 2 - s ** 2) * 0.559016994,
         s * c * cp * 1.936491673, s ** 2 * c2p * 0.968245837],
        [s ** 3 * (3 * cp ** 2 * sp - sp ** 3) * 1.045825033, s ** 2 * c * cp * sp * 5.123475383,
         s * (4 * c ** 2 - s ** 2) * sp * 0.810092587, (2 * c ** 3 - 3 * c * s ** 2) * 0.661437828,
         s * (4 * c ** 2 - s ** 2) * cp * 0.810092587, c * s ** 2 * c2p * 2.561737691,
         s ** 3 * (cp ** 3 - 3 * sp ** 2 * cp) * 1.045825033],
        [s ** 4 * (sp * cp * c2p) * 4.437059837, s ** 3 * c * (3 * cp ** 2 * sp - sp ** 3) * 3.1374751,
         s ** 2 * (sp * cp) * (7 * c ** 2 - 1) * 1.677050983, s * c * sp * (7 * c ** 2 - 3) * 1.185854123,
         (35 * c ** 4 - 30 * c ** 2 + 3) * 0.1875, s * c * cp * (7 * c ** 2 - 3) * 1.185854123,
         s ** 2 * c2p * (7 * c ** 2 - 1) * 0.838525492, s ** 3 * c * (cp ** 3 - 3 * cp * sp ** 2) * 3.1374751,
         s ** 4 * (cp ** 4 - 6 * cp ** 2 * sp ** 2 + sp ** 4) * 1.109264959],
    ]
    return torch.cat([y for band in bands[:max_degree + 1] for y in band], dim=1)


def spherical_laplacian(u, r, theta, phi):
    """operators.py:203-207 with ref_diff."""
    u_r, u_t, u_p = ref_diff(u, r), ref_diff(u, theta), ref_diff(u, phi)
    st = torch.sin(theta)
    r2 = r ** 2
    return (ref_diff(r2 * u_r, r) + ref_diff(st * u_t, theta) / st + ref_diff(u_p, phi) / st ** 2) / r2


# ------------------------------------------------------------------------------------------- generators
def sample_1d(n, t_min, t_max, dtype=torch.float32):
    """'equally-spaced-noisy' Generator1D (generators.py:139-158): normal(mean=linspace, std=(range/n)/4)."""
    grid = torch.linspace(t_min, t_max, n, dtype=dtype)
    std = ((t_max - t_min) / n) / 4.0
    return lambda: (torch.normal(mean=grid, std=std),)


def sample_2d(grid, xy_min, xy_max, dtype=torch.float32):
    """'equally-spaced-noisy' Generator2D (generators.py:253-266): meshgrid(ij).flatten + two normal draws
    (x first, then y) from the global CPU RNG."""
    gx = torch.linspace(xy_min[0], xy_max[0], grid[0], dtype=dtype)
    gy = torch.linspace(xy_min[1], xy_max[1], grid[1], dtype=dtype)
    mx, my = torch.meshgrid(gx, gy, indexing="ij")
    mx, my = mx.flatten(), my.flatten()
    sx = ((xy_max[0] - xy_min[0]) / grid[0]) / 4.0
    sy = ((xy_max[1] - xy_min[1]) / grid[1]) / 4.0
    return lambda: (torch.normal(mean=mx, std=sx), torch.normal(mean=my, std=sy))


def sample_spherical(n, r_min, r_max, dtype=torch.float32):
    """GeneratorSpherical 'equally-spaced-noisy' (generators.py:622-646): 3 rand + 3 randint + 1 rand, in that order."""
    def draw():
        a, b, c = torch.rand(n, dtype=dtype), torch.rand(n, dtype=dtype), torch.rand(n, dtype=dtype)
        denom = a + b + c
        x, y, z = torch.sqrt(a / denom) + 1e-6, torch.sqrt(b / denom) + 1e-6, torch.sqrt(c / denom) + 1e-6
        sx = torch.randint(0, 2, (n,), dtype=dtype) * 2 - 1
        sy = torch.randint(0, 2, (n,), dtype=dtype) * 2 - 1
        sz = torch.randint(0, 2, (n,), dtype=dtype) * 2 - 1
        x, y, z = x * sx, y * sy, z * sz
        theta = torch.acos(z)
        phi = -torch.atan2(y, x) + PI
        r = torch.sqrt((r_max ** 2 - r_min ** 2) * torch.rand(n, dtype=dtype) + r_min ** 2)
        return r, theta, phi
    return draw


# ------------------------------------------------------------------------------------------- closure / loop
LOSSES = {   # losses.py:4-12
    "l2": lambda r: (r ** 2).mean(),
    "l1": lambda r: torch.abs(r).mean(),
    "infinity": lambda r: r.abs().max(dim=1)[0].mean(),
}


def sobolev_loss(res, batch, semi=False):
    """losses.py:17-26: g_a = d(sum_e r_e)/dx_a by one reverse sweep each (operators.grad, operators.py:15-33 with
    grad_outputs = ones over the whole (N, n_eq) residual); h1 = mean([r, g]^2), h1 semi = mean(g^2)."""
    g = [torch.autograd.grad(res, x, grad_outputs=torch.ones_like(res), create_graph=True)[0] for x in batch]
    cols = g if semi else [res] + g
    return (torch.cat(cols, dim=1) ** 2).mean()


def closure(nets, enforcers, pde, coords, backward=True, loss="l2"):
    """One training closure (solvers.py:369-395) on given coordinates.

    ``coords``: list of 1-D (or (N,1)) tensors.  Returns dict(funcs (N,n_funcs), residuals (N,n_eq), loss 0-d)
    and leaves ``.grad`` accumulated on the parameters when ``backward``."""
    batch = [c.detach().reshape(-1, 1).requires_grad_(True) for c in coords]
    funcs = [e(n, *batch) for n, e in zip(nets, enforcers)]
    res = torch.cat(pde(*funcs, *batch), dim=1)
    loss = sobolev_loss(res, batch, semi=(loss == "h1 semi")) if loss in ("h1", "h1 semi") else LOSSES[loss](res)
    if backward:
        loss.backward()
    return dict(funcs=torch.cat(funcs, dim=1).detach(), residuals=res.detach(), loss=loss.detach())


def closure_chunked(nets, enforcers, pde, coords, chunk=65536, backward=True, keep=False):
    """``closure`` for batches too large to differentiate in one piece on the CPU (C5 at 1 048 576 points is ~43 GB
    of autograd graph).  The default loss (solvers.py:218) is a mean over points and the
    gradient a sum over points, so the batch is walked in chunks of ``chunk`` points, each contributing
    ``sum(r^2) / (N * n_eq)`` -- mathematically the closure above, evaluated piecewise; summation in fp64.
    Returns dict(loss, and -- with ``keep`` -- funcs / residuals of the whole batch)."""
    n = coords[0].numel()
    total, funcs, resid = 0.0, [], []
    n_eq = None
    for lo in range(0, n, chunk):
        batch = [c.detach().reshape(-1)[lo:lo + chunk].reshape(-1, 1).requires_grad_(True) for c in coords]
        f = [e(net, *batch) for net, e in zip(nets, enforcers)]
        res = torch.cat(pde(*f, *batch), dim=1)
        n_eq = res.shape[1]
        part = (res ** 2).sum() / (n * n_eq)
        if backward:
            part.backward()
        total += float(part.detach())
        if keep:
            funcs.append(torch.cat(f, dim=1).detach())
            resid.append(res.detach())
    out = dict(loss=torch.tensor(total, dtype=torch.float64))
    if keep:
        out.update(funcs=torch.cat(funcs), residuals=torch.cat(resid))
    return out


class TrainLoop:
    """The compute core of ``_run_epoch('train')`` (solvers.py:343-424) with the default optimiser
    ``Adam(lr=1e-3)`` (solvers.py:182): zero_grad once, accumulate over ``n_batches`` draws, one step."""

    def __init__(self, nets, enforcers, pde, sampler, n_batches=1, lr=1e-3):
        self.nets, self.enforcers, self.pde, self.sampler, self.n_batches = nets, enforcers, pde, sampler, n_batches
        self.opt = torch.optim.Adam(list(chain.from_iterable(n.parameters() for n in nets)), lr=lr)
        self.history = []

    def epoch(self):
        self.opt.zero_grad()
        tot = 0.0
        for _ in range(self.n_batches):
            out = closure(self.nets, self.enforcers, self.pde, self.sampler())
            tot += out["loss"].item()
        self.history.append(tot / self.n_batches)
        self.opt.step()
        return self.history[-1]


# ------------------------------------------------------------------------------------------- BASELINE configs
def lid_profile(x):
    return (1 - torch.exp(-50.0 * x)) * (1 - torch.exp(50.0 * (x - 1)))


def build_config(name, size=None, dtype=torch.float32):
    """The BASELINE.json configs expressed with the oracle's primitives.

    Returns dict(nets, enforcers, pde, sampler, n_points).  ``size``: points for c1, grid edge otherwise."""
    d = ref_diff
    zero = lambda s: 0
    if name == "c1":      # Lotka-Volterra, README.md:86-92
        n = size or 1024
        nets = [make_fcnn(1, 1, (32, 32), "sin", dtype) for _ in range(2)]
        enf = [ivp(0.0, 1.5), ivp(0.0, 1.0)]
        pde = lambda u, v, t: [d(u, t) - (u - u * v), d(v, t) - (u * v - v)]
        return dict(nets=nets, enforcers=enf, pde=pde, sampler=sample_1d(n, 0.1, 12.0, dtype), n_points=n)
    if name == "c2":      # Laplace, README.md:114-124
        g = size or 256
        nets = [make_fcnn(2, 1, (32, 32), "tanh", dtype)]
        enf = [dirichlet_bvp2d(0, lambda y: torch.sin(PI * y), 1, zero, 0, zero, 1, zero)]
        pde = lambda u, x, y: [d(u, x, 2) + d(u, y, 2)]
        return dict(nets=nets, enforcers=enf, pde=pde, sampler=sample_2d((g, g), (0, 0), (1, 1), dtype),
                    n_points=g * g)
    if name == "c3":      # viscous Burgers in (x, t)
        g = size or 512
        nu = 0.01 / PI
        nets = [make_fcnn(2, 1, (64, 64, 64), "tanh", dtype)]
        enf = [ibvp1d_dd(-1, 1, 0, lambda x: -torch.sin(PI * x), zero, zero)]
        pde = lambda u, x, t: [d(u, t) + u * d(u, x) - nu * d(u, x, 2)]
        return dict(nets=nets, enforcers=enf, pde=pde, sampler=sample_2d((g, g), (-1, 0), (1, 1), dtype),
                    n_points=g * g)
    if name == "c5":      # lid-driven cavity, experiments/lid-driven-cavity-RE400.ipynb cell 3
        g = size or 1024
        re = 400.0
        nets = [make_fcnn(2, 1, (64, 64, 64), "tanh", dtype) for _ in range(3)]
        enf = [dirichlet_bvp2d(0, zero, 1, zero, 0, zero, 1, lid_profile),
               dirichlet_bvp2d(0, zero, 1, zero, 0, zero, 1, zero), no_condition()]

        def pde(u, v, p, x, y):
            mx = u * d(u, x) + v * d(u, y) + d(p, x) - 1 / re * (d(u, x, 2) + d(u, y, 2))
            my = u * d(v, x) + v * d(v, y) + d(p, y) - 1 / re * (d(v, x, 2) + d(v, y, 2))
            return [mx, my, d(u, x) + d(v, y)]
        return dict(nets=nets, enforcers=enf, pde=pde, sampler=sample_2d((g, g), (0, 0), (1, 1), dtype),
                    n_points=g * g)
    if name == "c4":      # Poisson in a spherical shell, harmonic expansion (tests/test_pde_spherical.py:103-175 shape)
        n = size or 131072
        r0, r1 = 0.1, 3.0
        gauss = 1.0 / (2 * PI) ** 1.5
        kq = 1.0 / (4 * PI)
        v0 = kq / r0 * math.erf(r0 / math.sqrt(2.0))
        v1 = kq / r1 * math.erf(r1 / math.sqrt(2.0))
        # boundary coefficient rows are fp32 data (as a user script under set_tensor_type(float_bits=32) makes them)
        R0 = torch.zeros(25, dtype=torch.float32); R0[0] = 2 * v0; R0 = R0.to(dtype)
        R1 = torch.zeros(25, dtype=torch.float32); R1[0] = 2 * v1; R1 = R1.to(dtype)
        nets = [make_fcnn(1, 25, (32, 32), "tanh", dtype)]
        cond = dirichlet_bvp_spherical_basis(r0, R0, r1, R1)
        enf = [lambda net, r, th, ph: (cond(net, r) * real_spherical_harmonics(th, ph)).sum(dim=1, keepdim=True)]
        pde = lambda u, r, th, ph: [spherical_laplacian(u, r, th, ph) + gauss * torch.exp(-r ** 2 / 2)]
        return dict(nets=nets, enforcers=enf, pde=pde, sampler=sample_spherical(n, r0, r1, dtype), n_points=n)
    raise KeyError(name)
