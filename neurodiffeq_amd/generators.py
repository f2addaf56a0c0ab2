"""Collocation-point generators with the reference's names, arguments and draw order
(neurodiffeq/generators.py:107-1064).

Generators are the *input producer* of the hot path, not part of it: the classes with the reference's names always
sample on the host with torch's global CPU generator, in the same call sequence as the reference, so that a given
``torch.manual_seed`` yields bit-identical points (the north-star parity contract); the solver uploads each batch to
HBM as one SoA block.  Two additions without a reference counterpart keep the producer off the step's critical path:
:class:`ResidentBatchGenerator` (pre-sampled batches resident in HBM) and :class:`DeviceGenerator` (the same
distributions drawn by a Philox kernel on the MI355X, a fresh batch per epoch with no PCIe hand-off)."""
import ctypes
import os
import warnings
import weakref

import numpy as np
import torch

_CPU = torch.device("cpu")
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _chebyshev_first(a, b, n):
    nodes = torch.cos(((torch.arange(n, device=_CPU) + 0.5) / n) * np.pi)
    return (((a + b) + (b - a) * nodes) / 2).requires_grad_(True)


def _chebyshev_second(a, b, n):
    nodes = torch.cos(torch.arange(n, device=_CPU) / float(n - 1) * np.pi)
    return (((a + b) + (b - a) * nodes) / 2).requires_grad_(True)


def _chebyshev_second_noisy(a, b, n):
    nodes = torch.cos((torch.arange(n, device=_CPU) + (torch.rand(n, device=_CPU) * 2 - 1)) / float(n - 1) * np.pi)
    return (((a + b) + (b - a) * nodes) / 2).requires_grad_(True)


def _latin_hypercube(a, b, n):
    edges = torch.linspace(a, b, steps=n + 1, device=_CPU)
    pts = torch.rand(n, device=_CPU) * (edges[1] - edges[0])
    pts += edges[:-1]
    pts = pts[torch.randperm(n, device=_CPU)]
    return pts.requires_grad_(True)


def _log_bounds(t_min, t_max, whence):
    if t_min <= 0 or t_max <= 0:
        raise ValueError(f"the interval [{t_min}, {t_max}] cannot be used for log-sampling in {whence}; "
                         f"if you meant [10^{t_min}, 10^{t_max}], pass {10 ** t_min} and {10 ** t_max}")
    return np.log10(t_min), np.log10(t_max)


def _rows_inplace(z, scale, shift):
    """``z * scale + shift`` in place (two roundings, like ``mul_`` then ``add_``) with broadcasting, on numpy views: the
    same IEEE operations as torch's, minus torch's intra-op thread pool -- a block of a few hundred batches is above
    its parallel grain size, and waking that pool costs more than the arithmetic (measured 55 ms against 0.2 ms)."""
    zn = z.numpy()
    if scale is not None:
        np.multiply(zn, scale.numpy(), out=zn)
    np.add(zn, shift.numpy(), out=zn)
    return z


#: attributes a generator of the reference reads WHEN IT DRAWS (generators.py:107-416: the getters are lambdas over ``self``), i.e.
#: what a callback can change between epochs with effect -- the grid tensors, the noise widths, the getter itself.  (t_min / t_max /
#: grid are consumed by __init__: changing them later has no effect in the reference either.)
_LIVE = ("getter", "get_r", "examples", "grid_x", "grid_y", "grid_z", "noise_std", "noise_xstd", "noise_ystd", "size", "shape", "method")


def live_stamp(gen, names=_LIVE):
    """What ``gen`` would draw from right now, cheaply comparable: (attribute, value) for numbers, (attribute, id, version) for
    tensors / callables.  A shortcut that was derived from the generator's state (one-call bulk draws, a device-side sampler's
    descriptor) is only valid while this stamp is what it was."""
    out = [type(gen).get_examples]
    d = gen.__dict__
    for name in names:
        v = d.get(name)
        if isinstance(v, torch.Tensor):
            out.append((id(v), v._version))
        elif isinstance(v, (list, tuple)):
            out.append(tuple(v))
        else:
            out.append(v)
    return out


def _bulk_validate(gen):
    """First use, or a callback has changed what the generator draws from (noise width, grid tensors, the getter): whatever was
    cached or proven for the old state is void."""
    stamp = live_stamp(gen)
    if gen.__dict__.get("_bulk_stamp") != stamp:
        gen.__dict__.pop("_bulk_rows", None)
        gen._bulk_ok = None
        gen._bulk_stamp = stamp


def _checked_bulk(gen, k, draw):
    """``draw(k)`` -> host tensor [k][d][size] holding what ``k`` consecutive ``gen.get_examples()`` calls would return,
    drawn with ONE call into torch's CPU generator (the multi-epoch fit path of the solvers draws the batches of a whole
    chunk of epochs at once).  torch's ``normal_`` fills blocks of 16 values from 16 uniform draws, so one long call
    reproduces a sequence of short ones exactly when every short one is a whole number of blocks -- but that is a
    property of torch's kernels, not of its API: the first use on a generator proves it (two batches drawn both ways
    from the same RNG state, which is restored) and the generator falls back to one call per batch if it does not hold."""
    _bulk_validate(gen)
    ok = gen.__dict__.get("_bulk_ok")
    if ok is None:
        state = torch.get_rng_state()
        try:
            two = draw(2)
            torch.set_rng_state(state)
            seq = []
            for _ in range(2):
                ex = gen.get_examples()
                ex = [ex] if isinstance(ex, torch.Tensor) else list(ex)
                seq.append(torch.stack([e.detach().reshape(-1) for e in ex]))
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            draw(2)
            ok = bool(torch.equal(two, torch.stack(seq)) and torch.equal(after, torch.get_rng_state()))
        except Exception:       # noqa: BLE001 -- whatever went wrong, the one-call-per-batch path is always right
            ok = False
        torch.set_rng_state(state)
        gen._bulk_ok = ok
    return draw(k) if ok else None


class BaseGenerator:
    def __init__(self):
        self.size = None

    def get_examples(self):
        pass  # pragma: no cover

    def bulk_examples(self, k):
        """``k`` consecutive draws as one host tensor [k][d][size], bit-identical to ``k`` calls of ``get_examples`` and
        leaving torch's RNG in the same state -- or None when this generator has no such shortcut (callers then simply
        call ``get_examples`` ``k`` times)."""
        return None

    @staticmethod
    def check_generator(obj):
        if not isinstance(obj, BaseGenerator):
            raise ValueError(f"{obj} is not a generator")

    def __add__(self, other):
        self.check_generator(other)
        return ConcatGenerator(self, other)

    def __mul__(self, other):
        self.check_generator(other)
        return EnsembleGenerator(self, other)

    def __xor__(self, other):
        self.check_generator(other)
        return MeshGenerator(self, other)

    def _internal_vars(self):
        return dict(size=self.size)

    def __repr__(self):
        d = self._internal_vars()
        return f"{self.__class__.__name__}({', '.join(f'{k}={d[k]!r}' for k in d)})"


class Generator1D(BaseGenerator):
    """1-D points on [t_min, t_max] (generators.py:107-191); ``method`` as in the reference."""

    def __init__(self, size, t_min=0.0, t_max=1.0, method="uniform", noise_std=None, bit_exact_cpu=False):
        super().__init__()
        self.bit_exact_cpu = bool(bit_exact_cpu)      # True: never drawn on the device (see on_default_device)
        self.size, self.t_min, self.t_max, self.method = size, t_min, t_max, method
        self.noise_std = noise_std if noise_std else ((t_max - t_min) / size) / 4.0
        fixed = None
        if method == "uniform":
            self.examples = torch.zeros(size, requires_grad=True, device=_CPU)
            self.getter = lambda: self.examples + torch.rand(size, device=_CPU) * (t_max - t_min) + t_min
        elif method in ("equally-spaced", "equally-spaced-noisy"):
            fixed = torch.linspace(t_min, t_max, size, requires_grad=True, device=_CPU)
        elif method in ("log-spaced", "log-spaced-noisy"):
            lo, hi = _log_bounds(t_min, t_max, self.__class__)
            fixed = torch.logspace(lo, hi, size, requires_grad=True, device=_CPU)
        elif method in ("chebyshev", "chebyshev1"):
            fixed = _chebyshev_first(t_min, t_max, size)
        elif method == "chebyshev2":
            fixed = _chebyshev_second(t_min, t_max, size)
        elif method == "chebyshev2-noisy":
            self.getter = lambda: _chebyshev_second_noisy(t_min, t_max, size)
        elif method == "latin-hypercube":
            self.getter = lambda: _latin_hypercube(t_min, t_max, size)
        else:
            raise ValueError(f"Unknown method: {method}")
        if fixed is not None:
            self.examples = fixed
            if method.endswith("-noisy"):
                self.getter = lambda: torch.normal(mean=self.examples, std=self.noise_std)
            else:
                self.getter = lambda: self.examples

    def get_examples(self):
        return self.getter()

    def bulk_examples(self, k):
        # torch.normal(mean=m, std=s) is normal_(0, s) followed by add_(m): the same two steps over k batches at once
        if self.method in ("equally-spaced-noisy", "log-spaced-noisy") and self.size % 16 == 0:
            mean = self.examples.detach()
            return _checked_bulk(self, k, lambda kk: _rows_inplace(
                torch.empty(kk, 1, self.size, dtype=mean.dtype, device=_CPU).normal_(0, self.noise_std), None, mean))
        return None

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(t_min=self.t_min, t_max=self.t_max, method=self.method, noise_std=self.noise_std)
        return d


class Generator2D(BaseGenerator):
    """Points on a (possibly noisy) m x n grid over [x0,x1] x [y0,y1] (generators.py:194-314).  For the noisy grid
    the x noise is drawn before the y noise, each with one ``torch.normal`` over the flattened ij-meshgrid."""

    def __init__(self, grid=(10, 10), xy_min=(0.0, 0.0), xy_max=(1.0, 1.0), method="equally-spaced-noisy",
                 xy_noise_std=None, bit_exact_cpu=False):
        super().__init__()
        self.bit_exact_cpu = bool(bit_exact_cpu)
        self.grid, self.size = grid, grid[0] * grid[1]
        self.xy_min, self.xy_max, self.method, self.xy_noise_std = xy_min, xy_max, method, xy_noise_std
        axes = None
        if method in ("equally-spaced", "equally-spaced-noisy"):
            axes = [torch.linspace(xy_min[i], xy_max[i], grid[i], requires_grad=True, device=_CPU) for i in (0, 1)]
        elif method in ("chebyshev", "chebyshev1"):
            axes = [_chebyshev_first(xy_min[i], xy_max[i], grid[i]) for i in (0, 1)]
        elif method == "chebyshev2":
            axes = [_chebyshev_second(xy_min[i], xy_max[i], grid[i]) for i in (0, 1)]
        elif method == "latin-hypercube":
            axes = [_latin_hypercube(xy_min[i], xy_max[i], grid[i]) for i in (0, 1)]
        elif method == "chebyshev2-noisy":
            def draw():
                ax = [_chebyshev_second_noisy(xy_min[i], xy_max[i], grid[i]) for i in (0, 1)]
                gx, gy = torch.meshgrid(ax[0], ax[1], indexing="ij")
                return gx.flatten(), gy.flatten()
            self.getter = draw
        else:
            raise ValueError(f"Unknown method: {method}")
        if axes is not None:
            gx, gy = torch.meshgrid(axes[0], axes[1], indexing="ij")
            self.grid_x, self.grid_y = gx.flatten(), gy.flatten()
            if method == "equally-spaced-noisy":
                if xy_noise_std:
                    self.noise_xstd, self.noise_ystd = xy_noise_std
                else:
                    self.noise_xstd = ((xy_max[0] - xy_min[0]) / grid[0]) / 4.0
                    self.noise_ystd = ((xy_max[1] - xy_min[1]) / grid[1]) / 4.0
                self.getter = lambda: (torch.normal(mean=self.grid_x, std=self.noise_xstd),
                                       torch.normal(mean=self.grid_y, std=self.noise_ystd))
            else:
                self.getter = lambda: (self.grid_x, self.grid_y)

    def get_examples(self):
        return self.getter()

    def bulk_examples(self, k):
        # per batch the x noise is drawn before the y noise: one normal_ over [k][x | y][size] with per-element std
        # (torch.normal(mean=m, std=s) = normal_(0, s) + add_(m), and normal_(0, s) is the standard normal draw times s in
        # one rounding: one standard normal_ over [k][x | y][size], then the scale and the shift per row)
        if self.method == "equally-spaced-noisy" and self.size % 16 == 0:
            # (built from the generator's CURRENT grid tensors and noise widths on every call -- once per chunk of epochs: a copy
            # kept across calls would miss `g.grid_y.data.mul_(2.0)`, which no version counter sees)
            mean = torch.stack([self.grid_x.detach(), self.grid_y.detach()]).unsqueeze(0)
            std = torch.tensor([self.noise_xstd, self.noise_ystd], dtype=mean.dtype, device=_CPU).view(1, 2, 1)
            return _checked_bulk(self, k, lambda kk: _rows_inplace(
                torch.empty(kk, 2, self.size, dtype=mean.dtype, device=_CPU).normal_(), std, mean))
        return None

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(grid=self.grid, xy_min=self.xy_min, xy_max=self.xy_max, method=self.method,
                 xy_noise_std=self.xy_noise_std)
        return d


class GeneratorSpherical(BaseGenerator):
    """Points (r, theta, phi) with directions uniform-ish on the sphere (generators.py:572-655); same draw order as the
    reference: three ``rand`` for the direction, three ``randint`` signs, then the radius."""

    def __init__(self, size, r_min=0., r_max=1., method="equally-spaced-noisy", bit_exact_cpu=False):
        super().__init__()
        self.bit_exact_cpu = bool(bit_exact_cpu)
        if r_min < 0 or r_max < r_min:
            raise ValueError(f"Illegal range [{r_min}, {r_max}]")
        if method == "equally-spaced-noisy":        # r^2 ~ U[r_min^2, r_max^2]
            lo, span = r_min ** 2, r_max ** 2 - r_min ** 2
            self.get_r = lambda: torch.sqrt(span * torch.rand(self.shape, device=_CPU) + lo)
        elif method == "equally-radius-noisy":      # r ~ U[r_min, r_max]
            lo, span = r_min, r_max - r_min
            self.get_r = lambda: span * torch.rand(self.shape, device=_CPU) + lo
        else:
            raise ValueError(f"Unknown method: {method}")
        self.size, self.r_min, self.r_max, self.method = size, r_min, r_max, method
        self.shape = (size,)

    def get_examples(self):
        a, b, c = (torch.rand(self.shape, device=_CPU) for _ in range(3))
        denom = a + b + c
        eps = 1e-6
        x, y, z = (torch.sqrt(v / denom) + eps for v in (a, b, c))
        sx, sy, sz = (torch.randint(0, 2, self.shape, dtype=x.dtype, device=_CPU) * 2 - 1 for _ in range(3))
        x, y, z = x * sx, y * sy, z * sz
        theta = torch.acos(z).requires_grad_(True)
        phi = (-torch.atan2(y, x) + np.pi).requires_grad_(True)   # atan2 ranges (-pi, pi]; shift to [0, 2pi)
        r = self.get_r().requires_grad_(True)
        return r, theta, phi

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(r_min=self.r_min, r_max=self.r_max, method=self.method)
        return d


class Generator3D(BaseGenerator):
    """Points on a (noisy) 3-D grid (generators.py:317-416)."""

    def __init__(self, grid=(10, 10, 10), xyz_min=(0.0, 0.0, 0.0), xyz_max=(1.0, 1.0, 1.0),
                 method="equally-spaced-noisy", bit_exact_cpu=False):
        super().__init__()
        self.bit_exact_cpu = bool(bit_exact_cpu)
        self.grid, self.size = grid, grid[0] * grid[1] * grid[2]
        self.xyz_min, self.xyz_max, self.method = xyz_min, xyz_max, method
        if method in ("equally-spaced", "equally-spaced-noisy"):
            axes = [torch.linspace(xyz_min[i], xyz_max[i], grid[i], requires_grad=True, device=_CPU) for i in range(3)]
        elif method in ("chebyshev", "chebyshev1"):
            axes = [_chebyshev_first(xyz_min[i], xyz_max[i], grid[i]) for i in range(3)]
        elif method == "chebyshev2":
            axes = [_chebyshev_second(xyz_min[i], xyz_max[i], grid[i]) for i in range(3)]
        else:
            raise ValueError(f"Unknown method: {method}")
        gx, gy, gz = torch.meshgrid(*axes, indexing="ij")
        self.grid_x, self.grid_y, self.grid_z = gx.flatten(), gy.flatten(), gz.flatten()
        if method == "equally-spaced-noisy":
            self.noise_std = [((xyz_max[i] - xyz_min[i]) / grid[i]) / 4.0 for i in range(3)]
            self.getter = lambda: tuple(torch.normal(mean=m, std=s) for m, s in
                                        zip((self.grid_x, self.grid_y, self.grid_z), self.noise_std))
        else:
            self.getter = lambda: (self.grid_x, self.grid_y, self.grid_z)

    def get_examples(self):
        return self.getter()


class ConcatGenerator(BaseGenerator):
    """``g1 + g2``: concatenated samples (generators.py:658-691)."""

    def __init__(self, *generators):
        super().__init__()
        self.generators = generators
        self.size = sum(g.size for g in generators)

    def get_examples(self):
        draws = [g.get_examples() for g in self.generators]
        if isinstance(draws[0], torch.Tensor):
            return torch.cat(draws)
        return tuple(torch.cat(parts) for parts in zip(*draws))


class EnsembleGenerator(BaseGenerator):
    """``g1 * g2``: same number of points, more dimensions (generators.py:826-876)."""

    def __init__(self, *generators):
        super().__init__()
        self.size = generators[0].size
        for i, g in enumerate(generators):
            if g.size != self.size:
                raise ValueError(f"gens[{i}].size ({g.size}) != gens[0].size ({self.size})")
        self.generators = generators

    def get_examples(self):
        out = ()
        for g in self.generators:
            ex = g.get_examples()
            out += (ex,) if isinstance(ex, torch.Tensor) else tuple(ex)
        return out[0] if len(out) == 1 else out


class MeshGenerator(BaseGenerator):
    """``g1 ^ g2``: ij-meshgrid of the sub-generators' samples, flattened (generators.py:848-901); nested meshes are
    flattened into one (``(g1 ^ g2) ^ g3 == MeshGenerator(g1, g2, g3)``)."""

    def __init__(self, *generators):
        super().__init__()
        self.generators = []
        for g in generators:
            self.generators += list(g.generators) if isinstance(g, MeshGenerator) else [g]
        self.size = int(np.prod([g.size for g in self.generators]))

    def get_examples(self):
        parts = ()
        for g in self.generators:
            ex = g.get_examples()
            parts += (ex,) if isinstance(ex, torch.Tensor) else tuple(ex)
        if len(parts) == 1:
            return parts[0]
        return tuple(m.flatten() for m in torch.meshgrid(*parts, indexing="ij"))

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generators=self.generators)
        return d


class StaticGenerator(BaseGenerator):
    """Draw once, return the same points forever (generators.py:694-720)."""

    def __init__(self, generator):
        super().__init__()
        self.generator, self.size = generator, generator.size
        self.examples = generator.get_examples()

    def get_examples(self):
        return self.examples


class PredefinedGenerator(BaseGenerator):
    """Fixed user-supplied points (generators.py:723-756)."""

    def __init__(self, *xs):
        super().__init__()
        self.xs = [x if isinstance(x, torch.Tensor) else torch.tensor(x, device=_CPU) for x in xs]
        self.xs = [x.detach().clone().reshape(-1).requires_grad_(True) for x in self.xs]
        self.size = len(self.xs[0])
        if any(len(x) != self.size for x in self.xs):
            raise ValueError("tensors of different lengths encountered")
        if len(self.xs) == 1:
            self.xs = self.xs[0]

    def get_examples(self):
        return self.xs


class GeneratorND(BaseGenerator):
    """Points on an N-dimensional (noisy) grid with a sampling method per axis (generators.py:419-569):
    'equally-spaced', 'uniform', 'log-spaced', 'exp-spaced' (keyword ``base``), 'chebyshev' / 'chebyshev1', 'chebyshev2';
    keyword ``cut`` slices each axis, ``abs_value`` folds the jittered samples to non-negative values.  The jitter is
    one ``torch.normal`` per axis over the flattened ij-meshgrid, in axis order."""

    def __init__(self, grid=(10, 10), r_min=(0.0, 0.0), r_max=(1.0, 1.0), methods=("equally-spaced", "equally-spaced"),
                 noisy=True, r_noise_std=None, **kwargs):
        super().__init__()
        self.size = int(np.prod(grid))
        self.grid, self.r_min, self.r_max = grid, r_min, r_max
        self.methods, self.noisy, self.r_noise_std = methods, noisy, r_noise_std
        tup = lambda v: (v,) if isinstance(v, (int, float)) else v
        methods = [methods] if isinstance(methods, str) else methods
        grid, r_min, r_max = tup(grid), tup(r_min), tup(r_max)
        r_noise_std = tup(r_noise_std) if r_noise_std is not None else None
        n_dim = len(grid)
        cut = kwargs.pop("cut", tuple((None, None) for _ in range(n_dim)))
        base = tup(kwargs.pop("base", tuple(10 for _ in range(n_dim))))
        abs_value = kwargs.pop("abs_value", False)
        if kwargs:
            raise ValueError(f"Unknown keyword argument(s): {list(kwargs.keys())}")
        if cut[0] is None or isinstance(cut[0], (int, float)):
            cut = (cut,)
        axes, stds = [], []
        for i in range(n_dim):
            lo, hi, n, method = r_min[i], r_max[i], grid[i], methods[i]
            std = r_noise_std[i] if r_noise_std else ((hi - lo) / n) / 4.0
            if method == "equally-spaced":
                x = torch.linspace(lo, hi, n, requires_grad=True, device=_CPU)
                sd = std * torch.ones(n)
            elif method == "uniform":
                x = torch.zeros(n, requires_grad=True, device=_CPU) + torch.rand(n, device=_CPU) * (hi - lo) + lo
                sd = torch.zeros(n)
            elif method == "log-spaced":
                a, b = np.log10(lo), np.log10(hi)
                x = torch.logspace(a, b, n, requires_grad=True, device=_CPU)
                sd = std * torch.logspace(a, b, n, device=_CPU)
            elif method == "exp-spaced":
                x = torch.linspace(base[i] ** lo, base[i] ** hi, n, device=_CPU)
                x = (torch.log(x) / np.log(base[i])).clone().detach().requires_grad_(True)
                sd = (std * x).clone().detach()
            elif method in ("chebyshev", "chebyshev1"):
                x, sd = _chebyshev_first(lo, hi, n), std * torch.ones(n)
            elif method == "chebyshev2":
                x, sd = _chebyshev_second(lo, hi, n), std * torch.ones(n)
            else:
                raise ValueError(f"Unknown method: {method}")
            axes.append(x[cut[i][0]:cut[i][1]])
            stds.append(sd[cut[i][0]:cut[i][1]])
        self.grid_r = [m.flatten() for m in torch.meshgrid(*axes, indexing="ij")]
        self.grid_std = [m.flatten() for m in torch.meshgrid(*stds, indexing="ij")]
        if not noisy:
            self.getter = lambda: tuple(self.grid_r)
        elif abs_value:
            self.getter = lambda: tuple(torch.abs(torch.normal(m, s)) for m, s in zip(self.grid_r, self.grid_std))
        else:
            self.getter = lambda: tuple(torch.normal(m, s) for m, s in zip(self.grid_r, self.grid_std))

    def get_examples(self):
        return self.getter()

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(grid=self.grid, r_min=self.r_min, r_max=self.r_max, methods=self.methods, noisy=self.noisy,
                 r_noise_std=self.r_noise_std)
        return d


class TransformGenerator(BaseGenerator):
    """Applies ``transforms[i]`` to the i-th sample vector (``None`` = identity), or one ``transform`` to all of them
    at once (generators.py:752-801)."""

    def __init__(self, generator, transforms=None, transform=None):
        super().__init__()
        self.generator, self.size = generator, generator.size
        if transforms is not None and transform is not None:
            raise ValueError("transform and transforms cannot be both specified")
        if transforms is not None:
            self.trans = [(lambda x: x) if t is None else t for t in transforms]
        else:
            self.trans = transform if transform is not None else (lambda x: x)

    def get_examples(self):
        xs = self.generator.get_examples()
        if isinstance(xs, torch.Tensor):
            return self.trans(xs) if callable(self.trans) else self.trans[0](xs)
        if callable(self.trans):
            return self.trans(*xs)
        return tuple(t(x) for t, x in zip(self.trans, xs))

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generator=self.generator, trans=self.trans)
        return d


class FilterGenerator(BaseGenerator):
    """Keeps the samples where ``filter_fn(list of sample vectors)`` (a boolean mask) holds (generators.py:904-952)."""

    def __init__(self, generator, filter_fn, size=None, update_size=True):
        super().__init__()
        self.generator, self.filter_fn = generator, filter_fn
        self.size = generator.size if size is None else size
        self.update_size = update_size

    def get_examples(self):
        xs = self.generator.get_examples()
        xs = [xs] if isinstance(xs, torch.Tensor) else xs
        mask = self.filter_fn(xs)
        xs = [x[mask] for x in xs]
        if self.update_size:
            self.size = len(xs[0])
        return xs[0] if len(xs) == 1 else xs

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generator=self.generator, filter_fn=self.filter_fn)
        return d


class ResampleGenerator(BaseGenerator):
    """Shuffled sub-sample of another generator's output, with or without replacement (generators.py:955-993); the index
    draw (``randint`` / ``randperm``) precedes the inner generator's own draw, like in the reference."""

    def __init__(self, generator, size=None, replacement=False):
        super().__init__()
        self.generator = generator
        self.size = generator.size if size is None else size
        self.replacement = replacement

    def get_examples(self):
        if self.replacement:
            idx = torch.randint(self.generator.size, (self.size,), device=_CPU)
        else:
            idx = torch.randperm(self.generator.size, device=_CPU)[:self.size]
        xs = self.generator.get_examples()
        return xs[idx] if isinstance(xs, torch.Tensor) else [x[idx] for x in xs]

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generator=self.generator, replacement=self.replacement)
        return d


class BatchGenerator(BaseGenerator):
    """Serves another generator's samples ``batch_size`` at a time from a cache that is refilled when it runs short
    (generators.py:996-1043)."""

    def __init__(self, generator, batch_size):
        super().__init__()
        if generator.size <= 0:
            raise ValueError(f"generator has size {generator.size} <= 0")
        self.generator, self.size = generator, batch_size
        first = generator.get_examples()
        self.cached_xs = [first] if isinstance(first, torch.Tensor) else list(first)

    def get_examples(self):
        while len(self.cached_xs[0]) < self.size:
            more = self.generator.get_examples()
            more = [more] if isinstance(more, torch.Tensor) else more
            self.cached_xs = [torch.cat([x, m]) for x, m in zip(self.cached_xs, more)]
        batch = [x[:self.size] for x in self.cached_xs]
        self.cached_xs = [x[self.size:] for x in self.cached_xs]
        return batch[0] if len(batch) == 1 else batch

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generator=self.generator)
        return d


_STATIC_METHODS = {Generator1D: ("equally-spaced", "log-spaced", "chebyshev", "chebyshev1", "chebyshev2"),
                   Generator2D: ("equally-spaced", "chebyshev", "chebyshev1", "chebyshev2", "latin-hypercube"),
                   Generator3D: ("equally-spaced", "chebyshev", "chebyshev1", "chebyshev2")}


def draws_are_static(g):
    """True if ``g.get_examples()`` provably returns the same points on every call without touching the RNG (the default
    validation grids, StaticGenerator, PredefinedGenerator, combinations of those)."""
    if isinstance(g, SamplerGenerator):
        return draws_are_static(g.generator)
    if type(g) in _STATIC_METHODS:
        return g.method in _STATIC_METHODS[type(g)]
    if type(g) in (StaticGenerator, PredefinedGenerator):
        return True
    if type(g) is GeneratorND:
        return not g.noisy and "uniform" not in (g.methods if not isinstance(g.methods, str) else (g.methods,))
    if type(g) in (ConcatGenerator, EnsembleGenerator, MeshGenerator):
        return all(draws_are_static(x) for x in g.generators)
    return False


def draws_have_fixed_size(g):
    """True if every ``g.get_examples()`` is known to return ``g.size`` points (the reference's own generator classes
    except FilterGenerator / TransformGenerator, whose user callables may do anything)."""
    if isinstance(g, SamplerGenerator):
        return draws_have_fixed_size(g.generator)
    if type(g) in (Generator1D, Generator2D, Generator3D, GeneratorND, GeneratorSpherical, StaticGenerator,
                   PredefinedGenerator, ResidentBatchGenerator):
        return True
    if type(g) in (ConcatGenerator, EnsembleGenerator, MeshGenerator):
        return all(draws_have_fixed_size(x) for x in g.generators)
    if type(g) is ResampleGenerator and not g.replacement and g.size > g.generator.size:
        return False                  # randperm(generator.size)[:size] yields generator.size points, not `size`
    if type(g) in (ResampleGenerator, BatchGenerator):
        return draws_have_fixed_size(g.generator)
    return False


class SamplerGenerator(BaseGenerator):
    """Solver-side wrapper: always returns a list of ``(N, 1)`` columns that require grad (generators.py:1046-1064)."""

    def __init__(self, generator):
        super().__init__()
        self.generator, self.size = generator, generator.size
        self._last = None

    def get_examples(self):
        samples = self.generator.get_examples()
        if isinstance(samples, torch.Tensor):
            samples = [samples]
        if samples[0].device.type == "cuda" and not samples[0].requires_grad:
            if samples[0].dim() == 2:
                return samples                                   # resident batch: already (N, 1) views, zero-copy
            return [s.reshape(-1, 1) for s in samples]
        # a static generator hands back the very same tensors every time (default validation grids): serve the same
        # columns again (the engine recognises them and skips the upload); fresh leaves as far as autograd can tell
        last = self._last
        if last is not None and len(last[0]) == len(samples) and \
                all(a is b and b._version == v for a, b, v in zip(last[0], samples, last[1])):
            for c in last[2]:
                c.grad = None
            return last[2]
        cols = [s.reshape(-1, 1).detach().requires_grad_(True) for s in samples]
        self._last = (tuple(samples), tuple(s._version for s in samples), cols)
        return cols


class ResidentBatchGenerator(BaseGenerator):
    """Pre-sampled batches kept resident in HBM as SoA blocks ``[n_batches][d][ld]`` and served round-robin.

    New (no reference counterpart): it removes the host sampling + PCIe hand-off from a training step when the
    points can be drawn ahead of time -- ``ResidentBatchGenerator.presample(gen, k, device)`` draws ``k`` batches from
    any generator in the reference's RNG order (so they are the same points the reference would train on) and
    uploads them once.  ``get_examples`` returns views into the current block; the fused engine recognises them and
    reads the block in place (no copy)."""

    def __init__(self, blocks, size):
        super().__init__()
        self.blocks, self.size, self._next = blocks, size, 0
        self._views = {}

    @classmethod
    def presample(cls, generator, n_batches, device, lo=0, hi=None):
        draws = []
        for _ in range(n_batches):
            ex = generator.get_examples()
            ex = [ex] if isinstance(ex, torch.Tensor) else list(ex)
            draws.append(torch.stack([e.detach().reshape(-1)[lo:hi] for e in ex]))
        n = draws[0].shape[1]
        ld = (n + 63) // 64 * 64
        blocks = torch.zeros(n_batches, draws[0].shape[0], ld, dtype=torch.float32, device=device)
        blocks[:, :, :n] = torch.stack(draws).to(device)
        return cls(blocks, n)

    def get_examples(self):
        k = self._next % self.blocks.shape[0]
        self._next += 1
        views = self._views.get(k)
        if views is None:         # (N, 1) column views of the block's rows, built once per block
            blk = self.blocks[k]
            views = self._views[k] = [blk[i, :self.size].reshape(-1, 1) for i in range(blk.shape[0])]
        return views


# (N, 1) view lists handed out by DeviceGenerators -> the generator (engine.fast_train_epoch asks for a prefetch)
_DEVICE_SOURCES = {}


def device_source(batch):
    """The DeviceGenerator that handed out ``batch`` (its own list of views), or None."""
    ref = _DEVICE_SOURCES.get(id(batch))
    g = ref() if ref is not None else None
    return g if (g is not None and any(v is batch for v in g._views_all)) else None


class DeviceGenerator(BaseGenerator):
    """Draws the distribution of a reference generator ON the MI355X (csrc/ndq_sample.h through ``ndq_sample``).

    New (no reference counterpart): host sampling is the floor of a fused step
    (``torch.normal`` over 2 x 65 536 points costs ~10x the whole fused C2 step).  Wrapping a generator keeps its
    distribution -- grid geometry, jitter standard deviation, radial law -- but draws from a counter-based
    Philox4x32-10 stream: batch number ``k`` of a given ``seed`` (default ``torch.initial_seed()`` at construction) and
    ``stream_id`` (default: the process rank, so data-parallel ranks draw disjoint streams) is always the same
    points, yet they are NOT the numbers the reference's host generator would produce, hence opt-in.

    Supported: ``Generator1D`` ('uniform', 'equally-spaced', 'equally-spaced-noisy'), ``Generator2D`` / ``Generator3D``
    ('equally-spaced', 'equally-spaced-noisy'), ``GeneratorSpherical`` (both radial laws).  ``get_examples`` enqueues
    one kernel on the current stream and returns ``(N, 1)`` views of ONE resident SoA block which the fused engine
    reads in place; the block is overwritten by the next draw (stream-ordered, so the previous step has consumed it).
    (Drawing the next batch on a side stream while the current one trains was tried: the event waits between the two
    streams cost more than the 4 us kernel they hide -- 41 us per step instead of 33 -- and it was dropped.)
    """

    def __init__(self, generator, device=None, seed=None, stream_id=None, prefetch=False, dtype=torch.float32):
        """dtype: torch.float32, or torch.float64 for fp64 solvers (the reference's default precision): the kernel draws fp32
        points, the handed-out tensors are their exact images in double (one device-side copy per draw)."""
        super().__init__()
        from . import _lib
        if dtype not in (torch.float32, torch.float64):
            raise ValueError(f"dtype must be torch.float32 or torch.float64, got {dtype}")
        if not torch.cuda.is_available():
            raise _lib.NdqError("DeviceGenerator samples with a gfx950 kernel and needs an MI355X; use the wrapped "
                                "generator itself for host sampling")
        self.generator, self.size = generator, generator.size
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.stream_id = int(os.environ.get("RANK", "0")) if stream_id is None else int(stream_id)
        self.draw = 0
        self.desc = self.describe(generator)
        # the descriptor froze what the wrapped generator draws from; the reference reads it at every draw (generators.py:107-416),
        # so a callback that changes a noise width / replaces a grid tensor or the getter has to be seen: live_stamp per draw
        self._live_names = tuple(n for n in _LIVE if n in vars(generator))
        self._restamp()
        self._on_host = False
        self._L = _lib.lib()
        ld = (self.size + 63) // 64 * 64
        # prefetch=True: a solver on the single-launch native path lets the extra workgroups of its sums + tail kernel
        # draw the NEXT batch (ndq_fused_step.next_sampler) -- the sampler launch leaves the step.  The points are the
        # same (draw k is a function of (seed, k, stream_id) only).  Two blocks alternate (draw k lives in block k & 1), so
        # the tensors handed out for an epoch keep that epoch's points until the END of the following epoch.
        self.prefetch = bool(prefetch)
        self.blocks = [torch.zeros(self.desc.d, ld, dtype=torch.float32, device=self.device) for _ in range(2 if self.prefetch else 1)]
        self.dtype = dtype
        # fp64: the tensors handed out are views of double blocks the fp32 draws are copied into
        self._out_blocks = self.blocks if dtype == torch.float32 else [torch.zeros_like(b, dtype=dtype) for b in self.blocks]
        self._views_all = [[blk[i, :self.size].reshape(-1, 1) for i in range(self.desc.d)] for blk in self._out_blocks]
        self.block, self._views = self.blocks[0], self._views_all[0]
        self.prefetched = None       # draw number already sitting in its block, drawn ahead by a tail kernel
        self.launches = 0            # sampler kernels this generator launched itself (diagnostics / tests)
        # (weak: the registry must not keep a generator -- two device blocks -- alive after its solver is gone; the entries go
        # with it, before the ids of its view lists can be reused)
        for views in self._views_all:
            _DEVICE_SOURCES[id(views)] = weakref.ref(self)
            weakref.finalize(self, _DEVICE_SOURCES.pop, id(views), None)

    @staticmethod
    def describe(g):
        """``ndq_sampler_desc`` of a reference-style generator; ``ValueError`` for what the kernel does not draw."""
        from . import _lib
        d = _lib.SamplerDesc()
        grid_methods = ("equally-spaced", "equally-spaced-noisy")

        def grid(ns, lo, hi, std):
            d.kind, d.d = _lib.NDQ_SAMPLE_GRID, len(ns)
            for i in range(len(ns)):
                d.n[i], d.lo[i], d.hi[i], d.noise_std[i] = int(ns[i]), float(lo[i]), float(hi[i]), float(std[i])
        if isinstance(g, Generator1D) and g.method == "uniform":
            d.kind, d.d = _lib.NDQ_SAMPLE_UNIFORM, 1
            d.n[0], d.lo[0], d.hi[0] = g.size, g.t_min, g.t_max
        elif isinstance(g, Generator1D) and g.method in grid_methods:
            grid([g.size], [g.t_min], [g.t_max], [g.noise_std if g.method.endswith("noisy") else 0.0])
        elif isinstance(g, Generator2D) and g.method in grid_methods:
            grid(g.grid, g.xy_min, g.xy_max, [g.noise_xstd, g.noise_ystd] if g.method.endswith("noisy") else [0.0, 0.0])
        elif isinstance(g, Generator3D) and g.method in grid_methods:
            grid(g.grid, g.xyz_min, g.xyz_max, g.noise_std if g.method.endswith("noisy") else [0.0] * 3)
        elif isinstance(g, GeneratorSpherical):
            d.kind, d.d = _lib.NDQ_SAMPLE_SPHERICAL, 3
            d.n[0], d.lo[0], d.hi[0] = g.size, g.r_min, g.r_max
            d.radial = int(g.method == "equally-radius-noisy")
        else:
            raise ValueError(f"DeviceGenerator cannot draw {g!r} on the device")
        return d

    def get_examples(self):
        if torch._C._len_torch_function_stack():          # a global default-device mode: see engine.library_code
            with torch._C.DisableTorchFunction():
                return self._get_examples()
        return self._get_examples()

    def block_of(self, draw):
        """The block draw number ``draw`` lives in."""
        return self.blocks[draw & 1] if self.prefetch else self.blocks[0]

    def _restamp(self):
        g = self.generator
        self._stamp = live_stamp(g, self._live_names)
        d = vars(g)
        # the per-draw form of the same stamp: object identity of every live attribute + the version counters of the tensors
        self._quick = tuple((n, d.get(n)) for n in self._live_names)
        self._quick_t = tuple((v, v._version) for _, v in self._quick if isinstance(v, torch.Tensor))
        self._quick_f = type(g).get_examples

    def _unchanged(self):
        g = self.generator
        d = g.__dict__
        for n, v in self._quick:
            if d.get(n) is not v:
                return False
        for t, ver in self._quick_t:
            if t._version != ver:
                return False
        return type(g).get_examples is self._quick_f

    def _wrapped_changed(self):
        """The wrapped generator no longer draws what the descriptor says.  Noise widths alone: a new descriptor.  Anything else (grid
        tensors replaced / edited in place, another getter, another size): the wrapped generator's own host draw from now on,
        copied into the resident block -- what the reference would train on."""
        g = self.generator
        try:
            new = self.describe(g)
            same_shape = new.kind == self.desc.kind and new.d == self.desc.d and list(new.n) == list(self.desc.n) and g.size == self.size
        except Exception:       # noqa: BLE001
            new, same_shape = None, False
        numbers_only = same_shape and all(
            a == b for n, a, b in zip(("type",) + self._live_names, self._stamp, live_stamp(g, self._live_names))
            if not n.startswith("noise"))
        if numbers_only:
            self.desc = new
            self.prefetched = None              # (a batch drawn ahead by a tail kernel used the old widths)
        else:
            self._on_host, self.prefetch, self.prefetched = True, False, None
            warnings.warn("neurodiffeq_amd: a generator wrapped by DeviceGenerator was changed in a way the device sampler cannot "
                          "follow (grid tensors / getter / size); its own host draw is used from now on (the reference's numbers, "
                          "uploaded every epoch).", RuntimeWarning)
        self._restamp()

    def _host_examples(self):
        ex = self.generator.get_examples()
        ex = [ex] if isinstance(ex, torch.Tensor) else list(ex)
        if len(ex) != self.desc.d or ex[0].numel() != self.size:
            self.size = ex[0].numel()
            return [e.detach().reshape(-1, 1).to(self.device, self.dtype) for e in ex]      # (another shape altogether: plain tensors)
        block, views = self._out_blocks[0], self._views_all[0]
        for i, e in enumerate(ex):
            block[i, :self.size].copy_(e.detach().reshape(-1))
        self.block, self._views = self.blocks[0], views
        self.draw += 1
        return views

    def _get_examples(self):
        if self._on_host:
            return self._host_examples()
        if not self._unchanged():
            self._wrapped_changed()
            if self._on_host:
                return self._host_examples()
        block = self.block_of(self.draw)
        if self.prefetched == self.draw:          # a tail kernel has drawn this batch already
            self.prefetched = None
        else:
            stream = ctypes.c_void_p(_raw_stream(self.device.index))       # (torch.cuda.current_stream(): ~10 us per call)
            rc = self._L.ndq_sample(ctypes.byref(self.desc), self.seed, self.draw, self.stream_id, block.data_ptr(),
                                    block.shape[1], stream)
            if rc != 0:
                from . import _lib
                raise _lib.NdqError(f"ndq_sample failed with code {rc}")
            self.launches += 1
        slot = (self.draw & 1) if self.prefetch else 0
        if self.dtype != torch.float32:
            self._out_blocks[slot].copy_(block)          # (stream-ordered behind the draw)
        views = self._views_all[slot]
        self.block, self._views = block, views
        self.draw += 1
        return views

    def _internal_vars(self):
        d = super()._internal_vars()
        d.update(generator=self.generator, seed=self.seed, stream_id=self.stream_id)
        return d


# ------------------------------------------------------------------------------------------------ default-device sampling
# The reference draws on torch's DEFAULT device (generators.py:152,158,264-265: torch.linspace / torch.normal without a
# device argument); its import default is cuda where one exists (neurodiffeq/__init__.py:22), and there the noise comes from
# the GPU's Philox stream -- itself not the CPU generator's numbers.  Same rule here: a solver built while torch's default
# device is cuda draws the NOISE of Generator1D / 2D / 3D / Spherical on the MI355X (DeviceGenerator: Philox4x32-10, seeded
# from torch's cuda generator (one draw per wrapped generator), so torch.manual_seed still fixes the run; every rank draws the SAME batch and takes its
# shard); with a CPU default device nothing changes: host draws, bit for bit the reference's CPU numbers.  Index sampling
# (randperm / randint: ResampleGenerator, BatchGenerator, latin-hypercube, the spherical signs of the host path) and every
# wrapper generator stay on the CPU generator bit for bit in both cases.  ``Generator2D(..., bit_exact_cpu=True)`` or
# ``set_default_sampling("cpu")`` keep a generator on the host under a cuda default device as well.
_SAMPLING = {"mode": os.environ.get("NDQ_SAMPLING", "auto")}


def set_default_sampling(mode):
    """"auto" (default): follow torch's default device;  "cpu": always sample on the host (the reference's CPU numbers bit
    for bit, 17x slower per step at the headline size);  "device": device-side noise whenever an MI355X is visible."""
    if mode not in ("auto", "cpu", "device"):
        raise ValueError(f"mode must be 'auto', 'cpu' or 'device', got {mode!r}")
    _SAMPLING["mode"] = mode


def _seed_from_torch_cuda_rng():
    dev = torch.device("cuda", torch.cuda.current_device())
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64, device=dev).item())


def on_default_device(gen):
    """``gen`` itself, or -- when noise is to be drawn on the device (see above) and the kernel knows the distribution -- a
    DeviceGenerator around it."""
    mode = _SAMPLING["mode"]
    if mode == "cpu" or gen is None or getattr(gen, "bit_exact_cpu", False) or not torch.cuda.is_available():
        return gen
    if type(gen) not in (Generator1D, Generator2D, Generator3D, GeneratorSpherical):
        return gen
    if mode == "auto" and torch.get_default_device().type != "cuda":
        return gen
    dtype = torch.get_default_dtype()
    if dtype not in (torch.float32, torch.float64):
        return gen
    if not (type(gen) is GeneratorSpherical or gen.method == "uniform" or gen.method == "equally-spaced-noisy"):
        return gen                 # static grids are uploaded once and read in place; other laws: host
    try:
        # prefetch: on the single-launch native path the next batch is drawn by spare workgroups of the epoch's own sums /
        # tail launch (no sampler launch); the handed-out tensors keep an epoch's points until the end of the next epoch
        # (an fp64 default dtype -- the reference's import default is cuda + float64 -- gets the fp32 draws as doubles)
        # the Philox key is (seed, draw number, stream id): every wrapped generator needs a seed of its OWN, or two generators
        # of one law -- the default train / valid pair of SolverSpherical, every solver of a process -- would replay the same
        # points.  The reference's generators all consume the one global RNG of the default device; here each
        # takes its seed from that RNG once, at construction: torch.manual_seed still fixes the run, data-parallel ranks
        # seeded alike still draw the same batches, and no two generators share a stream.
        return DeviceGenerator(gen, seed=_seed_from_torch_cuda_rng(), stream_id=0, prefetch=True, dtype=dtype)
    except ValueError:
        return gen
