"""Device-side sampler (csrc/ndq_sample.h, generators.DeviceGenerator): Philox known answers, the numpy restatement's
distributions against the host generators it stands in for, and -- on the GPU -- the kernel against the restatement."""
import ctypes
import math

import numpy as np
import pytest
import torch
from scipy import stats

from oracle import philox_ref as P
from neurodiffeq_amd import _lib
from neurodiffeq_amd.generators import (DeviceGenerator, Generator1D, Generator2D, Generator3D, GeneratorSpherical)


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10 (counter, key -> output)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10([[c] for c in ctr], key)[:, 0]
        assert tuple(int(v) for v in got) == want


def test_linspace_restatement_is_torch_linspace():
    for lo, hi, n in [(0.0, 1.0, 256), (-1.0, 1.0, 511), (0.1, 12.0, 1024), (0.0, 1.0, 1), (2.0, 3.0, 2)]:
        assert np.array_equal(P.linspace(lo, hi, n), torch.linspace(lo, hi, n).numpy())


def test_restated_distributions_match_the_host_generators():
    torch.manual_seed(0)
    # noisy grid: jitter ~ N(0, (dx/4)^2) per axis, independent between axes, around the exact ij-meshgrid
    g = Generator2D((64, 48), (0.0, -1.0), (1.0, 2.0), "equally-spaced-noisy")
    pts = P.sample_grid(g.grid, g.xy_min, g.xy_max, [g.noise_xstd, g.noise_ystd], seed=7, draw=0)
    jx, jy = pts[0] - g.grid_x.detach().numpy(), pts[1] - g.grid_y.detach().numpy()
    assert stats.kstest(jx / g.noise_xstd, "norm").pvalue > 1e-3 and stats.kstest(jy / g.noise_ystd, "norm").pvalue > 1e-3
    assert abs(np.corrcoef(jx, jy)[0, 1]) < 0.06
    hx, hy = g.get_examples()
    assert stats.ks_2samp(jx, (hx - g.grid_x).detach().numpy()).pvalue > 1e-3
    exact = P.sample_grid(g.grid, g.xy_min, g.xy_max, [0.0, 0.0], seed=7, draw=0)
    assert np.array_equal(exact[0], g.grid_x.detach().numpy()) and np.array_equal(exact[1], g.grid_y.detach().numpy())
    # uniform
    u = P.sample_uniform(20000, [0.1], [12.0], seed=3, draw=5)[0]
    assert u.min() >= 0.1 and u.max() < 12.0 and stats.kstest((u - 0.1) / 11.9, "uniform").pvalue > 1e-3
    # spherical shell: same marginals as GeneratorSpherical
    for method, radial in (("equally-spaced-noisy", 0), ("equally-radius-noisy", 1)):
        h = GeneratorSpherical(20000, 0.5, 2.0, method)
        r, th, ph = (t.detach().numpy() for t in h.get_examples())
        s = P.sample_spherical(20000, 0.5, 2.0, radial, seed=11, draw=2)
        assert s[0].min() >= 0.5 and s[0].max() <= 2.0 and 0 <= s[1].min() and s[1].max() <= math.pi
        assert 0 <= s[2].min() and s[2].max() <= 2 * math.pi + 1e-6
        for a, b in zip(s, (r, th, ph)):
            assert stats.ks_2samp(a, b).pvalue > 1e-3
    # draws and streams differ, (seed, draw, stream) repeats
    a = P.words(1000, 1, 0, 0); b = P.words(1000, 1, 1, 0); c = P.words(1000, 1, 0, 1)
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99 and np.array_equal(a, P.words(1000, 1, 0, 0))


def test_descriptor_mapping_and_unsupported_generators():
    d = DeviceGenerator.describe(Generator2D((256, 128), (0, -1), (1, 1), "equally-spaced-noisy"))
    assert (d.kind, d.d, list(d.n)[:2]) == (_lib.NDQ_SAMPLE_GRID, 2, [256, 128])
    assert d.noise_std[0] == pytest.approx(1 / 256 / 4) and d.noise_std[1] == pytest.approx(2 / 128 / 4)
    d = DeviceGenerator.describe(Generator2D((8, 8), method="equally-spaced"))
    assert d.noise_std[0] == 0 and d.noise_std[1] == 0
    d = DeviceGenerator.describe(Generator1D(100, 0.5, 2.5, "uniform"))
    assert (d.kind, d.d, d.n[0], d.lo[0], d.hi[0]) == (_lib.NDQ_SAMPLE_UNIFORM, 1, 100, 0.5, 2.5)
    d = DeviceGenerator.describe(Generator1D(100, 0.0, 1.0, "equally-spaced-noisy", noise_std=0.125))
    assert d.kind == _lib.NDQ_SAMPLE_GRID and d.noise_std[0] == 0.125
    d = DeviceGenerator.describe(Generator3D((4, 5, 6)))
    assert (d.d, list(d.n)) == (3, [4, 5, 6])
    d = DeviceGenerator.describe(GeneratorSpherical(64, 0.1, 3.0, "equally-radius-noisy"))
    assert (d.kind, d.d, d.radial) == (_lib.NDQ_SAMPLE_SPHERICAL, 3, 1)
    for bad in (Generator1D(8, 0.1, 1.0, "log-spaced"), Generator2D((4, 4), method="chebyshev"), Generator1D(8) + Generator1D(8)):
        with pytest.raises(ValueError):
            DeviceGenerator.describe(bad)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.NdqError):
            DeviceGenerator(Generator1D(8))


# ---------------------------------------------------------------------------------------------------- on the MI355X
def _close(got, want, scale):
    return np.abs(got - want).max() <= 4e-6 * scale


@pytest.mark.gpu
def test_device_sampler_matches_the_restatement():
    torch.manual_seed(123)
    g2 = Generator2D((256, 256), (0, 0), (1, 1), "equally-spaced-noisy")
    dg = DeviceGenerator(g2, seed=99, stream_id=3)
    for draw in range(3):
        x, y = (v.reshape(-1).cpu().numpy() for v in dg.get_examples())
        want = P.sample_grid(g2.grid, g2.xy_min, g2.xy_max, [g2.noise_xstd, g2.noise_ystd], 99, draw, 3)
        assert _close(x, want[0], 1.0) and _close(y, want[1], 1.0)
    dg = DeviceGenerator(Generator2D((17, 5), (-1, 0), (1, 1), "equally-spaced"))
    x, y = (v.reshape(-1).cpu() for v in dg.get_examples())
    ref = Generator2D((17, 5), (-1, 0), (1, 1), "equally-spaced").get_examples()
    assert torch.equal(x, ref[0].detach()) and torch.equal(y, ref[1].detach())            # exact grid: bit-exact
    g1 = Generator1D(1000, 0.1, 12.0, "equally-spaced-noisy")
    t = DeviceGenerator(g1, seed=5).get_examples()[0].reshape(-1).cpu().numpy()
    assert _close(t, P.sample_grid([1000], [0.1], [12.0], [g1.noise_std], 5, 0, 0)[0], 12.0)
    t = DeviceGenerator(Generator1D(4099, -2.0, 3.0, "uniform"), seed=6).get_examples()[0].reshape(-1).cpu().numpy()
    assert _close(t, P.sample_uniform(4099, [-2.0], [3.0], 6, 0)[0], 5.0)
    g3 = Generator3D((7, 9, 11), (0, 0, 0), (1, 2, 3))
    got = [v.reshape(-1).cpu().numpy() for v in DeviceGenerator(g3, seed=8).get_examples()]
    want = P.sample_grid(g3.grid, g3.xyz_min, g3.xyz_max, g3.noise_std, 8, 0)
    assert all(_close(a, b, 3.0) for a, b in zip(got, want))
    for method, radial in (("equally-spaced-noisy", 0), ("equally-radius-noisy", 1)):
        gs = GeneratorSpherical(5000, 0.1, 3.0, method)
        got = [v.reshape(-1).cpu().numpy() for v in DeviceGenerator(gs, seed=9).get_examples()]
        want = P.sample_spherical(5000, 0.1, 3.0, radial, 9, 0)
        # acos / atan2 are ill-conditioned near the poles / axis: compare through the direction cosines
        assert _close(got[0], want[0], 3.0)
        assert _close(np.cos(got[1]), np.cos(want[1]), 1.0) and _close(np.sin(got[2]), np.sin(want[2]), 2.0)


@pytest.mark.gpu
def test_solver_trains_on_device_drawn_batches():
    """fit() with a DeviceGenerator: zero-sync native epochs on a block drawn in place each epoch; every epoch's loss is
    the loss of that epoch's (reproducible) batch under the oracle."""
    from oracle import autograd_ref as R
    from tests import configs
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2", 64)
    gen = DeviceGenerator(cfg["gen"], seed=42)
    solver.generator["train"].generator = gen
    solver.fused = "require"
    flat0 = R.get_flat(cfg["nets"]).cpu().clone()
    solver.fit(3, tqdm_file=None)
    assert solver.fused_active and gen.draw == 3
    hist = solver.metrics_history["train_loss"]
    g = cfg["gen"]
    torch.manual_seed(0)
    ocfg = R.build_config("c2", 64, dtype=torch.float64)
    R.set_flat(ocfg["nets"], flat0.double())
    batches = [P.sample_grid(g.grid, g.xy_min, g.xy_max, [g.noise_xstd, g.noise_ystd], 42, k) for k in range(3)]
    loop = R.TrainLoop(ocfg["nets"], ocfg["enforcers"], ocfg["pde"],
                       lambda it=iter(batches): [torch.from_numpy(c).double() for c in next(it)])
    want = [loop.epoch() for _ in range(3)]
    assert np.allclose(hist, want, rtol=2e-5), (hist, want)
