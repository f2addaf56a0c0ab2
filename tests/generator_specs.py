"""Generator constructions shared by tests/golden/make_golden.py (run on the reference's ``neurodiffeq.generators``) and
tests/test_generators_golden.py (run on ``neurodiffeq_amd.generators``): ``SPECS[name](G)`` builds the generator from
module ``G``.  Draws must agree bit for bit under the same ``torch.manual_seed`` (north_star: same sampled inputs)."""
import torch

SPECS = {
    "g1d_uniform": lambda G: G.Generator1D(33, 0.1, 2.0, "uniform"),
    "g1d_noisy": lambda G: G.Generator1D(33, 0.1, 2.0, "equally-spaced-noisy"),
    "g1d_log_noisy": lambda G: G.Generator1D(20, 0.1, 10.0, "log-spaced-noisy"),
    "g1d_cheb2_noisy": lambda G: G.Generator1D(17, -1.0, 1.0, "chebyshev2-noisy"),
    "g1d_lhs": lambda G: G.Generator1D(16, 0.0, 1.0, "latin-hypercube"),
    "g2d_noisy": lambda G: G.Generator2D((7, 5), (0.0, -1.0), (1.0, 1.0), "equally-spaced-noisy"),
    # Generator2D(method="chebyshev2-noisy") cannot be pinned: the reference's own get_examples raises there
    # (generators.py:303, its getter is a tuple)
    "g2d_lhs": lambda G: G.Generator2D((5, 5), method="latin-hypercube"),
    "g3d_noisy": lambda G: G.Generator3D((3, 4, 5), (0, 0, 0), (1, 2, 3)),
    "gsph": lambda G: G.GeneratorSpherical(64, 0.5, 2.0),
    "gsph_radius": lambda G: G.GeneratorSpherical(64, 0.5, 2.0, "equally-radius-noisy"),
    "nd_mixed": lambda G: G.GeneratorND((5, 4, 3), (0, 0.1, -1), (1, 2, 1), ("equally-spaced", "log-spaced", "chebyshev2")),
    "nd_uniform_exp_cut": lambda G: G.GeneratorND((6, 5), (0.0, 0.0), (1.0, 2.0), ("uniform", "exp-spaced"), base=(10, 2),
                                                  cut=((1, None), (None, -1))),
    "nd_1d_abs": lambda G: G.GeneratorND(7, 0.0, 1.0, "chebyshev", abs_value=True, r_noise_std=0.3),
    "nd_static": lambda G: G.GeneratorND((4, 4), noisy=False),
    "concat": lambda G: G.Generator1D(5, method="uniform") + G.Generator1D(7, 1.0, 2.0, "equally-spaced-noisy"),
    "ensemble": lambda G: G.Generator1D(12, method="uniform") * G.Generator2D((3, 4)),
    "mesh": lambda G: G.Generator1D(4) ^ G.Generator1D(3, method="uniform") ^ G.Generator1D(2),
    "static": lambda G: G.StaticGenerator(G.Generator2D((4, 4))),
    "predefined": lambda G: G.PredefinedGenerator([0.0, 0.5, 1.0], [1.0, 2.0, 3.0]),
    "transform_list": lambda G: G.TransformGenerator(G.Generator2D((4, 4)), transforms=[torch.sin, None]),
    "transform_fn": lambda G: G.TransformGenerator(G.Generator2D((4, 4)), transform=lambda x, y: (x + y, x - y)),
    "filter": lambda G: G.FilterGenerator(G.Generator2D((8, 8)), lambda xs: xs[0] > 0.5),
    "resample": lambda G: G.ResampleGenerator(G.Generator1D(50, method="uniform"), size=20),
    "resample_repl": lambda G: G.ResampleGenerator(G.Generator2D((6, 6)), size=50, replacement=True),
    "batch": lambda G: G.BatchGenerator(G.Generator1D(10, method="uniform"), 7),
    "sampler": lambda G: G.SamplerGenerator(G.Generator2D((3, 3))),
}


def draws(G, name, seed=3, n_draws=3):
    """[n_draws][n_vectors] list of numpy arrays + the generator's size after the draws"""
    torch.manual_seed(seed)
    g = SPECS[name](G)
    out = []
    for _ in range(n_draws):
        ex = g.get_examples()
        ex = [ex] if isinstance(ex, torch.Tensor) else list(ex)
        out.append([e.detach().reshape(-1).numpy().copy() for e in ex])
    return out, int(g.size)
