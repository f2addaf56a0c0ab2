"""Every generator of neurodiffeq_amd.generators draws bit for bit what the reference's does under the same seed
(tests/golden/generators.npz, produced by running the unmodified reference through tests/generator_specs.py)."""
import os

import numpy as np
import pytest

import neurodiffeq_amd.generators as G
from tests import generator_specs as S


@pytest.mark.parametrize("name", list(S.SPECS))
def test_draws_are_bit_identical_to_the_reference(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "generators.npz"))
    got, size = S.draws(G, name)
    assert size == int(gold[f"{name}/size"])
    for d, vectors in enumerate(got):
        assert len(vectors) == int(gold[f"{name}/n_vectors"])
        for v, arr in enumerate(vectors):
            want = gold[f"{name}/{d}/{v}"]
            assert arr.dtype == want.dtype and arr.shape == want.shape and np.array_equal(arr, want), (name, d, v)


def test_bulk_draws_follow_what_a_callback_changes_on_the_generator():
    """fit()'s multi-epoch path draws the batches of a whole chunk with ONE call (bulk_examples), proven once per generator to
    equal the sequence of get_examples() calls.  The reference's generators read their grid tensors / noise widths / getter when
    they draw, so that proof -- and anything derived from those attributes -- holds only for the state it was made in."""
    import torch
    from neurodiffeq_amd.generators import Generator1D, Generator2D

    def sequence(g, k):
        out = []
        for _ in range(k):
            ex = g.get_examples()
            ex = [ex] if isinstance(ex, torch.Tensor) else ex
            out.append(torch.stack([e.detach().reshape(-1) for e in ex]))
        return torch.stack(out)
    g = Generator2D((8, 8), (0, 0), (1, 1), "equally-spaced-noisy")
    g1 = Generator1D(32, 0.0, 1.0, "equally-spaced-noisy")
    changes = [(g, None), (g, lambda: setattr(g, "noise_xstd", 0.5)), (g, lambda: setattr(g, "grid_x", g.grid_x.detach() * 2)),
               (g, lambda: g.grid_y.data.mul_(3.0)),
               (g1, None), (g1, lambda: setattr(g1, "noise_std", 0.5)), (g1, lambda: g1.examples.data.mul_(2.0)),
               (g1, lambda: setattr(g1, "examples", torch.linspace(0, 3, 32)))]
    for gen, change in changes:
        if change:
            change()
        torch.manual_seed(1)
        bulk = gen.bulk_examples(3)
        after_bulk = torch.get_rng_state()
        torch.manual_seed(1)
        assert bulk is not None and torch.equal(bulk, sequence(gen, 3)) and torch.equal(after_bulk, torch.get_rng_state())
    for gen in (g, g1):            # another getter: no shortcut any more
        gen.getter = (lambda: (g.grid_x, g.grid_y)) if gen is g else (lambda: g1.examples)
        assert gen.bulk_examples(3) is None
