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
