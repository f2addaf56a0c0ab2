"""operators.* and function_basis.* evaluate exactly like the reference's (tests/golden/operators.npz, produced by
running the unmodified reference through tests/operator_specs.py), fp64."""
import os

import numpy as np
import pytest

import neurodiffeq_amd.function_basis as B
import neurodiffeq_amd.operators as O
from tests import operator_specs as S


@pytest.mark.parametrize("name", list(S.SPECS))
def test_matches_the_reference(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "operators.npz"))
    got = S.SPECS[name](O, B).detach().numpy()
    want = gold[name]
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(want).max())), np.abs(got - want).max()
