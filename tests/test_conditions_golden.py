"""Every condition class re-parameterises exactly like the reference's (tests/golden/conditions.npz, produced by
running the unmodified reference through tests/condition_specs.py), fp64."""
import os

import numpy as np
import pytest

import neurodiffeq_amd.conditions as C
import neurodiffeq_amd.networks as N
from tests import condition_specs as S


@pytest.mark.parametrize("name", list(S.SPECS))
def test_enforce_matches_the_reference(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "conditions.npz"))
    got = S.SPECS[name](C, N).detach().numpy()
    want = gold[name]
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-12, atol=1e-13), np.abs(got - want).max()
