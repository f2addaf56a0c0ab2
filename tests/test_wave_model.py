"""The lane-level numpy model of the HIP kernels (tests/wave_model.py) against the jet oracle (oracle/jet_ref.py):
checks the fragment layout / permuted contraction / LDS transpose index algebra without a GPU."""
import numpy as np
import pytest

from oracle import jet_ref as J
from tests.wave_model import Model

CASES = [
    # d, hidden, layers, act, second-order pairs
    (2, 32, 2, "tanh", [(0, 0), (1, 1)]),
    (2, 32, 2, "tanh", [(0, 0), (0, 1), (1, 1)]),
    (1, 32, 2, "sin", []),
    (2, 64, 3, "tanh", [(0, 0)]),
    (2, 32, 2, "swish", [(0, 0), (0, 1), (1, 1)]),     # the kernels' closed forms in (t, c) vs Leibniz on z * sigma(z)
    (1, 32, 2, "sigmoid", [(0, 0)]),
    (2, 32, 2, "aptx", [(0, 0), (1, 1)]),
]


@pytest.mark.parametrize("d,hidden,layers,act,pairs", CASES)
def test_wave_model_matches_jet_oracle(d, hidden, layers, act, pairs):
    rng = np.random.default_rng(3)
    dims = (d,) + (hidden,) * layers + (1,)
    npar = sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))
    flat = rng.standard_normal(npar) * 0.3
    x = rng.uniform(-1, 1, size=(d, 16))
    m = Model(flat, d, hidden, layers, act, pairs)
    streams = [()] + [(a,) for a in range(d)] + list(pairs)
    want = J.mlp_jets(flat, dims, act, list(x), streams)
    got = m.forward(x)
    for s, key in enumerate(streams):
        assert np.allclose(got[s], want[key][:, 0], rtol=1e-11, atol=1e-12), key
    gout = rng.standard_normal((len(streams), 16))
    gbar = {key: gout[s][:, None] for s, key in enumerate(streams)}
    wantg = J.mlp_jets_vjp(flat, dims, act, list(x), gbar)
    gotg = m.backward(x, gout)
    assert np.linalg.norm(gotg - wantg) <= 1e-11 * np.linalg.norm(wantg)
