"""Edge batches on the fused path and on plain torch modules (``fused="off"``) from one seed: 1 / 15 / 17 points, no points at all (a
FilterGenerator that kept nothing), coordinates far outside the domain, non-finite coordinates.  Wherever the reference produces
inf / nan (solvers.py:369-395 has no guard: a mean over nothing, an overflowing re-parameterisation) the fused path must produce
the same pattern instead of raising -- the first-use self-check of a closure kernel cannot compare nan with nan and stays
inconclusive on such a batch (engine.verify_fused)."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "one_point": lambda: torch.tensor([0.7]),
    "fifteen": lambda: torch.linspace(0.0, 2.0, 15),
    "seventeen": lambda: torch.linspace(0.0, 2.0, 17),
    "empty": lambda: torch.zeros(0),
    "huge": lambda: torch.tensor([0.0, 1.0, 1e6, 1e12, 1e30]),
    "tiny": lambda: torch.tensor([0.0, 1e-30, 1e-38, 1e-44, -1e-20]),
    "overflowing_decay": lambda: torch.linspace(-50.0, -1.0, 33),
    "with_nan": lambda: torch.tensor([0.0, 1.0, float("nan"), 2.0]),
    "with_inf": lambda: torch.tensor([0.0, 1.0, float("inf"), 2.0]),
    "repeated": lambda: torch.full((40,), 1.25),
}


def _run(fused, pts):
    from neurodiffeq_amd import autograd_ops, diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.generators import PredefinedGenerator
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver1D
    torch.manual_seed(2)
    s = Solver1D(lambda u, t: [diff(u, t, order=2) + u * diff(u, t) - torch.sin(t)], [IVP(0.0, 1.0, 0.5)],
                 nets=[FCNN(1, 1, hidden_units=(32, 32)).cuda()], train_generator=PredefinedGenerator(pts.clone()),
                 valid_generator=PredefinedGenerator(pts.clone()))
    s.fused = fused
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if fused == "off":
            with autograd_ops.native_autograd(False):
                s.fit(4)
        else:
            s.fit(4)
    flat = torch.cat([p.detach().reshape(-1) for p in s.nets[0].parameters()]).double().cpu().numpy()
    return np.array(s.metrics_history["train_loss"]), np.array(s.metrics_history["valid_loss"]), flat


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_batch_trains_like_plain_torch(name):
    fused, plain = _run("auto", CASES[name]()), _run("off", CASES[name]())
    for a, b, what in zip(fused, plain, ("train loss", "valid loss", "parameters")):
        assert a.shape == b.shape, what
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), (what, a, b)
        fin = np.isfinite(a) & np.isfinite(b)
        if fin.any():
            scale = np.linalg.norm(b[fin]) if what == "parameters" else np.abs(b[fin])
            err = np.linalg.norm(a[fin] - b[fin]) / max(scale, 1e-30) if what == "parameters" else np.max(np.abs(a[fin] - b[fin]) / np.maximum(scale, 1e-30))
            assert err < 5e-5, (what, err, a, b)
