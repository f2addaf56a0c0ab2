"""The API mirror's composite path (the reference's closure on torch autograd, what runs when a system is outside the
fused scope or no GPU is present) against the reference's own golden trajectories: same seeds -> same initial
parameters, same sampled batches, same three Adam epochs.  CPU only."""
import os

import numpy as np
import pytest
import torch

from tests import configs

SIZES = {"c1": 64, "c2": 16, "c3": 12, "c4": 96, "c5": 8, "w1": None, "w2": None, "w3": None, "w4": None, "w5": None, "w6": None, "w7": None, "w8": None, "w11": None, "w12": None, "w13": None, "w14": None, "w15": None, "w16": None, "w17": None, "w18": None, "w19": None, "w20": None, "w21": None}


def _flat(nets):
    return torch.cat([p.detach().reshape(-1) for n in nets for p in n.parameters()]).cpu().numpy()


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5", "w1", "w2", "w3", "w4", "w5", "w6", "w7", "w8", "w11", "w12", "w13", "w14", "w15", "w16", "w17", "w18", "w19", "w20", "w21"])
def test_composite_solver_trajectory_matches_reference(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    solver, cfg = configs.make_solver(name, SIZES[name])
    solver.fused = "off"
    for n in cfg["nets"]:
        n.to("cpu")
    solver.device = torch.device("cpu")
    assert np.array_equal(_flat(cfg["nets"]), gold["params0"])                      # default init, bit for bit
    torch.manual_seed(int(gold["seed"]) + 2)
    for _ in range(3):
        solver.run_train_epoch()
    assert not solver.fused_active
    hist = np.array(solver.metrics_history["train_loss"])
    assert np.allclose(hist, gold["traj_loss"], rtol=2e-6), (hist, gold["traj_loss"])
    p, want = _flat(cfg["nets"]), gold["traj_params"]
    assert np.linalg.norm(p - want) <= 2e-6 * np.linalg.norm(want)
