"""Pin the oracles (oracle/*.py) to the golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import autograd_ref as R
from oracle import jet_ref as J

SIZES = {"c1": 64, "c2": 16, "c3": 12, "c5": 8, "c4": 96}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"))


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c5", "c4"])
def test_generator_draws_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    torch.manual_seed(int(g["seed"]))
    cfg = R.build_config(name, SIZES[name])          # consumes the RNG exactly like the reference's net init
    torch.manual_seed(int(g["seed"]) + 1)
    d1 = np.stack([c.numpy() for c in cfg["sampler"]()])
    d2 = np.stack([c.numpy() for c in cfg["sampler"]()])
    assert np.array_equal(d1, g["coords"])
    assert np.array_equal(d2, g["draw2"])


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c5", "c4"])
def test_default_init_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    torch.manual_seed(int(g["seed"]))
    cfg = R.build_config(name, SIZES[name])
    assert np.array_equal(R.get_flat(cfg["nets"]).numpy(), g["params0"])


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c5", "c4"])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_closure_matches_reference(golden_dir, name, prec):
    g = _load(golden_dir, name)
    dt = torch.float64 if prec == "f64" else torch.float32
    cfg = R.build_config(name, SIZES[name], dtype=dt)
    R.set_flat(cfg["nets"], g["params0"])
    coords = [torch.from_numpy(c).to(dt) for c in g["coords"]]
    out = R.closure(cfg["nets"], cfg["enforcers"], cfg["pde"], coords)
    tol = 1e-12 if prec == "f64" else 2e-6
    assert rel_l2(out["funcs"].numpy(), g[f"funcs_{prec}"]) < tol
    assert rel_l2(out["residuals"].numpy(), g[f"residuals_{prec}"]) < tol
    assert abs(out["loss"].item() - float(g[f"loss_{prec}"])) <= tol * abs(float(g[f"loss_{prec}"]))
    assert rel_l2(R.get_flat_grad(cfg["nets"]).numpy(), g[f"grad_{prec}"]) < tol


@pytest.mark.parametrize("name,size", [("c1", 1024), ("c2", 256), ("c4", 131072), ("c3", 512), ("c5", 1024)])
def test_chunked_closure_matches_reference_at_stated_size(golden_dir, name, size):
    """The chunked oracle walk the at-size GPU parity tests rely on (autograd_ref.closure_chunked), against what the
    unmodified reference produced at the BASELINE size (<name>_full.npz); the batch comes from the port's generators
    under the same seed (C5: 1 048 576 points, ~20 s here)."""
    g = _load(golden_dir, f"{name}_full")
    torch.manual_seed(int(g["seed"]))
    cfg = R.build_config(name, size, dtype=torch.float64)
    R.set_flat(cfg["nets"], torch.from_numpy(g["params0"]).double())
    sampler = R.build_config(name, size)["sampler"]           # the draw is fp32, like the reference's generators
    torch.manual_seed(int(g["seed"]) + 1)
    coords = sampler()
    # same draw as the reference's (checksum budget: 1/8 ulp per value -- the spherical generator's transcendental CPU
    # kernels may differ in the last bit between host ISAs; a different draw is off by ~1e6 ulp per value)
    assert np.allclose(np.stack([c[:8].numpy() for c in coords]), g["coords_head"], rtol=3e-7, atol=0)
    bits = np.asarray([c.view(torch.int32).to(torch.int64).sum().item() for c in coords])
    assert np.all(np.abs(bits - g["coords_bits_sum"]) <= coords[0].numel() // 8), (bits, g["coords_bits_sum"])
    out = R.closure_chunked(cfg["nets"], cfg["enforcers"], cfg["pde"], [c.double() for c in coords], chunk=16384, keep=True)
    assert abs(out["loss"].item() - float(g["loss_f64"])) <= 1e-11 * abs(float(g["loss_f64"]))
    assert rel_l2(R.get_flat_grad(cfg["nets"]).numpy(), g["grad_f64"]) < 1e-11
    assert rel_l2(out["funcs"].sum(dim=0).numpy(), g["funcs_sum"]) < 1e-11
    assert rel_l2((out["residuals"] ** 2).sum(dim=0).numpy(), g["resid_sq_sum"]) < 1e-11


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_adam_trajectory_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    torch.manual_seed(int(g["seed"]))
    cfg = R.build_config(name, SIZES[name])
    loop = R.TrainLoop(cfg["nets"], cfg["enforcers"], cfg["pde"], cfg["sampler"])
    torch.manual_seed(int(g["seed"]) + 2)
    for _ in range(3):
        loop.epoch()
    assert np.allclose(loop.history, g["traj_loss"], rtol=2e-5)
    assert rel_l2(R.get_flat(cfg["nets"]).numpy(), g["traj_params"]) < 1e-5


def test_diff_known_answers(golden_dir):
    g = _load(golden_dir, "diff_known")
    x, y, z = [torch.tensor(g[k], requires_grad=True) for k in "xyz"]
    u = torch.sin(x) * torch.exp(y) + x * y * z ** 3
    assert np.allclose(R.ref_diff(u, x).detach().numpy(), g["u_x"], rtol=1e-12)
    assert np.allclose(R.ref_diff(u, x, 2).detach().numpy(), g["u_xx"], rtol=1e-12)
    assert np.allclose(R.ref_diff(R.ref_diff(u, y), z).detach().numpy(), g["u_yz"], rtol=1e-12)
    t = torch.tensor(g["t"], requires_grad=True)
    for k in range(1, 5):
        assert np.allclose(R.ref_diff(torch.exp(2 * t), t, k).detach().numpy(), g[f"exp_d{k}"], rtol=1e-12)
        assert np.allclose(R.ref_diff(t ** 2, t, k).detach().numpy(), g[f"sq_d{k}"], atol=1e-12)
    with pytest.raises(ValueError):
        R.ref_diff(u.reshape(-1), x)


# ---------------------------------------------------------------- jet oracle vs autograd oracle
ARCH = {"c1": ((1, 32, 32, 1), "sin"), "c2": ((2, 32, 32, 1), "tanh"), "c3": ((2, 64, 64, 64, 1), "tanh")}


@pytest.mark.parametrize("name", ["c1", "c2", "c3"])
def test_jet_streams_match_autograd(golden_dir, name):
    g = _load(golden_dir, name)
    dims, act = ARCH[name]
    npar = sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))
    flat = g["params0"][:npar].astype(np.float64)
    net = R.make_fcnn(dims[0], dims[-1], dims[1:-1], act, torch.float64)
    R.set_flat([net], flat)
    cs = [torch.tensor(c, dtype=torch.float64).reshape(-1, 1).requires_grad_(True) for c in g["coords"]]
    out = net(torch.cat(cs, dim=1))
    d = len(cs)
    streams = [(a,) for a in range(d)] + [(a, b) for a in range(d) for b in range(a, d)]
    jets = J.mlp_jets(flat, dims, act, [c.detach().numpy() for c in cs], streams)
    assert np.allclose(jets[()], out.detach().numpy(), rtol=1e-12, atol=1e-14)
    for m in streams:
        ref = out
        for a in m:
            ref = R.ref_diff(ref, cs[a])
        assert np.allclose(jets[m], ref.detach().numpy(), rtol=1e-10, atol=1e-12), m


@pytest.mark.parametrize("name", ["c1", "c2", "c3"])
def test_jet_vjp_matches_autograd(golden_dir, name):
    g = _load(golden_dir, name)
    dims, act = ARCH[name]
    npar = sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))
    flat = g["params0"][:npar].astype(np.float64)
    net = R.make_fcnn(dims[0], dims[-1], dims[1:-1], act, torch.float64)
    R.set_flat([net], flat)
    cs = [torch.tensor(c, dtype=torch.float64).reshape(-1, 1).requires_grad_(True) for c in g["coords"]]
    out = net(torch.cat(cs, dim=1))
    d = len(cs)
    streams = [()] + [(a,) for a in range(d)] + [(a, b) for a in range(d) for b in range(a, d)]
    rng = np.random.default_rng(0)
    gbar = {m: rng.standard_normal(out.shape) for m in streams}
    total = 0
    for m in streams:
        ref = out
        for a in m:
            ref = R.ref_diff(ref, cs[a])
        total = total + (ref * torch.tensor(gbar[m])).sum()
    total.backward()
    want = R.get_flat_grad([net]).numpy()
    got = J.mlp_jets_vjp(flat, dims, act, [c.detach().numpy() for c in cs], gbar)
    assert rel_l2(got, want) < 1e-11


@pytest.mark.parametrize("act", ["tanh", "sin", "sigmoid"])
def test_third_order_jets_match_nested_autograd(act):
    """The jet oracle's third-order recurrences (forward AND the adjoint with the fourth activation derivative) against
    three nested autograd sweeps (ref_diff, neurodiffeq.py:21-34) in fp64."""
    from oracle import jet_ref as J
    torch.manual_seed(0)
    net = R.make_fcnn(2, 2, (16, 16), act, torch.float64)
    flat = R.get_flat([net]).numpy()
    x, y = [torch.rand(7, 1, dtype=torch.float64, requires_grad=True) for _ in range(2)]
    out = net(torch.cat([x, y], 1))
    d = R.ref_diff
    z = J.mlp_jets(flat, (2, 16, 16, 2), act, [x.detach().numpy(), y.detach().numpy()], [(0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)])
    rng = np.random.default_rng(0)
    gb = {m: rng.standard_normal((7, 2)) for m in z}
    loss = 0
    for o in range(2):
        u = out[:, o:o + 1]
        cols = {(): u, (0,): d(u, x), (1,): d(u, y), (0, 0): d(u, x, 2), (0, 1): d(d(u, x), y), (1, 1): d(u, y, 2),
                (0, 0, 0): d(u, x, 3), (0, 0, 1): d(d(u, x, 2), y), (0, 1, 1): d(d(u, y, 2), x), (1, 1, 1): d(u, y, 3)}
        for m in z:
            assert rel_l2(z[m][:, o], cols[m].detach().numpy()) < 1e-10, (act, m)
            loss = loss + (torch.from_numpy(gb[m][:, o:o + 1]) * cols[m]).sum()
    loss.backward()
    got = J.mlp_jets_vjp(flat, (2, 16, 16, 2), act, [x.detach().numpy(), y.detach().numpy()], gb)
    assert rel_l2(got, R.get_flat_grad([net]).numpy()) < 1e-12


@pytest.mark.parametrize("act", ["tanh", "sin", "sigmoid"])
def test_fourth_order_jets_match_nested_autograd(act):
    """Round 6: the fourth-order recurrences (general Faa di Bruno over the set partitions of the index positions, forward
    AND the adjoint with the fifth activation derivative) against four nested autograd sweeps in fp64 -- all five
    quadruples of two coordinates (xxxx, xxxy, xxyy, xyyy, yyyy) with every sub-stream they need, two outputs."""
    from oracle import jet_ref as J
    torch.manual_seed(0)
    net = R.make_fcnn(2, 2, (12, 12), act, torch.float64)
    flat = R.get_flat([net]).numpy()
    x, y = [torch.rand(6, 1, dtype=torch.float64, requires_grad=True) for _ in range(2)]
    out = net(torch.cat([x, y], 1))
    d = R.ref_diff
    quads = [(0, 0, 0, 0), (0, 0, 0, 1), (0, 0, 1, 1), (0, 1, 1, 1), (1, 1, 1, 1)]
    z = J.mlp_jets(flat, (2, 12, 12, 2), act, [x.detach().numpy(), y.detach().numpy()], quads)
    assert len(z) == 1 + 2 + 3 + 4 + 5
    rng = np.random.default_rng(0)
    gb = {m: rng.standard_normal((6, 2)) for m in z}
    loss = 0
    coords = (x, y)
    for o in range(2):
        u = out[:, o:o + 1]
        for m in z:
            col = u
            for a in m:
                col = d(col, coords[a])
            assert rel_l2(z[m][:, o], col.detach().numpy()) < 1e-10, (act, m)
            loss = loss + (torch.from_numpy(gb[m][:, o:o + 1]) * col).sum()
    loss.backward()
    got = J.mlp_jets_vjp(flat, (2, 12, 12, 2), act, [x.detach().numpy(), y.detach().numpy()], gb)
    assert rel_l2(got, R.get_flat_grad([net]).numpy()) < 1e-11
    # the beam case: one coordinate, pure fourth derivative -- exactly the streams (), (0,), (0,0), (0,0,0), (0,0,0,0)
    assert J.close_streams([(0, 0, 0, 0)]) == [(), (0,), (0, 0), (0, 0, 0), (0, 0, 0, 0)]


@pytest.mark.parametrize("name,grid", [("c2", 64), ("c3", 48)])
def test_closure_at_the_reference_trained_state(golden_dir, name, grid):
    """Near convergence (tests/golden/<name>_trained.npz: the unmodified reference trained the config, then evaluated one
    batch in fp64 and fp32): the oracle reproduces the reference's fp64 closure there as well -- the residual is a
    cancellation of O(1) terms at this point."""
    g = np.load(os.path.join(golden_dir, f"{name}_trained.npz"))
    cfg = R.build_config(name, grid, dtype=torch.float64)
    R.set_flat(cfg["nets"], torch.from_numpy(g["params"]).double())
    coords = [torch.from_numpy(c).double() for c in g["coords"]]
    out = R.closure(cfg["nets"], cfg["enforcers"], cfg["pde"], coords)
    assert rel_l2(out["funcs"].numpy().ravel(), g["u_f64"]) < 1e-12
    assert rel_l2(out["residuals"].numpy(), g["residual_f64"]) < 1e-9         # (relative to a residual ~1e-2 of its terms)
    assert abs(out["loss"].item() - float(g["loss_f64"])) <= 1e-9 * float(g["loss_f64"])
    assert rel_l2(R.get_flat_grad(cfg["nets"]).numpy(), g["grad_f64"]) < 1e-9
    # how far the reference's own fp32 evaluation is from fp64 here: the yardstick of the GPU test
    assert rel_l2(g["residual_f32"], g["residual_f64"]) > 1e-7
