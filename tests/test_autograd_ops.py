"""The torch custom ops + autograd.Function seam (neurodiffeq_amd/autograd_ops.py): ``cond.enforce`` -> ``diff`` ->
``loss.backward()`` -> ``torch.optim`` written by hand, outside any Solver.

CPU (here): the autograd plumbing -- streams as extra outputs, input gradients as differentiable expressions of those
outputs, one adjoint call for the parameter gradients -- is exercised end to end with the two dispatcher ops backed
by the numpy jet oracle (registered as CPU kernels by this test only), against the reference's golden trajectory.
GPU (``-m gpu``): the same loop on the real ``ndq_mlp_jet_fwd`` / ``ndq_mlp_jet_bwd`` kernels."""
import math
import os

import numpy as np
import pytest
import torch

from neurodiffeq_amd import autograd_ops, diff
from neurodiffeq_amd.networks import FCNN
from oracle import autograd_ref as R
from oracle import jet_ref as J
from tests import configs

ACT = {0: "tanh", 1: "sin", 2: "sigmoid", 3: "swish", 4: "aptx", 5: "elu", 6: "softplus", 7: "gelu"}


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def cpu_ops():
    """CPU kernels for torch.ops.ndq.* from the jet oracle (fp64 inside, fp32 in / out like the HIP kernels)."""
    def dims(d, hidden, layers, n_out):
        return (d,) + (hidden,) * layers + (n_out,)

    def fwd(coords, params, n, order, hidden, layers, act, n_out):
        d = coords.shape[0]
        streams = autograd_ops._streams(d, order)
        z = J.mlp_jets(params.numpy(), dims(d, hidden, layers, n_out), ACT[act], [coords[a, :n].numpy() for a in range(d)],
                       streams)
        jets = torch.zeros(len(streams) * n_out, coords.shape[1], dtype=coords.dtype)
        for s, mi in enumerate(streams):
            jets[s * n_out:(s + 1) * n_out, :n] = torch.from_numpy(z[mi].T.copy()).to(coords.dtype)
        return jets

    def bwd(coords, params, gbar, n, order, hidden, layers, act, n_out):
        d = coords.shape[0]
        streams = autograd_ops._streams(d, order)
        g = {mi: gbar[s * n_out:(s + 1) * n_out, :n].numpy().T.astype(np.float64) for s, mi in enumerate(streams)}
        out = J.mlp_jets_vjp(params.numpy().astype(np.float64), dims(d, hidden, layers, n_out), ACT[act],
                             [coords[a, :n].numpy() for a in range(d)], g)
        return torch.from_numpy(out).to(coords.dtype)

    autograd_ops.mlp_jet_fwd.register_kernel("cpu")(fwd)
    autograd_ops.mlp_jet_bwd.register_kernel("cpu")(bwd)
    old = autograd_ops._DEVICE_TYPES
    autograd_ops._DEVICE_TYPES = ("cuda", "cpu")
    import neurodiffeq_amd.codegen as codegen
    keep = codegen.ensure_mlp_kernels
    codegen.ensure_mlp_kernels = lambda desc, f64=False: True
    autograd_ops._SPECS.clear()
    yield
    autograd_ops._DEVICE_TYPES = old
    codegen.ensure_mlp_kernels = keep
    autograd_ops._SPECS.clear()


def hand_written_epochs(name, size, device, epochs=3):
    """The reference's closure (solvers.py:369-395) written by hand on the public API, default Adam(1e-3)."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
    torch.manual_seed(int(gold["seed"]))
    cfg = configs.make(name, size)
    nets = [n.to(device) for n in cfg["nets"]]
    assert np.array_equal(R.get_flat(nets).cpu().numpy(), gold["params0"])
    opt = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=1e-3)
    torch.manual_seed(int(gold["seed"]) + 2)
    losses = []
    for _ in range(epochs):
        ex = cfg["gen"].get_examples()
        coords = [c.detach().reshape(-1, 1).to(device).requires_grad_(True) for c in ([ex] if isinstance(ex, torch.Tensor) else ex)]
        funcs = [cond.enforce(net, *coords) for net, cond in zip(nets, cfg["conds"])]
        res = torch.cat(cfg["pde"](*funcs, *coords), dim=1)
        loss = (res ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return gold, np.array(losses), R.get_flat(nets).cpu().numpy(), nets


@pytest.mark.parametrize("name,size", [("c2", 16), ("c1", 64), ("c3", 12)])
def test_hand_written_loop_matches_reference_trajectory_cpu_plumbing(cpu_ops, name, size):
    calls = {"fwd": 0}
    orig = autograd_ops.MlpJet.forward
    gold, losses, params, _ = hand_written_epochs(name, size, "cpu")
    assert np.max(np.abs(losses - gold["traj_loss"]) / np.abs(gold["traj_loss"])) < 2e-5, (losses, gold["traj_loss"])
    assert rel_l2(params, gold["traj_params"]) < 1e-5


def test_streams_serve_diff_and_mixed_second_derivatives(cpu_ops):
    torch.manual_seed(3)
    net = FCNN(3, 2, hidden_units=(32, 32))
    x, y, z = [torch.rand(11, 1, requires_grad=True) for _ in range(3)]
    used = []
    orig = autograd_ops.MlpJet.apply
    out = net(torch.cat([x, y, z], dim=1))
    assert out.grad_fn is not None and "MlpJet" in type(out.grad_fn).__name__
    u = (out[:, :1] * torch.sin(x) + out[:, 1:] * y * z)
    got = dict(ux=diff(u, x), uyz=diff(diff(u, y), z), uxx=diff(u, x, order=2), uzy=diff(diff(u, z), y))
    autograd_ops.set_native_autograd(False)
    try:
        out2 = net(torch.cat([x, y, z], dim=1))
        assert "MlpJet" not in type(out2.grad_fn).__name__
        u2 = (out2[:, :1] * torch.sin(x) + out2[:, 1:] * y * z)
        want = dict(ux=diff(u2, x), uyz=diff(diff(u2, y), z), uxx=diff(u2, x, order=2), uzy=diff(diff(u2, z), y))
    finally:
        autograd_ops.set_native_autograd(True)
    for k in got:
        assert rel_l2(got[k].detach().numpy(), want[k].detach().numpy()) < 2e-6, k
    # parameter gradients of a loss built from second derivatives: one adjoint call, same numbers as torch autograd
    loss = (got["uxx"] ** 2 + got["uyz"] ** 2 + u ** 2).mean()
    loss2 = (want["uxx"] ** 2 + want["uyz"] ** 2 + u2 ** 2).mean()
    g1 = torch.autograd.grad(loss, list(net.parameters()))
    g2 = torch.autograd.grad(loss2, list(net.parameters()))
    assert rel_l2(torch.cat([g.reshape(-1) for g in g1]).numpy(), torch.cat([g.reshape(-1) for g in g2]).numpy()) < 2e-6


def test_third_order_request_raises_and_plain_inputs_fall_back(cpu_ops):
    net = FCNN(1, 1)
    t = torch.linspace(0, 1, 9).reshape(-1, 1).requires_grad_(True)
    u = net(t)
    with pytest.raises(RuntimeError, match="order 3"):
        diff(u, t, order=3)
    # no gradient wanted: value-only stream set, still the custom op
    with torch.no_grad():
        v = net(t)
    assert torch.allclose(v, net.NN(t), atol=1e-6)
    # fp64 networks have kernels of their own (libndq64.so); mixed precision is the plain Sequential's business
    net64 = FCNN(1, 1).double()
    out = net64(t.double())
    assert "MlpJet" in type(out.grad_fn).__name__
    with pytest.raises(RuntimeError):
        net64(t)            # fp32 input into an fp64 network: torch's own dtype error, not a kernel launch


def test_fp64_closure_matches_reference_golden_cpu_plumbing(cpu_ops):
    """The reference's default precision through the same seam: one closure of C2 and C1 in fp64 against the golden
    fp64 vectors of the unmodified reference (1e-10: the oracle-backed CPU ops compute in fp64)."""
    _fp64_closure("cpu", 1e-10)


def _fp64_closure(device, tol):
    for name, size in (("c2", 16), ("c1", 64)):
        gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
        torch.manual_seed(0)
        cfg = configs.make(name, size)
        nets = [n.double().to(device) for n in cfg["nets"]]
        R.set_flat(nets, torch.from_numpy(gold["params0"]).double().to(device))
        coords = [torch.from_numpy(c).double().reshape(-1, 1).to(device).requires_grad_(True) for c in gold["coords"]]
        funcs = [cond.enforce(net, *coords) for net, cond in zip(nets, cfg["conds"])]
        assert all("MlpJet" in type(n(torch.cat(coords, 1)).grad_fn).__name__ for n in nets)
        res = torch.cat(cfg["pde"](*funcs, *coords), dim=1)
        loss = (res ** 2).mean()
        loss.backward()
        grad = torch.cat([p.grad.reshape(-1) for n in nets for p in n.parameters()]).cpu().numpy()
        assert rel_l2(torch.cat(funcs, 1).detach().cpu().numpy(), gold["funcs_f64"]) < tol
        assert rel_l2(res.detach().cpu().numpy(), gold["residuals_f64"]) < tol
        assert abs(loss.item() - float(gold["loss_f64"])) <= tol * abs(float(gold["loss_f64"]))
        assert rel_l2(grad, gold["grad_f64"]) < tol, (name, rel_l2(grad, gold["grad_f64"]))


def test_third_order_on_request(cpu_ops):
    """set_native_autograd(max_order=3): every third-order partial travels with the forward launch; diff(order=3) and
    the parameter gradient of a loss built on it equal torch autograd through the plain Sequential."""
    torch.manual_seed(5)
    net = FCNN(2, 1, hidden_units=(32, 32))
    x, y = [torch.rand(9, 1, requires_grad=True) for _ in range(2)]

    def build():
        u = net(torch.cat([x, y], dim=1)) * (1.0 + x * y)
        return diff(u, x, order=3) + diff(diff(u, y, order=2), x) + u
    with autograd_ops.native_autograd(True, max_order=3):
        r = build()
        assert "MlpJet" in type(net(torch.cat([x, y], dim=1)).grad_fn).__name__
        g1 = torch.autograd.grad((r ** 2).mean(), list(net.parameters()))
    with autograd_ops.native_autograd(False):
        r2 = build()
        g2 = torch.autograd.grad((r2 ** 2).mean(), list(net.parameters()))
    assert rel_l2(r.detach().numpy(), r2.detach().numpy()) < 2e-6
    assert rel_l2(torch.cat([g.reshape(-1) for g in g1]).numpy(), torch.cat([g.reshape(-1) for g in g2]).numpy()) < 2e-6


def test_fourth_order_on_request(cpu_ops):
    """set_native_autograd(max_order=4) (round 6): every fourth-order partial of a one- or two-input network travels with the
    forward launch; the biharmonic operator written with diff() and the parameter gradient of a loss built on it equal
    torch autograd through the plain Sequential."""
    torch.manual_seed(6)
    net = FCNN(2, 1, hidden_units=(16, 16))
    x, y = [torch.rand(7, 1, requires_grad=True) for _ in range(2)]

    def build():
        u = net(torch.cat([x, y], dim=1)) * (1.0 + x * y)
        return diff(u, x, order=4) + 2.0 * diff(diff(u, x, order=2), y, order=2) + diff(u, y, order=4) + u
    with autograd_ops.native_autograd(True, max_order=4):
        r = build()
        assert "MlpJet" in type(net(torch.cat([x, y], dim=1)).grad_fn).__name__
        g1 = torch.autograd.grad((r ** 2).mean(), list(net.parameters()))
    with autograd_ops.native_autograd(False):
        r2 = build()
        g2 = torch.autograd.grad((r2 ** 2).mean(), list(net.parameters()))
    assert rel_l2(r.detach().numpy(), r2.detach().numpy()) < 5e-6
    assert rel_l2(torch.cat([g.reshape(-1) for g in g1]).numpy(), torch.cat([g.reshape(-1) for g in g2]).numpy()) < 5e-6
    assert autograd_ops._spec_for(FCNN(3, 1, hidden_units=(16, 16)), 4) is None       # three inputs: order 4 stays on plain torch


@pytest.mark.gpu
def test_fourth_order_on_the_hip_kernels():
    """The same biharmonic expression on the MI355X: forward + nested diff() sweeps served by the fourth-order stream kernels
    (an extension module: every partial up to order four of a 2 -> 32 -> 32 -> 1 tanh network, 15 streams), the parameter
    gradient by ONE adjoint launch; against plain torch autograd on the same device."""
    torch.manual_seed(6)
    net = FCNN(2, 1, hidden_units=(32, 32)).to("cuda")
    x, y = [torch.rand(500, 1, device="cuda", requires_grad=True) for _ in range(2)]

    def build():
        u = net(torch.cat([x, y], dim=1)) * (1.0 + x * y)
        return diff(u, x, order=4) + 2.0 * diff(diff(u, x, order=2), y, order=2) + diff(u, y, order=4) + u
    with autograd_ops.native_autograd(True, max_order=4):
        r = build()
        assert "MlpJet" in type(net(torch.cat([x, y], dim=1)).grad_fn).__name__
        g1 = torch.autograd.grad((r ** 2).mean(), list(net.parameters()))
    with autograd_ops.native_autograd(False):
        r2 = build()
        g2 = torch.autograd.grad((r2 ** 2).mean(), list(net.parameters()))
    assert rel_l2(r.detach().cpu().numpy(), r2.detach().cpu().numpy()) < 1e-5
    assert rel_l2(torch.cat([g.reshape(-1) for g in g1]).cpu().numpy(), torch.cat([g.reshape(-1) for g in g2]).cpu().numpy()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("coordinate_grads", [False, True])
@pytest.mark.parametrize("name,size", [("c2", 16), ("c1", 64), ("c3", 12), ("c4", 96)])
def test_hand_written_loop_on_the_hip_kernels_matches_reference_trajectory(name, size, coordinate_grads):
    """enforce -> diff -> .backward() -> torch.optim.Adam, three epochs, HIP forward / adjoint
    kernels underneath (asserted through the dispatcher ops' call counts), against tests/golden/<name>.npz.
    ``coordinate_grads`` (round 3): with the default (True) ``loss.backward()`` also fills the ``.grad`` of the sampled
    coordinates like the reference does -- for a second-order residual that is one more forward launch (third-order
    streams) per step; False is what a Solver's composite path runs with."""
    if name == "c4":
        pytest.skip("C4's enforcer is a SolverSpherical hook; covered by the closure test below")
    calls = {"fwd": 0, "bwd": 0}
    L = autograd_ops._lib.lib()
    fwd, bwd = L.ndq_mlp_jet_fwd, L.ndq_mlp_jet_bwd

    class Counting:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)
    L.ndq_mlp_jet_fwd, L.ndq_mlp_jet_bwd = Counting(fwd, "fwd"), Counting(bwd, "bwd")
    try:
        with autograd_ops.native_autograd(True, coordinate_grads=coordinate_grads):
            gold, losses, params, nets = hand_written_epochs(name, size, "cuda")
    finally:
        L.ndq_mlp_jet_fwd, L.ndq_mlp_jet_bwd = fwd, bwd
    second_order = name in ("c2", "c3")
    assert calls["fwd"] == 3 * len(nets) * (2 if (coordinate_grads and second_order) else 1), calls
    assert calls["bwd"] == 3 * len(nets), calls
    assert np.max(np.abs(losses - gold["traj_loss"]) / np.abs(gold["traj_loss"])) < 2e-5, (losses, gold["traj_loss"])
    assert rel_l2(params, gold["traj_params"]) < 1e-5


@pytest.mark.gpu
def test_fp64_closure_on_the_f64_mfma_kernels_matches_reference_golden():
    """fp64 networks (the reference's default precision, neurodiffeq/__init__.py:22) on libndq64.so: the stream kernels
    compiled for double (v_mfma_f64_16x16x4_f64).  One closure of C2 and C1 against the reference's golden fp64 vectors."""
    _fp64_closure("cuda", 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("d,act,n_out,order", [(1, "tanh", 1, 2), (2, "tanh", 1, 2), (1, "sin", 1, 2), (2, "tanh", 3, 1),
                                               (3, "tanh", 1, 2), (1, "sigmoid", 1, 3), (2, "sin", 1, 0)])
def test_fp64_stream_kernels_match_jet_oracle(d, act, n_out, order):
    """ndq64_mlp_jet_fwd / ndq64_mlp_jet_bwd through the dispatcher ops against the numpy jet oracle, ragged batch."""
    from neurodiffeq_amd import codegen
    act_id = {"tanh": 0, "sin": 1, "sigmoid": 2}[act]
    desc = autograd_ops._desc(d, order, 32, 2, act_id, n_out)
    assert codegen.ensure_mlp_kernels(desc, f64=True)
    rng = np.random.default_rng(d * 100 + order)
    dims = (d, 32, 32, n_out)
    flat = np.concatenate([rng.uniform(-1, 1, a * b + b) / np.sqrt(a) for a, b in zip(dims[:-1], dims[1:])])
    n, ld = 1003, 1024
    coords = np.zeros((d, ld)); coords[:, :n] = rng.uniform(-1, 1, (d, n))
    streams = autograd_ops._streams(d, order)
    jets = torch.ops.ndq.mlp_jet_fwd(torch.from_numpy(coords).cuda(), torch.from_numpy(flat).cuda(), n, order, 32, 2, act_id, n_out)
    want = J.mlp_jets(flat, dims, act, list(coords[:, :n]), streams)
    for s, mi in enumerate(streams):
        assert rel_l2(jets[s * n_out:(s + 1) * n_out, :n].T.cpu().numpy(), want[mi]) < 1e-12, mi
    gbar = np.zeros((len(streams) * n_out, ld)); gbar[:, :n] = rng.standard_normal((len(streams) * n_out, n))
    grad = torch.ops.ndq.mlp_jet_bwd(torch.from_numpy(coords).cuda(), torch.from_numpy(flat).cuda(), torch.from_numpy(gbar).cuda(),
                                     n, order, 32, 2, act_id, n_out)
    gb = {mi: gbar[s * n_out:(s + 1) * n_out, :n].T for s, mi in enumerate(streams)}
    assert rel_l2(grad.cpu().numpy(), J.mlp_jets_vjp(flat, dims, act, list(coords[:, :n]), gb)) < 1e-11


@pytest.mark.gpu
def test_solver_in_the_reference_default_precision_trains_on_the_f64_kernels():
    """``set_tensor_type(device='cuda', float_bits=64)`` -- what importing the reference does (``__init__.py:22``) -- then a
    plain Solver2D, three ways: (a) the fused fp64 pipeline (traced pointwise kernel compiled in double between the fp64
    stream kernels of libndq64.so: engine.FusedSystem(dtype=float64)); (b) fused path off: the reference's closure with
    the networks' forward / backward on the fp64 stream kernels through the custom-op seam; (c) the same closure on
    plain torch fp64 autograd.  All three agree to 1e-9."""
    from neurodiffeq_amd.utils import set_tensor_type
    try:
        set_tensor_type(device="cuda", float_bits=64)
        runs = {}
        for mode in ("fused", "seam", "torch"):
            torch.manual_seed(0)
            solver, cfg = configs.make_solver("c2", 12)
            assert next(cfg["nets"][0].parameters()).dtype == torch.float64
            solver.fused = "require" if mode == "fused" else "off"
            torch.manual_seed(5)
            with autograd_ops.native_autograd(mode != "torch"):
                for _ in range(3):
                    solver.run_train_epoch()
                assert solver.fused_active == (mode == "fused")
                if mode == "fused":
                    assert solver._fused_sys.f64 and solver._fused_sys.fusedk is not None     # the closure kernel in double
                if mode == "seam":
                    ex = [c.detach().to("cuda").requires_grad_(True) for c in solver._generate_batch("train")]
                    assert "MlpJet" in type(cfg["nets"][0](torch.cat(ex, 1)).grad_fn).__name__
            runs[mode] = (np.array(solver.metrics_history["train_loss"]), R.get_flat(cfg["nets"]).cpu().numpy())
    finally:
        set_tensor_type(device="cpu", float_bits=32)
    for mode in ("fused", "seam"):
        assert np.allclose(runs[mode][0], runs["torch"][0], rtol=1e-9), (mode, runs[mode][0], runs["torch"][0])
        assert rel_l2(runs[mode][1], runs["torch"][1]) < 1e-9, mode


@pytest.mark.gpu
def test_multi_output_network_closure_on_the_hip_kernels_matches_golden():
    """C4's shape outside a solver: FCNN(1 -> 25) coefficient network, DirichletBVPSphericalBasis.enforce, harmonics,
    spherical_laplacian -- loss and parameter gradient of one closure against the reference's golden vectors."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4.npz"))
    torch.manual_seed(0)
    cfg = configs.make("c4", 96)
    net = cfg["nets"][0].to("cuda")
    R.set_flat([net], torch.from_numpy(gold["params0"]).cuda())
    coords = [torch.from_numpy(c).reshape(-1, 1).cuda().requires_grad_(True) for c in gold["coords"]]
    cond = cfg["conds"][0]
    cond.R_0, cond.R_1 = cond.R_0.cuda(), cond.R_1.cuda()     # what set_tensor_type('cuda') does for a user script
    u = cfg["enforcer"](net, cond, coords)
    assert "MlpJet" in type(net(coords[0]).grad_fn).__name__
    res = torch.cat(cfg["pde"](u, *coords), dim=1)
    loss = (res ** 2).mean()
    loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    assert abs(loss.item() - float(gold["loss_f64"])) / abs(float(gold["loss_f64"])) < 1e-5
    assert rel_l2(grad, gold["grad_f64"]) < 1e-5
    assert rel_l2(res.detach().cpu().numpy(), gold["residuals_f64"]) < 1e-5


def _residual(net, x, scale):
    """u = net(scale * x); r = u'' + u (second-order: the adjoint reaches the TOP-order stream of the forward launch)"""
    u = net(scale * x)
    return diff(u, x, order=2) + u


@pytest.mark.parametrize("act", [torch.nn.Tanh, torch.nn.Softplus, torch.nn.ReLU])
def test_gradients_flowing_through_the_network_input_are_never_dropped(cpu_ops, act):
    """a trainable tensor UPSTREAM of the network input (learnable input scale), and the coordinate
    gradient itself, must come out of loss.backward() exactly as torch autograd produces them -- the top-order stream's
    input gradient needs the streams one order up (tanh, Softplus: a forward launch with third-order streams; ReLU has no
    HIP kernels at all and runs on torch either way)."""
    torch.manual_seed(2)
    net = FCNN(1, 1, hidden_units=(32, 32), actv=act)
    x0 = torch.rand(13, 1)

    def run(native, how):
        x = x0.clone().requires_grad_(True)
        scale = torch.nn.Parameter(torch.tensor(1.3))
        for p in net.parameters():
            p.grad = None
        with autograd_ops.native_autograd(native):
            loss = (_residual(net, x, scale) ** 2).mean()
            if how == "backward":
                loss.backward()
                return [x.grad, scale.grad] + [p.grad for p in net.parameters()]
            if how == "params_only":
                return list(torch.autograd.grad(loss, list(net.parameters())))
            loss.backward(create_graph=True)            # parameter gradients that are differentiable themselves
            assert all(p.grad is not None for p in net.parameters())
            return [x.grad, scale.grad] + [p.grad for p in net.parameters()]

    for how in ("backward", "params_only", "create_graph"):
        got, want = run(True, how), run(False, how)
        assert all(g is not None for g in got), how
        for g, w in zip(got, want):
            assert rel_l2(g.detach().numpy(), w.detach().numpy()) < 5e-6, how


def test_coordinate_gradients_can_be_skipped_but_upstream_parameters_cannot(cpu_ops):
    """set_native_autograd(coordinate_grads=False): plain coordinate leaves keep .grad = None (no extra launch); with a
    trainable tensor upstream of the network input the gradient is produced regardless."""
    torch.manual_seed(4)
    net = FCNN(1, 1, hidden_units=(32, 32))
    x0 = torch.rand(9, 1)
    with autograd_ops.native_autograd(True, coordinate_grads=False):
        x = x0.clone().requires_grad_(True)
        ((diff(net(x), x, order=2)) ** 2).mean().backward()
        assert x.grad is None and all(p.grad is not None for p in net.parameters())
        x = x0.clone().requires_grad_(True)
        scale = torch.nn.Parameter(torch.tensor(0.7))
        (_residual(net, x, scale) ** 2).mean().backward()
        got = scale.grad.item()
    with autograd_ops.native_autograd(False):
        x = x0.clone().requires_grad_(True)
        scale = torch.nn.Parameter(torch.tensor(0.7))
        (_residual(net, x, scale) ** 2).mean().backward()
    assert abs(got - scale.grad.item()) <= 5e-6 * abs(scale.grad.item())
