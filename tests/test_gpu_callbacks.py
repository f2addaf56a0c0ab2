"""Callbacks that change the SOLVER between epochs (the reference runs them after every epoch and re-reads everything at the
next batch: solvers.py:369-395, 496-497).  Every scenario trains twice from the same seed -- on the fused MI355X path, and as
the reference does it (``fused="off"`` with the custom-op seam switched off: plain torch modules under torch autograd) -- with
the same callback firing at the same epochs; loss histories and final parameters have to agree to fp32 accuracy.  A change the
fused path did not notice shows as a trajectory that splits at the callback's epoch."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import autograd_ref as R

pytestmark = pytest.mark.gpu

EPOCHS, AT = 8, (3, 5)


def _solver(fused, kind="ode"):
    s = _make(kind)
    s.fused = fused
    return s


def _make(kind):
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP, DirichletBVP2D
    from neurodiffeq_amd.generators import Generator1D, Generator2D
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver1D, Solver2D
    torch.manual_seed(11)
    if kind == "ode":
        nets = [FCNN(1, 1, hidden_units=(32, 32)).cuda()]
        return Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], nets=nets,
                        train_generator=Generator1D(64, 0.0, 2.0, method="equally-spaced"),
                        valid_generator=Generator1D(32, 0.0, 2.0, method="equally-spaced"))
    if kind == "system":
        nets = [FCNN(1, 1, hidden_units=(32, 32)).cuda() for _ in range(2)]
        return Solver1D(lambda u, v, t: [diff(u, t) - (u - u * v), diff(v, t) - (u * v - v)], [IVP(0.0, 1.5), IVP(0.0, 1.0)], nets=nets,
                        train_generator=Generator1D(64, 0.1, 4.0, method="equally-spaced"),
                        valid_generator=Generator1D(32, 0.1, 4.0, method="equally-spaced"))
    zero = lambda v: 0 * v
    if kind in ("swish", "elu"):
        from functools import partial
        from neurodiffeq_amd.networks import Swish
        nets = [FCNN(2, 1, hidden_units=(32, 32), actv=partial(Swish, beta=1.3) if kind == "swish" else nn.ELU).cuda()]
    elif kind == "wide":          # layer-by-layer kernels (csrc/ndq_deep.h): 128 x 2
        nets = [FCNN(2, 1, hidden_units=(128, 128)).cuda()]
    elif kind == "resnet":
        from neurodiffeq_amd.networks import Resnet
        nets = [Resnet(2, 1, hidden_units=(32, 32)).cuda()]
    else:
        nets = [FCNN(2, 1, hidden_units=(32, 32)).cuda()]
    return Solver2D(lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2)],
                    [DirichletBVP2D(0, lambda y: torch.sin(3.14159265 * y), 1, zero, 0, zero, 1, zero)], nets=nets,
                    train_generator=Generator2D((12, 12), (0, 0), (1, 1), method="equally-spaced"),
                    valid_generator=Generator2D((8, 8), (0, 0), (1, 1), method="equally-spaced"))


def _nn(net):
    return net.NN if hasattr(net, "NN") else net.residual.NN


def _lr(s):
    s.optimizer.param_groups[0]["lr"] *= 0.3


def _scale_weights_no_grad(s):
    with torch.no_grad():
        _nn(s.nets[0])[0].weight.mul_(0.9)


def _shift_bias_through_data(s):
    _nn(s.nets[0])[2].bias.data.add_(0.05)


def _load_state_dict(s):
    sd = {k: 0.95 * v for k, v in s.nets[0].state_dict().items()}
    s.nets[0].load_state_dict(sd)


def _replace_a_layer(s):
    torch.manual_seed(99)
    old = _nn(s.nets[0])[2]
    new = nn.Linear(old.in_features, old.out_features).cuda()
    _nn(s.nets[0])[2] = new
    # (the optimiser has to learn about the new parameters as any torch user would tell it)
    s.optimizer = torch.optim.Adam(itertools.chain.from_iterable(n.parameters() for n in s.nets), lr=1e-3)


def _reassign_a_weight(s):
    layer = _nn(s.nets[0])[0]
    layer.weight = nn.Parameter(0.5 * layer.weight.detach().clone())
    s.optimizer = torch.optim.Adam(itertools.chain.from_iterable(n.parameters() for n in s.nets), lr=1e-3)


def _new_optimizer_sgd(s):
    s.optimizer = torch.optim.SGD(itertools.chain.from_iterable(n.parameters() for n in s.nets), lr=1e-2, momentum=0.5)


def _freeze_first_layer(s):
    _nn(s.nets[0])[0].weight.requires_grad_(False)
    _nn(s.nets[0])[0].bias.requires_grad_(False)


def _freeze_whole_second_net(s):
    s.nets[-1].requires_grad_(False)


def _doubling_hook(s):
    if not getattr(s, "_hooked_once", False):
        s._hooked_once = True
        _nn(s.nets[0])[1].register_forward_hook(lambda m, i, o: 1.5 * o)


def _hook_on_the_network(s):
    if not getattr(s, "_hooked_once", False):
        s._hooked_once = True
        s.nets[0].register_forward_hook(lambda m, i, o: o + 0.1)


def _swap_activation(s):
    _nn(s.nets[0])[1] = nn.Sigmoid()


def _new_equations(s):
    from neurodiffeq_amd import diff
    if len(s.nets) == 1 and s.diff_eqs.__code__.co_argcount == 2:
        s.diff_eqs = lambda u, t: [diff(u, t) + 2.0 * u]


def _new_initial_value(s):
    s.conditions[0].u_0 = 1.0 + 0.25 * s.global_epoch
    if hasattr(s.conditions[0], "u0"):
        s.conditions[0].u0 = s.conditions[0].u_0


def _more_batches(s):
    s.n_batches["train"] = 2


def _new_generator(s):
    from neurodiffeq_amd.generators import Generator1D, SamplerGenerator
    s.generator["train"] = SamplerGenerator(Generator1D(48, 0.0, 1.5, method="equally-spaced"))      # (solvers.py:148-149 wraps them so)


def _weight_decay(s):
    s.optimizer.param_groups[0]["weight_decay"] = 0.01


def _zero_the_moments(s):
    for st in s.optimizer.state.values():
        st["exp_avg"].zero_()


def _clip_weights(s):
    with torch.no_grad():
        for p in s.nets[0].parameters():
            p.clamp_(-0.5, 0.5)


def _betas_and_eps(s):
    s.optimizer.param_groups[0]["betas"] = (0.8, 0.9)
    s.optimizer.param_groups[0]["eps"] = 1e-6


def _replace_the_network(s):
    from neurodiffeq_amd.networks import FCNN
    torch.manual_seed(123)
    s.nets[0] = FCNN(_nn(s.nets[0])[0].in_features, 1, hidden_units=(16, 16)).cuda()
    s.optimizer = torch.optim.Adam(itertools.chain.from_iterable(n.parameters() for n in s.nets), lr=1e-3)


def _replace_the_condition(s):
    from neurodiffeq_amd.conditions import IVP
    s.conditions[0] = IVP(0.0, 0.5)


def _new_loss(s):
    s.loss_fn = lambda r, f, x: (r.abs()).mean()


def _add_param_group(s):
    if len(s.optimizer.param_groups) == 1:
        extra = nn.Parameter(torch.zeros(1, device="cuda"))
        s.optimizer.add_param_group({"params": [extra], "lr": 1e-2})


def _perturb_under_inference_mode(s):
    with torch.inference_mode():
        _nn(s.nets[0])[4].weight.add_(0.01)


SCENARIOS = {
    "lr": ("ode", _lr), "scale_weights_no_grad": ("ode", _scale_weights_no_grad), "bias_through_data": ("pde", _shift_bias_through_data),
    "load_state_dict": ("ode", _load_state_dict), "replace_a_layer": ("ode", _replace_a_layer), "reassign_a_weight": ("pde", _reassign_a_weight),
    "new_optimizer_sgd": ("ode", _new_optimizer_sgd), "freeze_first_layer": ("ode", _freeze_first_layer),
    "freeze_second_net": ("system", _freeze_whole_second_net), "hook_on_an_activation": ("ode", _doubling_hook),
    "hook_on_the_network": ("pde", _hook_on_the_network), "swap_activation": ("ode", _swap_activation), "new_equations": ("ode", _new_equations),
    "new_initial_value": ("ode", _new_initial_value), "more_batches": ("ode", _more_batches), "new_generator": ("ode", _new_generator),
    "betas_and_eps": ("pde", _betas_and_eps), "replace_the_network": ("ode", _replace_the_network),
    "replace_the_condition": ("ode", _replace_the_condition), "new_loss": ("ode", _new_loss), "add_param_group": ("ode", _add_param_group),
    "perturb_under_inference_mode": ("ode", _perturb_under_inference_mode),
    "wide_lr": ("wide", _lr), "wide_hook": ("wide", _doubling_hook), "wide_freeze": ("wide", _freeze_first_layer),
    "wide_load_state_dict": ("wide", _load_state_dict), "wide_reassign_a_weight": ("wide", _reassign_a_weight),
    "resnet_scale_skip": ("resnet", lambda s: s.nets[0].skip_connection.weight.data.mul_(0.5)),
    "resnet_hook_on_the_skip": ("resnet", lambda s: None if getattr(s, "_hk", False) else (setattr(s, "_hk", True), s.nets[0].skip_connection.register_forward_hook(lambda m, i, o: 2.0 * o))),
    "resnet_freeze_the_skip": ("resnet", lambda s: s.nets[0].skip_connection.weight.requires_grad_(False)),
    "swish_beta_changed": ("swish", lambda s: [setattr(m, "beta", 0.8 * m.beta) for m in s.nets[0].NN if hasattr(m, "beta")]),
    "elu_alpha_changed": ("elu", lambda s: [setattr(m, "alpha", 0.5 * m.alpha) for m in s.nets[0].NN if isinstance(m, nn.ELU)]),
    "weight_decay": ("pde", _weight_decay), "zero_the_moments": ("system", _zero_the_moments), "clip_weights": ("pde", _clip_weights),
}


def _train(fused, kind, change):
    from neurodiffeq_amd import autograd_ops
    s = _solver(fused, kind)

    def cb(solver):
        if solver.global_epoch == 1:
            solver._was_fused_at_first = solver.fused_active          # (before any change: the scenario starts on the fused path)
        if solver.global_epoch in AT:
            change(solver)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if fused == "off":
            with autograd_ops.native_autograd(False):
                s.fit(EPOCHS, callbacks=[cb])
        else:
            s.fit(EPOCHS, callbacks=[cb])
    return (np.array(s.metrics_history["train_loss"]), np.array(s.metrics_history["valid_loss"]),
            R.get_flat(s.nets).double().cpu().numpy(), s)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_a_callback_that_changes_the_solver_takes_effect_as_in_the_reference(name):
    kind, change = SCENARIOS[name]
    ft, fv, fp, fs = _train("auto", kind, change)
    pt, pv, pp, ps = _train("off", kind, change)
    assert len(ft) == len(pt) == EPOCHS and fs._was_fused_at_first and not ps._was_fused_at_first
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))
    assert rel(ft, pt) < 2e-5, (name, "train", ft, pt)          # (measured: <= 2e-6; north_star: 1e-5 per closure)
    assert rel(fv, pv) < 2e-5, (name, "valid", fv, pv)
    assert np.linalg.norm(fp - pp) <= 2e-5 * np.linalg.norm(pp), (name, "parameters")


def test_the_fused_path_is_used_until_the_change_and_left_loudly_when_it_must():
    kind, change = SCENARIOS["hook_on_an_activation"]
    from neurodiffeq_amd import autograd_ops          # noqa: F401
    s = _solver("auto", kind)
    s.fit(2)
    assert s.fused_active
    change(s)
    with pytest.warns(RuntimeWarning, match="NOT on the fused MI355X path"):
        s.fit(1)
    assert not s.fused_active


def test_a_metric_that_reads_python_state_follows_it():
    """Metrics are re-evaluated every batch by the reference (solvers.py:377-379); traced into the kernels they would freeze the
    numbers they read.  They are re-probed every epoch; one that changed is evaluated on the host from then on, training stays
    fused."""
    from neurodiffeq_amd import autograd_ops, diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.generators import Generator1D
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver1D
    import warnings

    def run(fused):
        ref = {"amp": 1.0}
        torch.manual_seed(5)
        s = Solver1D(lambda u, t: [diff(u, t) + u], [IVP(0.0, 1.0)], nets=[FCNN(1, 1, hidden_units=(32, 32)).cuda()],
                     train_generator=Generator1D(64, 0.0, 2.0, method="equally-spaced"),
                     valid_generator=Generator1D(32, 0.0, 2.0, method="equally-spaced"),
                     metrics={"err": lambda u, t: ((u - ref["amp"] * torch.exp(-t)) ** 2).mean()})
        s.fused = fused

        def cb(solver):
            if solver.global_epoch in AT:
                ref["amp"] *= 1.5
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if fused == "off":
                with autograd_ops.native_autograd(False):
                    s.fit(EPOCHS, callbacks=[cb])
            else:
                s.fit(EPOCHS, callbacks=[cb])
        h = s.metrics_history
        return np.array(h["train_loss"]), np.array(h["train__err"]), np.array(h["valid__err"]), s
    ft, fm, fv, fs = run("auto")
    pt, pm, pv, ps = run("off")
    assert fs.fused_active and fs._metrics_follow_state
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))
    assert rel(ft, pt) < 2e-5 and rel(fm, pm) < 2e-5 and rel(fv, pv) < 2e-5, (fm, pm)
    assert pm[AT[0]] != pytest.approx(pm[AT[0] - 1], rel=0.05)          # (the metric did move when the callback fired)


def test_autocast_sends_the_solver_to_the_references_closure():
    """`with torch.autocast("cuda"): solver.fit(...)`: the reference's layers run in half precision there; the fused kernels have one
    precision, so the solver runs the reference's closure while autocast is on (same numbers as plain torch modules) and returns to
    the fused path afterwards."""
    from neurodiffeq_amd import autograd_ops
    import warnings

    def run(fused):
        s = _solver(fused, "ode")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ctx = autograd_ops.native_autograd(False) if fused == "off" else autograd_ops.native_autograd(None)
            with ctx:
                s.fit(2)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    s.fit(3)
                    inside = s._fused_system(1)
                s.fit(2)
                after = s._fused_system(1)
        said = any("torch.autocast is enabled" in str(m.message) for m in w)
        return np.array(s.metrics_history["train_loss"]), inside, after, said
    a, inside, after, said = run("auto")
    b, _, _, _ = run("off")
    assert inside is None and after is not None and said
    assert np.allclose(a[:2], b[:2], rtol=2e-5) and np.allclose(a, b, rtol=5e-2)      # (bf16 layers: the two runs round alike, not equal)
