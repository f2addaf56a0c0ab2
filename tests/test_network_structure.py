"""What a network IS can change between epochs (the reference calls ``net(x)`` every batch, so hooks, replaced layers, frozen
parameters and re-parametrised weights all take effect at once, solvers.py:369-395).  ``networks.describe`` refuses everything
the kernels would not reproduce, and ``networks.STRUCTURE`` is bumped by the change itself, so the solver re-keys its fused
system without walking module trees every epoch."""
import copy
import pickle

import pytest
import torch
import torch.nn as nn

from neurodiffeq_amd import networks
from neurodiffeq_amd.networks import FCNN, MonomialNN, Resnet, STRUCTURE, describe, track_structure


def test_plain_networks_are_described():
    for net in (FCNN(2, 1), FCNN(1, 3, hidden_units=(16, 16, 16)), Resnet(2, 1), nn.Sequential(nn.Linear(2, 8), nn.Tanh(), nn.Linear(8, 1)),
                nn.Sequential(MonomialNN(2), FCNN(4, 1))):
        assert describe(net) is not None, net


class _Doubling(FCNN):
    def forward(self, t):
        return 2.0 * super().forward(t)


class _Impostor(nn.Module):          # keeps a Sequential under .NN like an FCNN, computes something else
    def __init__(self):
        super().__init__()
        self.NN = nn.Sequential(nn.Linear(2, 8), nn.Tanh(), nn.Linear(8, 1))

    def forward(self, t):
        return self.NN(t) * t[:, 0:1]


class _Gated(nn.Linear):
    def forward(self, x):
        return super().forward(x) * 0.5


class _Twice(nn.Sequential):
    def forward(self, x):
        return super().forward(super().forward(x)) if False else 2.0 * super().forward(x)


def _with_forward_hook(where):
    net = FCNN(2, 1)
    target = {"net": net, "seq": net.NN, "layer": net.NN[0], "act": net.NN[1]}[where]
    target.register_forward_hook(lambda m, i, o: 2.0 * o)
    return net


def _frozen_layer():
    net = FCNN(2, 1)
    net.NN[0].weight.requires_grad_(False)
    return net


def _tied():
    net = FCNN(2, 1, hidden_units=(8, 8, 8))
    net.NN[4].weight = net.NN[2].weight
    return net


def _weight_norm():
    net = FCNN(2, 1)
    net.NN[2] = torch.nn.utils.weight_norm(net.NN[2])
    return net


def _parametrized():
    import torch.nn.utils.parametrize as P

    class Sym(nn.Module):
        def forward(self, w):
            return 0.5 * (w + w.transpose(0, 1))
    net = FCNN(2, 1, hidden_units=(8, 8))
    P.register_parametrization(net.NN[2], "weight", Sym())
    return net


REFUSED = {
    "subclass_with_its_own_forward": lambda: _Doubling(2, 1),
    "another_module_with_a_Sequential_under_NN": _Impostor,
    "linear_subclass_forward": lambda: nn.Sequential(_Gated(2, 8), nn.Tanh(), nn.Linear(8, 1)),
    "sequential_subclass_forward": lambda: _Twice(nn.Linear(2, 8), nn.Tanh(), nn.Linear(8, 1)),
    "hook_on_the_network": lambda: _with_forward_hook("net"),
    "hook_on_the_sequential": lambda: _with_forward_hook("seq"),
    "hook_on_a_layer": lambda: _with_forward_hook("layer"),
    "hook_on_an_activation": lambda: _with_forward_hook("act"),
    "pre_hook": lambda: (lambda n: (n.NN[0].register_forward_pre_hook(lambda m, i: (2.0 * i[0],)), n)[1])(FCNN(2, 1)),
    "backward_hook": lambda: (lambda n: (n.NN[0].register_full_backward_hook(lambda m, gi, go: None), n)[1])(FCNN(2, 1)),
    "frozen_layer": _frozen_layer,
    "tied_weights": _tied,
    "weight_norm": _weight_norm,
    "parametrized_weight": _parametrized,
    "resnet_with_a_hook_on_the_skip": lambda: (lambda n: (n.skip_connection.register_forward_hook(lambda m, i, o: o), n)[1])(Resnet(2, 1)),
}


@pytest.mark.parametrize("name", sorted(REFUSED))
def test_describe_refuses_what_the_kernels_would_not_reproduce(name):
    net = REFUSED[name]()
    assert describe(net) is None
    # ... and such a network still computes what torch says on the composite path (plain forward, hooks and all)
    x = torch.rand(5, 2)
    assert net(x).shape == (5, 1)


def test_structural_changes_bump_the_stamp():
    net = FCNN(2, 1)
    track_structure(net)
    changes = [
        lambda: net.NN.__setitem__(0, nn.Linear(2, 32)),                                   # a layer replaced
        lambda: setattr(net.NN[2], "weight", nn.Parameter(torch.zeros(32, 32))),          # a weight re-assigned
        lambda: net.NN.append(nn.Tanh()),                                                  # a module added
        lambda: net.NN[0].register_buffer("scale", torch.ones(1)),
        lambda: net.NN[0].register_forward_hook(lambda m, i, o: o).remove(),              # (added and removed: two bumps)
        lambda: net.register_forward_pre_hook(lambda m, i: None),
        lambda: net.NN[1].register_full_backward_hook(lambda m, gi, go: None),
        lambda: torch.nn.utils.weight_norm(net.NN[2]),
    ]
    for change in changes:
        before = STRUCTURE[0]
        change()
        assert STRUCTURE[0] > before, change
        track_structure(net)          # (what the rebuild that follows a bump does: new modules are tracked from then on)
    # what is NOT a structural change costs nothing: values, gradients, train / eval, untracked networks
    before = STRUCTURE[0]
    other = FCNN(2, 1)
    other.NN[0] = nn.Linear(2, 32)
    with torch.no_grad():
        net.NN[0].weight.mul_(0.5)
    net.NN[0].bias.data.add_(1.0)
    net.eval()
    net.train()
    net.zero_grad()
    assert STRUCTURE[0] == before


def test_tracked_networks_copy_and_pickle_as_plain_modules():
    net = FCNN(2, 1)
    track_structure(net)
    assert type(net.NN[0]._forward_hooks).__name__ == "_NotifyingHooks"
    for twin in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert type(twin.NN[0]._forward_hooks).__name__ == "OrderedDict"
        assert torch.equal(twin.NN[0].weight, net.NN[0].weight)
        before = STRUCTURE[0]
        twin.NN[0].register_forward_hook(lambda m, i, o: o)          # an untracked copy: not the solver's business
        assert STRUCTURE[0] == before
    h = net.NN[0].register_forward_hook(lambda m, i, o: o)
    assert describe(net) is None
    h.remove()
    assert describe(net) is not None


def test_the_custom_op_seam_asks_again_after_a_structural_change():
    from neurodiffeq_amd import autograd_ops
    net = FCNN(2, 1)
    info = describe(net)
    assert info is not None and len(info["params"]) == 6
    track_structure(net)
    cache = autograd_ops._SPECS.setdefault(net, {})
    cache["structure"] = STRUCTURE[0]
    cache[(0, torch.float32)] = ("stale",)
    net.NN[0].register_forward_hook(lambda m, i, o: 2.0 * o)
    # the cached answer belongs to another structure: dropped, describe() now refuses, the plain forward (with the hook) runs
    assert autograd_ops._spec_for(net, 0, torch.float32) is None
    x = torch.rand(4, 2)
    with torch.no_grad():
        want = net.NN[2](torch.tanh(2.0 * net.NN[0]._conv_forward(x) if False else 2.0 * torch.nn.functional.linear(x, net.NN[0].weight, net.NN[0].bias)))
        want = net.NN[4](torch.tanh(want))
        assert torch.allclose(net(x), want)

