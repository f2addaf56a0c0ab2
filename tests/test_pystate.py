"""The state watch behind the re-probe of diff_eqs / the conditions (neurodiffeq_amd/_pystate.py; reference behaviour:
solvers.py:380 re-evaluates the user's callables every batch, so ANY Python state they read takes effect at the next epoch).
Every case builds an equation callable that reads a value from somewhere, checks that a fresh watch is clean, changes the
value the way a callback would, and checks that the watch is dirty."""
import functools
import types

import numpy as np
import pytest
import torch

from neurodiffeq_amd._pystate import StateWatch

NU = 1.0


def _helper():
    return NU


class _Box:
    pass


class _Namespace:
    nu = 1.0


def _global():
    def f(u, t):
        return [u * NU]

    def m():
        global NU
        NU = NU + 1.0
    return f, m


def _through_helper_function():
    def f(u, t):
        return [u * _helper()]

    def m():
        global NU
        NU = NU + 1.0
    return f, m


def _closure_cell():
    nu = 1.0

    def f(u, t):
        return [u * nu]

    def m():
        nonlocal nu
        nu = 5.0
    return f, m


def _dict_entry():
    d = {"v": 1.0}
    return (lambda u, t: [u * d["v"]]), (lambda: d.__setitem__("v", 2.0))


def _nested_containers():
    d = {"a": {"b": [1.0, 2.0]}}
    return (lambda u, t: [u * d["a"]["b"][1]]), (lambda: d["a"]["b"].__setitem__(1, 9.0))


def _four_levels():
    d = {"a": {"b": {"c": {"d": 1.0}}}}
    return (lambda u, t: [u * d["a"]["b"]["c"]["d"]]), (lambda: d["a"]["b"]["c"].__setitem__("d", 2.0))


def _object_attribute():
    b = _Box()
    b.nu = 1.0
    return (lambda u, t: [u * b.nu]), (lambda: setattr(b, "nu", 2.0))


def _default_argument():
    def f(u, t, nu=1.0):
        return [u * nu]
    return f, (lambda: setattr(f, "__defaults__", (2.0,)))


def _partial_argument():
    lst = [1.0]

    def g(l, u, t):
        return [u * l[0]]
    return functools.partial(g, lst), (lambda: lst.__setitem__(0, 2.0))


def _callable_object():
    class Eq:
        def __init__(self):
            self.nu = 1.0

        def __call__(self, u, t):
            return [u * self.nu]
    e = Eq()
    return e, (lambda: setattr(e, "nu", 2.0))


def _bound_method():
    class Eq:
        def __init__(self):
            self.nu = 1.0

        def eqs(self, u, t):
            return [u * self.nu]
    e = Eq()
    return e.eqs, (lambda: setattr(e, "nu", 2.0))


def _class_value_shadowed_on_the_instance():
    class Eq:
        nu = 1.0

        def __call__(self, u, t):
            return [u * self.nu]
    e = Eq()
    return e, (lambda: setattr(e, "nu", 2.0))


def _class_value_changed_on_the_class():
    class Eq:
        nu = 1.0

        def __call__(self, u, t):
            return [u * self.nu]
    e = Eq()
    return e, (lambda: setattr(Eq, "nu", 2.0))


def _class_as_namespace():
    class K:
        nu = 1.0
    return (lambda u, t: [u * K.nu]), (lambda: setattr(K, "nu", 2.0))


def _module_level_class_as_namespace():
    def m():
        _Namespace.nu = _Namespace.nu + 1.0
    return (lambda u, t: [u * _Namespace.nu]), m


def _simple_namespace():
    ns = types.SimpleNamespace(nu=1.0)
    return (lambda u, t: [u * ns.nu]), (lambda: setattr(ns, "nu", 2.0))


def _numpy_scalar():
    d = {"v": np.float32(1.0)}
    return (lambda u, t: [u * float(d["v"])]), (lambda: d.__setitem__("v", np.float32(2.0)))


def _numpy_array_in_place():
    a = np.array([1.0, 2.0])
    return (lambda u, t: [u * a[0]]), (lambda: a.__setitem__(0, 7.0))


def _tensor_in_place():
    t0 = torch.tensor(1.0)
    return (lambda u, t: [u * t0]), (lambda: t0.mul_(2.0))


def _int_becomes_float():
    d = {"v": 1}
    return (lambda u, t: [u * d["v"]]), (lambda: d.__setitem__("v", 1.0))


def _function_in_a_list():
    coef = [lambda: 1.0]
    return (lambda u, t: [u * coef[0]()]), (lambda: coef.__setitem__(0, lambda: 2.0))


CASES = [_global, _through_helper_function, _closure_cell, _dict_entry, _nested_containers, _four_levels, _object_attribute,
         _default_argument, _partial_argument, _callable_object, _bound_method, _class_value_shadowed_on_the_instance,
         _class_value_changed_on_the_class, _class_as_namespace, _module_level_class_as_namespace, _simple_namespace,
         _numpy_scalar, _numpy_array_in_place, _tensor_in_place, _int_becomes_float, _function_in_a_list]


@pytest.mark.parametrize("make", CASES, ids=[c.__name__.strip("_") for c in CASES])
def test_state_watch_sees_the_change(make):
    f, mutate = make()
    watch = StateWatch([f])
    assert len(watch) > 0
    assert not watch.dirty()
    assert not watch.dirty()          # (checking does not disturb it)
    mutate()
    assert watch.dirty()


def test_rewriting_the_same_value_is_not_a_change():
    d = {"v": 1.0}
    watch = StateWatch([lambda u, t: [u * d["v"]]])
    d["v"] = 1.0
    assert not watch.dirty()


def test_a_stateless_lambda_costs_nothing():
    watch = StateWatch([lambda u, t: [u + t]])
    assert len(watch) == 0 and not watch.dirty()


def test_library_code_and_solver_bookkeeping_are_not_walked():
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import IVP
    from neurodiffeq_amd.solvers import Solver1D

    class Mine(Solver1D):
        def eqs(self, u, t):
            return [diff(u, t) + self.rate * u]

    s = Mine.__new__(Mine)
    s.rate = 1.0
    Solver1D.__init__(s, s.eqs, [IVP(0.0, 1.0)], t_min=0.0, t_max=1.0)
    watch = StateWatch([s.diff_eqs] + list(s.conditions))
    assert not watch.dirty()
    s.local_epoch = 17                 # the solver's own counters are not equation state
    s.metrics_history["train_loss"].append(0.5)
    assert not watch.dirty()
    s.rate = 2.0
    assert watch.dirty()
