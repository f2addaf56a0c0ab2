"""``import neurodiffeq`` -> this package, with the reference's out-of-scope layers riding on top.

This package re-implements the hot path of NeuroDiffGym/neurodiffeq (``diff`` / ``operators`` / ``networks`` /
``conditions`` / ``generators`` / ``solvers`` / ``losses``) and nothing else: the reference's callbacks, monitors, legacy
``ode`` / ``pde`` / ``pde_spherical`` entry points, ``temporal``, ``hypersolver`` are plain Python on top of that core and
stay the reference's own code.  ``install()`` makes the two compose in one process:

    import neurodiffeq_amd.compat as compat
    compat.install()                       # or install(reference_dir="/path/to/site-packages/neurodiffeq")
    from neurodiffeq.solvers import Solver2D            # -> neurodiffeq_amd.solvers (MI355X kernels)
    from neurodiffeq.callbacks import StopCallback      # -> the reference's callbacks.py, bound to the classes above

Every ``neurodiffeq.<sub>`` this package implements resolves to ``neurodiffeq_amd.<sub>``; any other submodule is loaded from
the reference distribution's source file AS ``neurodiffeq.<sub>``, so its relative imports (``from .solvers import ...``)
bind to this package.  The reference's ``__init__`` is never executed (its import side effect -- float64 + cuda defaults,
``__init__.py:22`` -- is the caller's choice here: ``neurodiffeq_amd.utils.set_tensor_type``).  tests/test_reference_suite.py
runs the reference's own test files for the hot path through this shim.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

OWN = ("neurodiffeq", "operators", "networks", "conditions", "generators", "solvers", "losses", "utils", "function_basis",
       "_version_utils", "autograd_ops", "optim", "parallel")


class _ReferenceLayers(importlib.abc.MetaPathFinder):
    """Finds ``neurodiffeq.<sub>`` for submodules this package does not implement in the reference's source directory."""

    def __init__(self, reference_dir):
        self.reference_dir = reference_dir

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("neurodiffeq.") or fullname.count(".") != 1:
            return None
        sub = fullname.split(".", 1)[1]
        if sub in OWN or self.reference_dir is None:
            return None
        for cand, pkg in ((os.path.join(self.reference_dir, sub + ".py"), False),
                          (os.path.join(self.reference_dir, sub, "__init__.py"), True)):
            if os.path.exists(cand):
                return importlib.util.spec_from_file_location(fullname, cand, submodule_search_locations=[os.path.dirname(cand)] if pkg else None)
        return None


def _locate_reference():
    """Directory of an installed ``neurodiffeq`` distribution, found WITHOUT importing it."""
    if "neurodiffeq" in sys.modules:
        return None
    try:
        spec = importlib.util.find_spec("neurodiffeq")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    return list(spec.submodule_search_locations)[0]


def install(reference_dir=None):
    """Alias ``neurodiffeq`` to this package (module docstring).  ``reference_dir``: the ``neurodiffeq/`` source directory of
    the reference distribution for the layers this package does not implement (default: an installed ``neurodiffeq``, if
    any; without one only the hot-path modules are importable).  Idempotent; returns the directory in use (or None)."""
    import neurodiffeq_amd
    if reference_dir is None:
        current = sys.modules.get("neurodiffeq")
        if current is not None and current is not neurodiffeq_amd:
            raise RuntimeError("the reference package `neurodiffeq` is already imported in this process; call "
                               "neurodiffeq_amd.compat.install() before anything imports it")
        reference_dir = _locate_reference()
    for f in [f for f in sys.meta_path if isinstance(f, _ReferenceLayers)]:
        sys.meta_path.remove(f)
    sys.modules["neurodiffeq"] = neurodiffeq_amd
    for sub in OWN:
        sys.modules["neurodiffeq." + sub] = importlib.import_module("neurodiffeq_amd." + sub)
    sys.meta_path.insert(0, _ReferenceLayers(reference_dir))
    return reference_dir
