"""Run the REFERENCE's own test files for the hot path against this package ("reuse as parity suite").

``python tests/reference_suite_runner.py <reference root> [amd|ref]`` -- ``amd``: every ``neurodiffeq[.sub]`` import of the
reference's tests resolves to ``neurodiffeq_amd[.sub]`` (``neurodiffeq_amd.compat.install``: the out-of-scope layers --
callbacks, monitors, legacy ode / pde modules -- are the reference's own files bound to this package's classes) and the
reference's import side effect (``__init__.py:22``: ``set_tensor_type`` -> float64; here on the CPU) is applied by hand; ``ref``: the unmodified reference itself, as the
control.  Prints one JSON line: {"passed": [...], "failed": [...]} of pytest node ids.  The reference's tests are read
from the reference checkout where it lies (never copied); tests/test_reference_suite.py skips when it is absent."""
import importlib
import json
import os
import sys

#: the test files of the functions SURVEY.md 8(a) puts on the hot path (the others test monitors, callbacks, legacy
#: ode / pde entry points, the numerical solvers: out of scope)
FILES = ["test_neurodiffeq.py", "test_operators_cartesian.py", "test_operators_cylindrical.py", "test_operators_identities.py",
         "test_operators_spherical.py", "test_networks.py", "test_conditions.py", "test_generators.py", "test_losses.py",
         "test_solvers.py", "test_function_basis.py"]


def main():
    ref_root, which = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "amd")
    os.environ.setdefault("MPLBACKEND", "Agg")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if which == "amd":
        sys.path.insert(0, here)
        # (tests/golden/_refshim: `seaborn`, `ordered_set` -- plotting / container dependencies of the reference's monitors that
        # are not installed here)
        sys.path.insert(1, os.path.join(here, "tests", "golden", "_refshim"))
        from neurodiffeq_amd import compat
        compat.install(reference_dir=os.path.join(ref_root, "neurodiffeq"))
        from neurodiffeq_amd.utils import set_tensor_type
        set_tensor_type("cpu", 64)
    else:
        # (tests/golden/_refshim: `seaborn`, `ordered_set` -- plotting / container dependencies that are not installed here)
        sys.path[:0] = [os.path.join(here, "tests", "golden", "_refshim"), ref_root]
        import neurodiffeq  # noqa: F401  (sets float64 itself)
    import pytest

    class Collect:
        def __init__(self):
            self.passed, self.failed = [], []

        def pytest_runtest_logreport(self, report):
            if report.when == "call":
                (self.passed if report.passed else self.failed).append(report.nodeid.split("tests/")[-1])
            elif report.failed:
                self.failed.append(report.nodeid.split("tests/")[-1])

        def pytest_collectreport(self, report):
            if report.failed:
                self.failed.append(str(report.nodeid).split("tests/")[-1] + "::<collection>")

    c = Collect()
    files = [os.path.join(ref_root, "tests", f) for f in FILES]
    pytest.main(["-q", "-x" if os.environ.get("REFSUITE_X") else "-q", "-p", "no:cacheprovider", "--rootdir", ref_root, "-W", "ignore",
                 "--continue-on-collection-errors", *files], plugins=[c])
    print("REFSUITE " + json.dumps({"passed": sorted(c.passed), "failed": sorted(set(c.failed))}))


if __name__ == "__main__":
    main()
