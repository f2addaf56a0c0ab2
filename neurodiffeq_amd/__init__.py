"""neurodiffeq_amd -- MI355X-native training core behind the NeuroDiffGym/neurodiffeq solver API.

Same names as the reference package (``diff``, ``operators``, ``networks.FCNN``, ``conditions``, ``generators``,
``solvers.Solver1D/Solver2D``) so that ``import neurodiffeq_amd as neurodiffeq`` is a drop-in for the
``Solver*.fit()`` inner loop; the loop's compute runs as hand-written gfx950 HIP kernels.

Unlike the reference (``__init__.py:22``) importing this package does not change torch's global default dtype or
device; the fused path is fp32 and puts the networks on the GPU itself."""
from .neurodiffeq import diff, safe_diff, unsafe_diff  # noqa: F401
from . import operators, networks, conditions, generators, solvers, losses, utils, function_basis  # noqa: F401
from . import autograd_ops  # noqa: F401  (registers torch.ops.ndq.*)
from .autograd_ops import set_native_autograd  # noqa: F401

__version__ = "0.1.0"
