"""Global dtype / device helpers with the reference's names (neurodiffeq/utils.py:10-68).

Difference from the reference, on purpose: importing :mod:`neurodiffeq_amd` does NOT flip the process-wide default
dtype to fp64 / device to cuda (reference ``__init__.py:22``) -- a script that only changes its import line therefore
trains in fp32 unless it calls ``set_tensor_type(float_bits=64)`` itself (INTEGRATION.md leads with this).  fp32
systems run on the single-launch closure kernels and the native epochs; fp64 networks -- the reference's default
precision -- run fused in double as well (single-network systems on the closure kernel compiled for fp64, the others on the
three-kernel pipeline: fp64 stream kernels of libndq64.so around the traced pointwise kernel compiled in double; epoch
bookkeeping and Adam on the device), or, for shapes whose weights do not fit LDS in double, the reference's
closure on torch autograd with the network forward / backward on the fp64 stream kernels."""
import random
import re

import numpy as np
import torch


def set_tensor_type(device=None, float_bits=32):
    if not isinstance(float_bits, int):
        raise ValueError(f"float_bits must be int, got {type(float_bits)}")
    if float_bits == 32:
        torch.set_default_dtype(torch.float32)
    elif float_bits == 64:
        torch.set_default_dtype(torch.float64)
    else:
        raise ValueError(f"float_bits must be 32 or 64, got {float_bits}")
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if device != "cpu" and not re.fullmatch(r"cuda(?::\d+)?", device):
        raise ValueError(f"Unknown device '{device}'; device must be either 'cuda', 'cuda:x' where x is the device "
                         f"number, 'cpu'")
    # (torch.set_default_device installs a process-wide TorchFunctionMode that intercepts EVERY tensor method call, ~1 us
    # each -- also for 'cpu', where it changes nothing: the CPU default is restored by removing the mode instead)
    torch.set_default_device(None if device == "cpu" else device)


def safe_mkdir(path):
    """``mkdir -p`` (utils.py:44-45; the reference's callbacks / monitors call it)."""
    import os
    os.makedirs(path, exist_ok=True)


def set_seed(seed_value, ignore_numpy=False, ignore_torch=False, ignore_random=False):
    if not ignore_numpy:
        np.random.seed(seed_value)
    if not ignore_torch:
        torch.manual_seed(seed_value)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed_value)
    if not ignore_random:
        random.seed(seed_value)
