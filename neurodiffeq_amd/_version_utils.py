"""Deprecated names, the way the reference handles them (``_version_utils.py:5-48``): an old keyword is renamed to the new
one with a ``FutureWarning``; passing both is a ``KeyError``; an old class name constructs the new class with a warning."""
import functools
import warnings


def deprecated_alias(**aliases):
    """``@deprecated_alias(x_0='u_0')``: the decorated callable accepts ``x_0=`` as the old spelling of ``u_0=``."""
    def decorate(fn):
        @functools.wraps(fn)
        def accepting_old_names(*args, **kwargs):
            for old, new in aliases.items():
                if old not in kwargs:
                    continue
                if new in kwargs:
                    raise KeyError(f"{fn.__name__} received both `{old}` (deprecated) and `{new}` (recommended)")
                warnings.warn(f"The argument `{old}` is deprecated for {fn.__name__}; use `{new}` instead.", FutureWarning)
                kwargs[new] = kwargs.pop(old)
            return fn(*args, **kwargs)
        return accepting_old_names
    return decorate


def warn_deprecate_class(new_class):
    """A callable standing for a retired class name: warns, then builds ``new_class``."""
    @functools.wraps(new_class)
    def construct(*args, **kwargs):
        warnings.warn(f"This class name is deprecated, use {new_class} instead", FutureWarning)
        return new_class(*args, **kwargs)
    return construct
