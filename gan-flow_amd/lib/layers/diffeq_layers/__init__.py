from .basic import ConcatSquashLinear  # noqa: F401
