from .basic import (ConcatLinear, ConcatLinear_v2, ConcatSquashLinear, IgnoreLinear, LAYER_TYPES,  # noqa: F401
                    SquashLinear)
