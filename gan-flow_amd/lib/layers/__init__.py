"""Same names as the reference's `ffjord.lib.layers` package (module paths recorded in the checkpoints,
SURVEY.md section 2a)."""
from .container import SequentialFlow  # noqa: F401
from .cnf import CNF  # noqa: F401
from .odefunc import ODEfunc, ODEnet, divergence_bf, divergence_approx  # noqa: F401
from .diffeq_layers.basic import ConcatSquashLinear  # noqa: F401
from .wrappers.cnf_regularization import RegularizedODEfunc  # noqa: F401
from .normalization import MovingBatchNorm1d, MovingBatchNormNd  # noqa: F401
