from .cnf_regularization import RegularizedODEfunc  # noqa: F401
