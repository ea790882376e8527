"""Pass-through wrapper the reference's checkpoints carry even with zero regularisers
(ffjord/lib/layers/wrappers/cnf_regularization.py)."""
import torch.nn as nn


class RegularizedODEfunc(nn.Module):
    def __init__(self, odefunc, regularization_fns):
        super().__init__()
        self.odefunc = odefunc
        self.regularization_fns = tuple(regularization_fns)
        if self.regularization_fns:
            raise NotImplementedError("kinetic/Jacobian regularisers are not built into the HIP path "
                                      "(the reference's runs use none: regularization_fns=())")

    def before_odeint(self, *args, **kwargs):
        self.odefunc.before_odeint(*args, **kwargs)

    def forward(self, t, state):
        return self.odefunc(t, state)

    @property
    def _num_evals(self):
        return self.odefunc._num_evals
