"""RegularizedODEfunc and the kinetic regularisers of ffjord/lib/layers/wrappers/cnf_regularization.py (SURVEY.md section 2a,
section 8a row a6, section 8f row N4).  The reference's checkpoints carry the wrapper with zero regularisers: a pass-through.
With `l1_regularzation_fn` / `l2_regularzation_fn` (the original's spelling) in `regularization_fns`, CNF.forward integrates
one more state per regulariser inside the HIP solve kernels (cnf_solve_fwd_reg / cnf_solve_bwd_reg); the functions below are
MARKERS that name the integrand to those kernels, as `divergence_bf` / `divergence_approx` do in odefunc.py.  The other
regularisers of the original (directional, Jacobian norms) need second derivatives through the sweep and are not built."""
import torch.nn as nn


def _marker(kind):
    def fn(x, logp, dx, dlogp, unused_context):
        raise RuntimeError("%s_regularzation_fn is evaluated inside the HIP solve kernels (csrc/cnf_device.h, reg_integrand); "
                           "it is not callable on tensors -- the CPU restatement is oracle.%s_regularzation_fn" % (kind, kind))
    fn.__name__ = "%s_regularzation_fn" % kind
    fn.cnf_reg_kind = kind
    return fn


l1_regularzation_fn = _marker("l1")   # mean over the batch of mean_i |dz_i/dt|
l2_regularzation_fn = _marker("l2")   # mean over the batch of ||dz/dt||_2 / sqrt(dim)


class RegularizedODEfunc(nn.Module):
    def __init__(self, odefunc, regularization_fns):
        super().__init__()
        self.odefunc = odefunc
        self.regularization_fns = tuple(regularization_fns)
        kinds = []
        for fn in self.regularization_fns:
            kind = getattr(fn, "cnf_reg_kind", None)
            if kind is None:
                raise NotImplementedError("regulariser %r is not built into the HIP path: built are l1_regularzation_fn and "
                                          "l2_regularzation_fn of this module (the reference's runs use none)" % (fn,))
            kinds.append(kind)
        self.reg_kinds = tuple(kinds)

    def before_odeint(self, *args, **kwargs):
        self.odefunc.before_odeint(*args, **kwargs)

    def forward(self, t, state):
        return self.odefunc(t, state)

    @property
    def _num_evals(self):
        return self.odefunc._num_evals
