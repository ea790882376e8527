"""CPU restatement of the FFJORD module stack (test oracle; see oracle/__init__.py).

The classes mirror the types pinned by the reference's checkpoints (module paths
`ffjord.lib.layers.{container,cnf,odefunc}`, `...diffeq_layers.basic`, `...wrappers.cnf_regularization`
recorded inside /root/reference/model_results/*/ffjord.pth) so that the fixture state_dicts load by name.
The divergence is computed with autograd exactly the way the reference does (d reverse passes for the exact
trace, one vector-Jacobian product for Hutchinson); the solver is oracle.odeint.
"""
import math

import torch
import torch.nn as nn

from .odeint import odeint, odeint_adjoint


class ConcatSquashLinear(nn.Module):
    """y = Linear(x) * sigmoid(gate(t)) + hyper_bias(t)   (a1 of section 8)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(1, dim_out, bias=False)
        self._hyper_gate = nn.Linear(1, dim_out)

    def forward(self, t, x):
        tt = t.reshape(1, 1)
        return self._layer(x) * torch.sigmoid(self._hyper_gate(tt)) + self._hyper_bias(tt)


# The other layers of ffjord/lib/layers/diffeq_layers/basic.py that are special cases of the concatsquash form
# (SURVEY.md section 8f row N4).  UPSTREAM-RECALL: the submodule is not vendored and no checkpoint uses them.
class IgnoreLinear(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)

    def forward(self, t, x):
        return self._layer(x)


class ConcatLinear(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in + 1, dim_out)

    def forward(self, t, x):
        tt = torch.ones_like(x[:, :1]) * t
        return self._layer(torch.cat([tt, x], 1))


class ConcatLinear_v2(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(1, dim_out, bias=False)

    def forward(self, t, x):
        return self._layer(x) + self._hyper_bias(t.reshape(1, 1))


class SquashLinear(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper = nn.Linear(1, dim_out)

    def forward(self, t, x):
        return self._layer(x) * torch.sigmoid(self._hyper(t.reshape(1, 1)))


_LAYERS = {"ignore": IgnoreLinear, "concat": ConcatLinear, "concatcoord": ConcatLinear, "concat_v2": ConcatLinear_v2,
           "squash": SquashLinear, "concatsquash": ConcatSquashLinear}
_ACTIVATIONS = {"tanh": nn.Tanh, "softplus": nn.Softplus, "relu": nn.ReLU, "elu": nn.ELU}


class ODEnet(nn.Module):
    """Stack of time-conditioned linear layers with an activation between (not after) them (a2)."""

    def __init__(self, hidden_dims, input_shape, strides=None, conv=False, layer_type="concatsquash",
                 nonlinearity="tanh", num_squeeze=0):
        super().__init__()
        assert not conv and layer_type in _LAYERS, "oracle restates the fully connected time-conditioned layers only"
        self.num_squeeze = num_squeeze
        dims = (input_shape[0],) + tuple(hidden_dims) + (input_shape[0],)
        self.layers = nn.ModuleList(_LAYERS[layer_type](a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.activation_fns = nn.ModuleList(_ACTIVATIONS[nonlinearity]() for _ in range(len(dims) - 2))

    def forward(self, t, y):
        h = y
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            h = layer(t, h)
            if i < last:
                h = self.activation_fns[i](h)
        return h


def divergence_bf(dx, y, **_unused):
    """Exact trace of d(dx)/dy: one reverse pass per data dimension, graph kept for double backward (a3)."""
    total = 0.0
    for i in range(y.shape[1]):
        total = total + torch.autograd.grad(dx[:, i].sum(), y, create_graph=True)[0][:, i]
    return total


def divergence_approx(f, y, e=None):
    """Hutchinson estimate e^T (df/dy) e with a fixed probe e (a4)."""
    vjp = torch.autograd.grad(f, y, e, create_graph=True)[0]
    return (vjp * e).reshape(y.shape[0], -1).sum(dim=1)


class ODEfunc(nn.Module):
    """(y, logp) -> (f(t, y), -div f); counts evaluations; owns the Hutchinson probe (a5)."""

    def __init__(self, diffeq, divergence_fn="approximate", residual=False, rademacher=False):
        super().__init__()
        assert divergence_fn in ("brute_force", "approximate")
        self.diffeq = diffeq
        self.residual = residual
        self.rademacher = rademacher
        self.divergence_fn = divergence_bf if divergence_fn == "brute_force" else divergence_approx
        self.register_buffer("_num_evals", torch.tensor(0.0))
        self._e = None

    def before_odeint(self, e=None):
        self._e = e
        self._num_evals.fill_(0)

    def num_evals(self):
        return self._num_evals.item()

    def forward(self, t, states):
        assert len(states) >= 2
        y = states[0]
        self._num_evals += 1
        t = torch.as_tensor(t).to(y)
        if self._e is None:
            if self.rademacher:
                self._e = torch.randint(0, 2, y.shape).to(y) * 2 - 1
            else:
                self._e = torch.randn_like(y)
        with torch.enable_grad():
            y.requires_grad_(True)
            t.requires_grad_(True)
            dy = self.diffeq(t, y)
            if not self.training and dy.reshape(dy.shape[0], -1).shape[1] == 2:
                div = divergence_bf(dy, y)          # eval mode on 2-D data always uses the exact trace
            else:
                div = self.divergence_fn(dy, y, e=self._e)
            div = div.reshape(y.shape[0], 1)
        if self.residual:
            dy = dy - y
            div = div - float(y[0].numel())
        return (dy, -div) + tuple(torch.zeros_like(s).requires_grad_(True) for s in states[2:])


# The kinetic regularisers of ffjord/lib/layers/wrappers/cnf_regularization.py (un-vendored submodule; restated as recalled,
# original spelling of the names kept): each returns ONE scalar per evaluation, the batch mean of a function of dx = dz/dt,
# which RegularizedODEfunc appends to the dynamics as the slope of one more state.
def _batch_root_mean_squared(tensor):
    tensor = tensor.view(tensor.shape[0], -1)
    return torch.mean(torch.norm(tensor, p=2, dim=1) / tensor.shape[1] ** 0.5)


def l1_regularzation_fn(x, logp, dx, dlogp, unused_context):
    del x, logp, dlogp
    return torch.mean(torch.abs(dx))


def l2_regularzation_fn(x, logp, dx, dlogp, unused_context):
    del x, logp, dlogp
    return _batch_root_mean_squared(dx)


class RegularizedODEfunc(nn.Module):
    """Wrapper present in the checkpoints even with zero regularisers: a pass-through then (a6)."""

    def __init__(self, odefunc, regularization_fns):
        super().__init__()
        self.odefunc = odefunc
        self.regularization_fns = tuple(regularization_fns)

    def before_odeint(self, *args, **kwargs):
        self.odefunc.before_odeint(*args, **kwargs)

    def forward(self, t, state):
        with torch.enable_grad():
            x, logp = state[:2]
            x.requires_grad_(True)
            logp.requires_grad_(True)
            dstate = self.odefunc(t, (x, logp))
            if len(state) > 2:
                dx, dlogp = dstate[:2]
                return dstate + tuple(fn(x, logp, dx, dlogp, None) for fn in self.regularization_fns)
            return dstate

    @property
    def _num_evals(self):
        return self.odefunc._num_evals


class CNF(nn.Module):
    """One continuous flow block: integrates (z, logp) over [0, sqrt_end_time**2] (a7)."""

    def __init__(self, odefunc, T=1.0, train_T=False, regularization_fns=None, solver="dopri5", atol=1e-5, rtol=1e-5):
        super().__init__()
        root = torch.sqrt(torch.tensor(float(T)))
        if train_T:
            self.register_parameter("sqrt_end_time", nn.Parameter(root))
        else:
            self.register_buffer("sqrt_end_time", root)
        nreg = 0
        if regularization_fns is not None:
            odefunc = RegularizedODEfunc(odefunc, regularization_fns)
            nreg = len(regularization_fns)
        self.odefunc = odefunc
        self.nreg = nreg
        self.regularization_states = None
        self.solver, self.atol, self.rtol = solver, atol, rtol
        self.test_solver, self.test_atol, self.test_rtol = solver, atol, rtol
        self.solver_options = {}
        self.stats = {}
        self.adjoint_error_control = "state_only"   # | 'rows' | 'all' (odeint_adjoint)

    def forward(self, z, logpz=None, integration_times=None, reverse=False):
        _logpz = torch.zeros(z.shape[0], 1).to(z) if logpz is None else logpz
        if integration_times is None:
            # torch.tensor([...]) on purpose: it detaches the end time, so it never trains
            integration_times = torch.tensor([0.0, float(self.sqrt_end_time * self.sqrt_end_time)]).to(z)
        if reverse:
            integration_times = torch.flip(integration_times, [0])
        self.odefunc.before_odeint()
        self.stats = {}
        if self.training:
            reg = tuple(torch.tensor(0.0).to(z) for _ in range(self.nreg))
            dop = self.solver == "dopri5"
            state_t = odeint_adjoint(
                self.odefunc, (z, _logpz) + reg, integration_times.to(z),
                atol=[self.atol, self.atol] + [1e20] * len(reg) if dop else self.atol,
                rtol=[self.rtol, self.rtol] + [1e20] * len(reg) if dop else self.rtol,
                method=self.solver, options=self.solver_options, stats=self.stats,
                adjoint_control=self.adjoint_error_control)
        else:
            state_t = odeint_adjoint(
                self.odefunc, (z, _logpz), integration_times.to(z),
                atol=self.test_atol, rtol=self.test_rtol, method=self.test_solver, stats=self.stats)
        if len(integration_times) == 2:
            state_t = tuple(s[1] for s in state_t)
        z_t, logpz_t = state_t[:2]
        self.regularization_states = state_t[2:]
        return (z_t, logpz_t) if logpz is not None else z_t

    def get_regularization_states(self):
        reg_states = self.regularization_states
        self.regularization_states = None
        return reg_states

    def num_evals(self):
        return self.odefunc._num_evals.item()


class SequentialFlow(nn.Module):
    """Chain of flow blocks; `reverse=True` walks it backwards (a8)."""

    def __init__(self, layersList):
        super().__init__()
        self.chain = nn.ModuleList(layersList)

    def forward(self, x, logpx=None, reverse=False, inds=None):
        if inds is None:
            inds = range(len(self.chain) - 1, -1, -1) if reverse else range(len(self.chain))
        if logpx is None:
            for i in inds:
                x = self.chain[i](x, reverse=reverse)
            return x
        for i in inds:
            x, logpx = self.chain[i](x, logpx, reverse=reverse)
        return x, logpx


def get_transforms(model):
    """-> (sample_fn, density_fn), the two eval entry points used at /root/reference/model_plot.py:94-98 (a12)."""

    def sample_fn(z, logpz=None):
        return model(z, reverse=True) if logpz is None else model(z, logpz, reverse=True)

    def density_fn(x, logpx=None):
        return model(x, reverse=False) if logpx is None else model(x, logpx, reverse=False)

    return sample_fn, density_fn


def standard_normal_logprob(z):
    return -0.5 * math.log(2 * math.pi) - z.pow(2) / 2


def build_cnf(dim, hidden_dims=(64, 64, 64), num_blocks=1, T=0.5, train_T=True, divergence_fn="brute_force",
              solver="dopri5", atol=1e-5, rtol=1e-5, nonlinearity="tanh", rademacher=False, residual=False,
              layer_type="concatsquash", regularization_fns=()):
    """The model the reference's checkpoints hold: `num_blocks` CNF blocks, concatsquash/tanh ODEnet."""
    blocks = []
    for _ in range(num_blocks):
        net = ODEnet(tuple(hidden_dims), (dim,), strides=None, conv=False, layer_type=layer_type,
                     nonlinearity=nonlinearity)
        fn = ODEfunc(net, divergence_fn=divergence_fn, residual=residual, rademacher=rademacher)
        blocks.append(CNF(fn, T=T, train_T=train_T, regularization_fns=tuple(regularization_fns), solver=solver, atol=atol, rtol=rtol))
    return SequentialFlow(blocks)
