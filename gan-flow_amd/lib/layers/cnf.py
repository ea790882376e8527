"""CNF block with the reference's signature and attributes (ffjord/lib/layers/cnf.py; SURVEY.md section 2a, section 8a
row a7, section 8b).  Where the reference calls `odeint_adjoint(self.odefunc, (z, logpz), integration_times, atol=[a, a],
rtol=[r, r], method=self.solver, options=self.solver_options)`, this class makes ONE typed call into libcnf_hip.so
for the forward solve and one for the adjoint pass (torch.autograd.Function below).  Tensors must live on a HIP
device: there is no CPU path."""
import torch
import torch.nn as nn

from ... import _native
from .wrappers.cnf_regularization import RegularizedODEfunc


def _inner(odefunc):
    return odefunc.odefunc if isinstance(odefunc, RegularizedODEfunc) else odefunc


class _Solve(torch.autograd.Function):
    """(z0, logp0, theta) -> (z1, logp1); backward = the adjoint ODE, integrated by the HIP kernel."""

    @staticmethod
    def forward(ctx, z, logp, flat_params, block, t0, t1, opts, net, e):
        z1, logp1, stats = _native.solve_fwd(net, flat_params.detach(), z.detach(), logp.detach(), t0, t1, opts, e)
        block._after_solve(stats, backward=False)
        ctx.block, ctx.t0, ctx.t1, ctx.opts, ctx.net = block, t0, t1, opts, net
        ctx.fwd_stats = stats      # device words: the adjoint solve exits at once if this solve failed (opts.after)
        ctx.save_for_backward(z1, logp1, flat_params.detach(), e if e is not None else z.new_zeros(0))
        return z1, logp1

    @staticmethod
    def backward(ctx, grad_z, grad_logp):
        z1, logp1, flat_params, e = ctx.saved_tensors
        e = e if e.numel() else None
        grad_z = torch.zeros_like(z1) if grad_z is None else grad_z
        grad_logp = torch.zeros_like(logp1) if grad_logp is None else grad_logp
        gx, gl, gp, stats = _native.solve_bwd(ctx.net, flat_params, z1, logp1, grad_z, grad_logp, ctx.t0, ctx.t1,
                                              _native.with_after(ctx.opts, ctx.fwd_stats), e)
        ctx.block._after_solve(stats, backward=True)
        return gx, gl, gp, None, None, None, None, None, None


class _SolveTimes(torch.autograd.Function):
    """(z0, logp0, theta) -> the state at every time of `times`, stacked (odeint with len(t) > 2): one adaptive forward
    integration (cnf_solve_fwd_times); backward = OdeintAdjointMethod.backward's loop over the intervals, last first,
    each an adjoint solve started from the stored state at its upper time, the output cotangent of that time added to
    the one carried back (oracle/odeint.py::_Adjoint.backward restates it)."""

    @staticmethod
    def forward(ctx, z, logp, flat_params, block, times, opts, net, e):
        z_t, logp_t, stats = _native.solve_fwd_times(net, flat_params.detach(), z.detach(), logp.detach(), times, opts, e)
        block._after_solve(stats, backward=False)
        ctx.block, ctx.times, ctx.opts, ctx.net = block, tuple(times), opts, net
        ctx.fwd_stats = stats
        ctx.save_for_backward(z_t, logp_t, flat_params.detach(), e if e is not None else z.new_zeros(0))
        return z_t, logp_t

    @staticmethod
    def backward(ctx, grad_z, grad_logp):
        z_t, logp_t, flat_params, e = ctx.saved_tensors
        e = e if e.numel() else None
        grad_z = torch.zeros_like(z_t) if grad_z is None else grad_z
        grad_logp = torch.zeros_like(logp_t) if grad_logp is None else grad_logp
        n = len(ctx.times)
        a_z, a_l = grad_z[n - 1].contiguous(), grad_logp[n - 1].contiguous()
        g_theta = torch.zeros_like(flat_params)
        after = ctx.fwd_stats
        for i in range(n - 1, 0, -1):
            gx, gl, gp, stats = _native.solve_bwd(ctx.net, flat_params, z_t[i].contiguous(), logp_t[i].contiguous(), a_z, a_l,
                                                  ctx.times[i - 1], ctx.times[i], _native.with_after(ctx.opts, after), e)
            ctx.block._after_solve(stats, backward=True)
            after = stats
            a_z, a_l = gx + grad_z[i - 1], gl + grad_logp[i - 1]
            g_theta = g_theta + gp
        return a_z, a_l, g_theta, None, None, None, None, None


class _SolveReg(torch.autograd.Function):
    """The regularised solve: (z0, logp0, theta) -> (z1, logp1, regularisation states[n]); backward = cnf_solve_bwd_reg."""

    @staticmethod
    def forward(ctx, z, logp, flat_params, block, t0, t1, opts, net, e, reg):
        z1, logp1, regs, stats = _native.solve_fwd_reg(net, reg, flat_params.detach(), z.detach(), logp.detach(), t0, t1, opts, e)
        block._after_solve(stats, backward=False)
        ctx.block, ctx.t0, ctx.t1, ctx.opts, ctx.net, ctx.reg = block, t0, t1, opts, net, reg
        ctx.fwd_stats = stats
        ctx.save_for_backward(z1, logp1, regs, flat_params.detach(), e if e is not None else z.new_zeros(0))
        return z1, logp1, regs

    @staticmethod
    def backward(ctx, grad_z, grad_logp, grad_regs):
        z1, logp1, regs, flat_params, e = ctx.saved_tensors
        e = e if e.numel() else None
        grad_z = torch.zeros_like(z1) if grad_z is None else grad_z
        grad_logp = torch.zeros_like(logp1) if grad_logp is None else grad_logp
        grad_regs = torch.zeros_like(regs) if grad_regs is None else grad_regs
        gx, gl, gp, stats = _native.solve_bwd_reg(ctx.net, ctx.reg, flat_params, z1, logp1, regs, grad_z, grad_logp, grad_regs,
                                                  ctx.t0, ctx.t1, _native.with_after(ctx.opts, ctx.fwd_stats), e)
        ctx.block._after_solve(stats, backward=True)
        return gx, gl, gp, None, None, None, None, None, None, None


class CNF(nn.Module):
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
        self.solver = solver
        self.atol = atol
        self.rtol = rtol
        self.test_solver = solver
        self.test_atol = atol
        self.test_rtol = rtol
        self.solver_options = {}
        # not in the reference: knobs of the fused integrator
        self.tile_rows = 0          # 0 = one step size for the whole batch (reference semantics); 16 = per-tile control
        self.sync_stats = True      # read the solver status back after every solve (one host sync, as the reference has)
        self.spread_adjoint = False  # backward(): CNF_FLAG_SPREAD, two workgroups per 16-row tile (include/cnf_hip.h); same
                                     # gradients, faster for batches <= 2048 rows; not for solves sharing a GPU with others
        # which tensors of the adjoint state drive backward()'s step control: "state_only" =
        # (z, logp), what the reference's length-2 tolerance lists do in torchdiffeq 0.0.1 as recalled; "all" adds
        # (a_z, a_logp) (CNF_FLAG_ADJ_CONTROL_ALL, include/cnf_hip.h).  Neither is pinned by a reference artefact.
        self.adjoint_error_control = "state_only"
        self.last_stats = {}

    # ------------------------------------------------------------------------------------------------
    def _after_solve(self, stats_words, backward):
        inner = _inner(self.odefunc)
        if backward:
            # the reference's counter keeps running through backward(): +1 for f(T, y_T), then the augmented stages
            inner._num_evals += (stats_words[1] + 1).to(inner._num_evals)
        else:
            inner._num_evals.copy_(stats_words[1])
        key = "backward" if backward else "forward"
        if self.sync_stats:
            st = _native.Stats(stats_words)
            self.last_stats[key] = st
            st.raise_for_status()
        else:
            self.last_stats[key] = stats_words

    def solve_plan(self, z, integration_times=None, reverse=False, reset_counter=True):
        """What one solve of this block needs, as plain values: (net, t0, t1, opts, e).  Also resets the evaluation
        counter / Hutchinson probe the way the reference's CNF.forward does (before_odeint); reset_counter=False
        drops the probe only (no device work: for callers that keep the counter themselves)."""
        if not z.is_cuda:
            raise _native.CnfError("CNF.forward needs tensors on a HIP device (got %s): the integrator is a HIP kernel "
                                   "and has no CPU fallback" % z.device)
        if integration_times is None:
            t0, t1 = 0.0, self._end_time()
        else:
            if len(integration_times) < 2:
                raise ValueError("integration_times needs at least two times")
            # more than two times: forward() passes the whole list to cnf_solve_fwd_times; (t0, t1) are its ends
            t0, t1 = float(integration_times[0]), float(integration_times[-1])
        if reverse:
            t0, t1 = t1, t0
        inner = _inner(self.odefunc)
        if reset_counter:
            self.odefunc.before_odeint()
        else:
            inner._e = None
        if self.training:
            solver, atol, rtol, options = self.solver, self.atol, self.rtol, self.solver_options
        else:
            solver, atol, rtol, options = self.test_solver, self.test_atol, self.test_rtol, {}
        if self.adjoint_error_control not in ("state_only", "all"):
            raise ValueError("adjoint_error_control must be 'state_only' or 'all' (got %r)" % (self.adjoint_error_control,))
        flags = (_native.FLAG_SPREAD if self.spread_adjoint else 0) | \
                (_native.FLAG_ADJ_CONTROL_ALL if self.adjoint_error_control == "all" else 0)
        opts = _native.make_opts(solver, rtol, atol, options.get("step_size"), self.tile_rows, options.get("max_num_steps", 0),
                                 flags=flags)
        return inner.net_desc(), t0, t1, opts, inner.probe(z)

    def forward(self, z, logpz=None, integration_times=None, reverse=False):
        net, t0, t1, opts, e = self.solve_plan(z, integration_times, reverse)
        _logpz = torch.zeros(z.shape[0], 1, device=z.device, dtype=z.dtype) if logpz is None else logpz
        flat = _inner(self.odefunc).diffeq.flat_parameters()
        if integration_times is not None and len(integration_times) > 2:
            # the reference returns the states stacked over t then (`state_t = odeint(...)`, no `[1]`)
            times = [float(t) for t in integration_times]
            if reverse:
                times = times[::-1]   # _flip(integration_times, 0)
            z_t, logpz_t = _SolveTimes.apply(z, _logpz, flat, self, times, opts, net, e)
            self.regularization_states = ()
        elif self.training and self.nreg > 0:
            # the reference integrates the regularisers' states in training mode only (its test branch passes two states)
            reg = _native.reg_desc(self.odefunc.reg_kinds)
            z_t, logpz_t, regs = _SolveReg.apply(z, _logpz, flat, self, t0, t1, opts, net, e, reg)
            self.regularization_states = tuple(regs[i] for i in range(self.nreg))
        else:
            z_t, logpz_t = _Solve.apply(z, _logpz, flat, self, t0, t1, opts, net, e)
            self.regularization_states = ()
        return (z_t, logpz_t) if logpz is not None else z_t

    def _end_time(self):
        """T = sqrt_end_time**2 as a host float (fp32 product, like the reference's torch.tensor([0, s*s]).to(z)).
        Read back from the device only when the tensor changed: the value is detached from the graph, so it never
        trains, and a read-back per solve would be a host sync."""
        s = self.sqrt_end_time
        key = (s._version, s.data_ptr())
        if getattr(self, "_end_time_key", None) != key:
            self._end_time_val = float(s.detach() * s.detach())
            self._end_time_key = key
        return self._end_time_val

    def get_regularization_states(self):
        reg_states = self.regularization_states
        self.regularization_states = None
        return reg_states

    def num_evals(self):
        return self.odefunc._num_evals.item()
