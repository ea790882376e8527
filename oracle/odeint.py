"""CPU restatement of torchdiffeq==0.0.1 `odeint` / `odeint_adjoint` (test oracle; see oracle/__init__.py).

The reference pins the dependency at /root/reference/requirements.txt:17 and reaches it through the
(un-vendored) ffjord submodule; the algorithm restated here is the one recorded in SURVEY.md Appendix B
(dopri5 controller, fixed-grid rk4) and Appendix D (adjoint).  States are TUPLES of tensors, exactly one
scalar time and one scalar step are shared by the whole batch.

Facts of the original that matter for parity and are kept:
  * adaptive solvers keep `t`/`dt` as float64 scalars, but every stage time and every product
    `dt * beta` is rounded to the state dtype before it touches the state (Appendix B.3);
  * the error ratio of a state tensor is the MEAN over all its elements (whole batch) of
    (err / (atol + rtol*max(|y0|,|y1|)))**2; a step is accepted iff every ratio <= 1 (B.4);
  * tolerances given as lists are zipped with the state tuple, so a list shorter than the state
    leaves the trailing tensors out of error control -- this is how the adjoint pass ends up
    controlling only (z, logp) (Appendix D.3);
  * the step is never clipped to the end time; the output at T is the quartic interpolant (B.6);
  * decreasing time spans are integrated as increasing ones with time and function negated.

One documented deviation: the initial-step heuristic receives `rtol[0]`/`atol[0]` as scalars.  Applied to
all six tensors of the adjoint state it divides by the identically-zero dynamics of the logp adjoint, so
`init_tensors` (default: as many tensors as there are tolerances) limits it to the controlled tensors.
"""
import math

import torch

# ----------------------------------------------------------------------------------------------
# Dormand-Prince 5(4), Shampine's tableau
# ----------------------------------------------------------------------------------------------
DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
DP_BETA = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
DP_C_SOL = DP_BETA[-1] + (0.0,)
DP_C_ERR = (
    35 / 384 - 1951 / 21600,
    0.0,
    500 / 1113 - 22642 / 50085,
    125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400,
    11 / 84 - 649 / 6300,
    -1.0 / 60.0,
)
DP_C_MID = (
    6025192743 / 30085553152 / 2,
    0.0,
    51252292925 / 65400821598 / 2,
    -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2,
    -1776094331 / 19743644256 / 2,
    11237099 / 235043384 / 2,
)


def _rms(x):
    return x.norm() / math.sqrt(x.numel())


def _weighted(dt, coeffs, ks):
    """sum_j (dt*c_j)*k_j, accumulated left to right starting from python 0 (as the original does)."""
    acc = 0
    for c, k in zip(coeffs, ks):
        acc = acc + (dt * c) * k
    return acc


def _as_list(tol, n):
    return list(tol) if isinstance(tol, (list, tuple)) else [tol] * n


class _Negated:
    """f(t, y) -> -f(-t, y): lets a decreasing span be integrated as an increasing one."""

    def __init__(self, f):
        self.f = f

    def __call__(self, t, y):
        return tuple(-v for v in self.f(-t, y))


def initial_step(func, t0, y0, order, rtol, atol, f0, n_tensors=None):
    """Hairer-style first step (Appendix B.2).  rtol/atol are scalars here.  Costs one evaluation."""
    n = len(y0) if n_tensors is None else n_tensors
    t0 = t0.to(y0[0])
    scale = [atol + y.abs() * rtol for y in y0[:n]]
    d0 = [_rms(y / s) for y, s in zip(y0, scale)]
    d1 = [_rms(f / s) for f, s in zip(f0, scale)]
    if max(d0).item() < 1e-5 or max(d1).item() < 1e-5:
        h0 = torch.tensor(1e-6).to(t0)
    else:
        h0 = 0.01 * max(a / b for a, b in zip(d0, d1))
    y1 = tuple(y + h0 * f for y, f in zip(y0, f0))
    f1 = func(t0 + h0, y1)
    d2 = [_rms((b - a) / s) / h0 for a, b, s in zip(f0, f1, scale)]
    if max(d1).item() <= 1e-15 and max(d2).item() <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6).to(h0), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1 + d2)) ** (1.0 / float(order + 1))
    return torch.min(100 * h0, h1)


class Dopri5:
    """Adaptive Dormand-Prince with FSAL and dense output (Appendix B.2-B.6)."""

    order = 5

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, init_tensors=None, stats=None, **_ignored):
        self.func, self.y0 = func, y0
        self.rtol, self.atol = _as_list(rtol, len(y0)), _as_list(atol, len(y0))
        self.first_step = first_step
        f64 = dict(dtype=torch.float64, device=y0[0].device)
        self.safety = torch.tensor(safety, **f64)
        self.ifactor = torch.tensor(ifactor, **f64)
        self.dfactor = torch.tensor(dfactor, **f64)
        self.max_num_steps = max_num_steps
        self.init_tensors = min(len(self.rtol), len(y0)) if init_tensors is None else init_tensors
        self.stats = stats if stats is not None else {}
        self.stats.update(n_accepted=0, n_rejected=0, dts=[])

    # -- one attempt -----------------------------------------------------------------------------
    def _rk_step(self, y0, f0, t0, dt):
        dtype = y0[0].dtype
        t0 = t0.to(dtype)
        dt = dt.to(dtype)
        ks = [[f] for f in f0]
        yi = y0
        for alpha, beta in zip(DP_ALPHA, DP_BETA):
            ti = t0 + alpha * dt
            yi = tuple(y + _weighted(dt, beta, k) for y, k in zip(y0, ks))
            for k, f in zip(ks, self.func(ti, yi)):
                k.append(f)
        y1 = yi                                        # c_sol == last beta row (+0): y1 is the 6th stage point
        f1 = tuple(k[-1] for k in ks)
        err = tuple(_weighted(dt, DP_C_ERR, k) for k in ks)
        return y1, f1, err, ks

    def _ratios(self, err, y0, y1):
        out = []
        for e, a, r, u, v in zip(err, self.atol, self.rtol, y0, y1):   # zip truncates to len(tolerances)
            q = e / (a + r * torch.max(u.abs(), v.abs()))
            out.append(torch.mean(q * q))
        return out

    def _next_dt(self, dt, ratios):
        worst = max(ratios)
        if worst == 0:
            return dt * self.ifactor
        dfactor = torch.tensor(1.0, dtype=torch.float64) if worst < 1 else self.dfactor
        root = torch.sqrt(worst).to(dt)
        expo = torch.tensor(1 / self.order).to(dt)
        factor = torch.max(1 / self.ifactor, torch.min(root ** expo / self.safety, 1 / dfactor))
        return dt / factor

    @staticmethod
    def _fit(y0, y1, ks, dt):
        dt = dt.to(y0[0])
        coeffs = []
        for u, v, k in zip(y0, y1, ks):
            mid = u + _weighted(dt, DP_C_MID, k)
            f0, f1 = k[0], k[-1]
            a = (-2 * dt) * f0 + (2 * dt) * f1 + -8 * u + -8 * v + 16 * mid
            b = (5 * dt) * f0 + (-3 * dt) * f1 + 18 * u + 14 * v + -32 * mid
            c = (-4 * dt) * f0 + dt * f1 + -11 * u + -5 * v + 16 * mid
            coeffs.append((a, b, c, dt * f0, u))
        return coeffs

    @staticmethod
    def _evaluate(coeffs, t0, t1, t):
        dtype = coeffs[0][0].dtype
        t0, t1, t = t0.to(dtype), t1.to(dtype), t.to(dtype)
        assert bool((t0 <= t) & (t <= t1)), "invalid interpolation, fails `t0 <= t <= t1`"
        x = ((t - t0) / (t1 - t0)).to(dtype)
        x2 = x * x
        x3 = x2 * x
        x4 = x3 * x
        return tuple(a * x4 + b * x3 + c * x2 + d * x + e for a, b, c, d, e in coeffs)

    # -- driver ----------------------------------------------------------------------------------
    def integrate(self, t):
        assert bool((t[1:] > t[:-1]).all()), "t must be strictly increasing or decreasing"
        t = t.to(self.y0[0].device, torch.float64)
        y0 = self.y0
        f0 = self.func(t[0].to(y0[0]), y0)
        if self.first_step is None:
            dt = initial_step(self.func, t[0], y0, 4, self.rtol[0], self.atol[0], f0, self.init_tensors).to(t)
        else:
            dt = torch.as_tensor(self.first_step).to(t)
        t_lo = t_hi = t[0]
        coeffs = [(y,) * 5 for y in y0]
        out = [y0]
        for target in t[1:]:
            n = 0
            while target > t_hi:
                assert n < self.max_num_steps, "max_num_steps exceeded ({}>={})".format(n, self.max_num_steps)
                assert t_hi + dt > t_hi, "underflow in dt {}".format(dt.item())
                for y in y0:
                    assert bool(torch.isfinite(y).all()), "non-finite values in state `y`: {}".format(y)
                y1, f1, err, ks = self._rk_step(y0, f0, t_hi, dt)
                ratios = self._ratios(err, y0, y1)
                accept = all(bool(r <= 1) for r in ratios)
                self.stats["dts"].append(float(dt))
                if accept:
                    coeffs = self._fit(y0, y1, ks, dt)
                    t_lo, t_hi = t_hi, t_hi + dt
                    y0, f0 = y1, f1
                    self.stats["n_accepted"] += 1
                else:
                    self.stats["n_rejected"] += 1
                dt = self._next_dt(dt, ratios)
                n += 1
            out.append(self._evaluate(coeffs, t_lo, t_hi, target))
        return tuple(torch.stack(c) for c in zip(*out))


def _rk4_38(func, t, dt, y):
    k1 = func(t, y)
    k2 = func(t + dt / 3, tuple(a + dt * b / 3 for a, b in zip(y, k1)))
    k3 = func(t + dt * 2 / 3, tuple(a + dt * (b / -3 + c) for a, b, c in zip(y, k1, k2)))
    k4 = func(t + dt, tuple(a + dt * (b - c + d) for a, b, c, d in zip(y, k1, k2, k3)))
    return tuple((b + 3 * c + 3 * d + e) * (dt / 8) for b, c, d, e in zip(k1, k2, k3, k4))


def _euler(func, t, dt, y):
    return tuple(dt * f for f in func(t, y))


def _midpoint(func, t, dt, y):
    half = tuple(a + f * (dt / 2) for a, f in zip(y, func(t, y)))
    return tuple(dt * f for f in func(t + dt / 2, half))


def fixed_grid(t, step_size=None):
    """The time grid of the fixed-step solvers (Appendix B.6): multiples of step_size, last clipped to t[-1]."""
    if step_size is None:
        return t
    n = torch.ceil((t[-1] - t[0]) / step_size + 1).item()
    grid = torch.arange(0, n).to(t) * step_size + t[0]
    if grid[-1] > t[-1]:
        grid[-1] = t[-1]
    return grid


class FixedGrid:
    def __init__(self, step, func, y0, step_size=None, stats=None, **_ignored):
        self.step, self.func, self.y0, self.step_size = step, func, y0, step_size
        self.stats = stats if stats is not None else {}

    def integrate(self, t):
        assert bool((t[1:] > t[:-1]).all())
        t = t.to(self.y0[0])
        grid = fixed_grid(t, self.step_size).to(self.y0[0])
        assert grid[0] == t[0] and grid[-1] == t[-1]
        out, j, y = [self.y0], 1, self.y0
        for a, b in zip(grid[:-1], grid[1:]):
            prev = y
            y = tuple(u + du for u, du in zip(y, self.step(self.func, a, b - a, y)))
            while j < len(t) and b >= t[j]:
                if t[j] == b:
                    out.append(y)
                else:
                    out.append(tuple(u + (v - u) / (b - a) * (t[j] - a) for u, v in zip(prev, y)))
                j += 1
        self.stats["n_steps"] = len(grid) - 1
        return tuple(torch.stack(c) for c in zip(*out))


SOLVERS = {
    "dopri5": Dopri5,
    "rk4": lambda *a, **k: FixedGrid(_rk4_38, *a, **k),
    "euler": lambda *a, **k: FixedGrid(_euler, *a, **k),
    "midpoint": lambda *a, **k: FixedGrid(_midpoint, *a, **k),
}


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """Integrate dy/dt = func(t, y) and return y at every time of `t` (stacked along dim 0).

    `y0` is a tensor or a tuple of tensors; `func(t, y)` returns the same structure.
    """
    single = torch.is_tensor(y0)
    if single:
        y0 = (y0,)
        inner = func
        func = lambda t_, y_: (inner(t_, y_[0]),)  # noqa: E731
    assert all(torch.is_floating_point(y) for y in y0) and torch.is_floating_point(t)
    if bool((t[1:] < t[:-1]).all()):
        t, func = -t, _Negated(func)
    opts = dict(options or {})
    method = method or "dopri5"
    if method == "dopri5":
        sol = Dopri5(func, y0, rtol=rtol, atol=atol, **opts).integrate(t)
    else:
        sol = SOLVERS[method](func, y0, **opts).integrate(t)
    return sol[0] if single else sol


# ----------------------------------------------------------------------------------------------
# adjoint (Appendix D)
# ----------------------------------------------------------------------------------------------
def _flat(tensors, like):
    parts = [torch.zeros_like(p).reshape(-1) if g is None else g.reshape(-1) for g, p in zip(tensors, like)]
    return torch.cat(parts) if parts else torch.zeros(0)


class _Adjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, n_state, t, rtol, atol, method, options, stats, control, flat_params, *y0):
        ctx.func, ctx.rtol, ctx.atol, ctx.method, ctx.options, ctx.stats = func, rtol, atol, method, options, stats
        ctx.control = control
        with torch.no_grad():
            ans = odeint(func, tuple(y0), t, rtol=rtol, atol=atol, method=method,
                         options=dict(options or {}, stats=stats.setdefault("forward", {})))
        ctx.save_for_backward(t, flat_params, *ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_out):
        t, flat_params, *ans = ctx.saved_tensors
        func = ctx.func
        n = len(ans)
        params = tuple(p for p in func.parameters() if p.requires_grad)

        def augmented(time, aug):
            y, adj = aug[:n], aug[n:2 * n]
            with torch.enable_grad():
                time = time.to(y[0].device).detach().requires_grad_(True)
                y = tuple(v.detach().requires_grad_(True) for v in y)
                f = func(time, y)
                g_t, *g_rest = torch.autograd.grad(f, (time,) + y + params, tuple(-a for a in adj),
                                                   allow_unused=True, retain_graph=True)
            g_y = tuple(torch.zeros_like(v) if g is None else g for g, v in zip(g_rest[:n], y))
            g_t = torch.zeros_like(time) if g_t is None else g_t
            g_p = _flat(g_rest[n:], params).to(y[0])
            return tuple(v.detach() for v in f) + g_y + (g_t, g_p)

        with torch.no_grad():
            adj_y = tuple(g[-1] for g in grad_out)
            adj_p = torch.zeros_like(flat_params)
            adj_t = torch.tensor(0.0).to(t)
            time_grads = []
            for i in range(ans[0].shape[0] - 1, 0, -1):
                y_i = tuple(a[i] for a in ans)
                f_i = func(t[i], y_i)
                dL_dt = sum(torch.dot(f.reshape(-1), g[i].reshape(-1)).reshape(1) for f, g in zip(f_i, grad_out))
                adj_t = adj_t - dL_dt
                time_grads.append(dL_dt)
                aug0 = y_i + adj_y + (adj_t, adj_p)
                # which tensors of the augmented state the controller sees (Appendix D.3).  The solver zips its
                # tolerance lists with the state tuple, so the LENGTH of the lists is the switch: n = the forward
                # state only (what CNF's lists do, 'state_only'), 2n = + the per-row adjoints ('rows', the kernels'
                # CNF_FLAG_ADJ_CONTROL_ALL), 2n + 2 = everything, as scalar tolerances would ('all')
                n_ctl = {"state_only": None, "rows": 2 * n, "all": 2 * n + 2}[ctx.control]
                b_rtol, b_atol = ctx.rtol, ctx.atol
                if n_ctl is not None:
                    b_rtol = [_as_list(ctx.rtol, 1)[0]] * n_ctl
                    b_atol = [_as_list(ctx.atol, 1)[0]] * n_ctl
                sol = odeint(augmented, aug0, torch.stack([t[i], t[i - 1]]), rtol=b_rtol, atol=b_atol,
                             method=ctx.method,
                             options=dict(ctx.options or {}, stats=ctx.stats.setdefault("backward", {})))
                adj_y = tuple(s[1] + g[i - 1] for s, g in zip(sol[n:2 * n], grad_out))
                adj_t, adj_p = sol[2 * n][1], sol[2 * n + 1][1]
            time_grads.append(adj_t.reshape(1))
        return (None, None, torch.cat(time_grads[::-1]), None, None, None, None, None, None, adj_p) + adj_y


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None, stats=None,
                   adjoint_control="state_only"):
    """`odeint` whose backward integrates the adjoint ODE instead of differentiating the solver.

    `func` must be an nn.Module (its parameters receive gradients).  `stats` (a dict) receives the
    forward / backward step counters.  `adjoint_control`: 'state_only' (default: the tolerances as passed, i.e. for
    CNF's length-2 lists error control over the forward state only), 'rows' (+ the adjoints of the state tensors) or
    'all' (+ the time and parameter adjoints) -- see _Adjoint.backward.  For 'rows' / 'all' the initial-step heuristic
    runs over the same tensors with the original's arithmetic: the adjoint of logp has identically zero dynamics, so
    its d0 / d1 is infinite, h0 = inf, the second evaluation sees a non-finite state, and python's max() -- which skips
    NaNs that are not its first argument -- leaves (0.01 / max(d1))^(1/5) as the first step.
    """
    assert adjoint_control in ("state_only", "rows", "all"), adjoint_control
    assert isinstance(func, torch.nn.Module), "func must be an nn.Module (parameters are needed)"
    single = torch.is_tensor(y0)
    if single:
        y0 = (y0,)
        inner = func

        class _Tup(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.inner = inner

            def forward(self, t_, y_):
                return (self.inner(t_, y_[0]),)

        func = _Tup()
    flat_params = _flat([p for p in func.parameters() if p.requires_grad],
                        [p for p in func.parameters() if p.requires_grad])
    stats = {} if stats is None else stats
    sol = _Adjoint.apply(func, len(y0), t, rtol, atol, method, options, stats, adjoint_control, flat_params, *y0)
    return sol[0] if single else sol
