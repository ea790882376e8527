"""CPU: the oracle against everything the reference holds for this path:
trained-checkpoint weights, the stored NFE counter, the analytic training density, Appendix C."""
import numpy as np
import pytest
import torch

import oracle
from gan_flow_amd.gu import GausUniffMixture
from _helpers import appendix_c, fixture_state_dict, oracle_fixture_model


@pytest.mark.parametrize("which", ["unimodal", "multimodal"])
def test_state_dict_names_match_checkpoint(which):
    sd = fixture_state_dict(which)
    m = oracle.build_cnf(1)
    assert set(m.state_dict().keys()) == set(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert abs(float(sd["chain.0.sqrt_end_time"]) - 0.70710677) < 1e-7
    assert sum(v.numel() for k, v in sd.items() if "diffeq" in k) == 9092


@pytest.mark.parametrize("which", ["unimodal", "multimodal"])
def test_field_known_answers(which):
    m = oracle_fixture_model(which)
    fn = m.chain[0].odefunc.odefunc
    for t, z, f_ref, dfdz_ref in appendix_c()["field"][which]:
        f, neg_div = fn(torch.tensor(t), (torch.tensor([[z]]), torch.zeros(1, 1)))
        assert abs(f.item() - f_ref) <= 1e-5 * max(1.0, abs(f_ref))
        assert abs(-neg_div.item() - dfdz_ref) <= 1e-5 * max(1.0, abs(dfdz_ref))


def test_stored_nfe_unimodal_is_reproduced_exactly():
    """The checkpoint's `_num_evals` (116 = 2 + 6*(15 accepted + 4 rejected)) was left by a 10 000-row sampling solve
    (model_plot.py:97); dopri5(1e-5) on 10 000 N(0,1) draws must give the same count."""
    m = oracle_fixture_model("unimodal")
    sample_fn, _ = oracle.get_transforms(m)
    for seed in (0, 1):
        z = torch.randn(10000, 1, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            sample_fn(z)
        st = m.chain[0].stats["forward"]
        assert m.chain[0].num_evals() == appendix_c()["stored_nfe"]["unimodal"]
        assert (st["n_accepted"], st["n_rejected"]) == (15, 4)


def test_stored_nfe_multimodal_is_close():
    m = oracle_fixture_model("multimodal")
    sample_fn, _ = oracle.get_transforms(m)
    z = torch.randn(10000, 1, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        sample_fn(z)
    assert abs(m.chain[0].num_evals() - appendix_c()["stored_nfe"]["multimodal"]) <= 60  # draw-dependent (stiff field)


@pytest.mark.parametrize("which", ["unimodal", "multimodal"])
def test_solves_match_tight_reference(which):
    kat = appendix_c()
    m = oracle_fixture_model(which)
    sample_fn, density_fn = oracle.get_transforms(m)
    x = torch.tensor(kat["forward"][which]["x"]).reshape(-1, 1)
    with torch.no_grad():
        z, dl = density_fn(x, torch.zeros_like(x))
    bulk = torch.tensor(kat["forward"][which]["z"]).abs() < 4
    tol = 5e-4 if which == "unimodal" else 3e-3   # fp32 dopri5 at 1e-5 with a 5-row batch norm; stiff multimodal field
    assert (z.ravel() - torch.tensor(kat["forward"][which]["z"])).abs()[bulk].max() < tol
    logpx = (oracle.standard_normal_logprob(z) - dl).ravel()
    assert (logpx - torch.tensor(kat["forward"][which]["logpx"])).abs()[bulk].max() < tol
    zz = torch.tensor(kat["reverse"][which]["z"], dtype=torch.float32).reshape(-1, 1)
    with torch.no_grad():
        xx, dl2 = sample_fn(zz, torch.zeros_like(zz))
    scale = 1.0 if which == "unimodal" else 80.0
    assert (xx.ravel() - torch.tensor(kat["reverse"][which]["x"])).abs().max() < 1e-3 * scale
    assert (dl2.ravel() - torch.tensor(kat["reverse"][which]["delta_logp"])).abs().max() < (2e-3 if which == "unimodal" else 5e-3)


@pytest.mark.parametrize("which", ["unimodal", "multimodal"])
def test_float64_tight_solves_match_appendix_c(which):
    """Same weights, float64 state, tolerances 1e-10: the oracle must land on the DOP853 values to ~1e-6."""
    kat = appendix_c()
    m = oracle_fixture_model(which, torch.float64)
    m.chain[0].test_atol = m.chain[0].test_rtol = 1e-10
    sample_fn, density_fn = oracle.get_transforms(m)
    x = torch.tensor(kat["forward"][which]["x"], dtype=torch.float64).reshape(-1, 1)
    with torch.no_grad():
        z, dl = density_fn(x, torch.zeros_like(x))
    assert (z.ravel() - torch.tensor(kat["forward"][which]["z"], dtype=torch.float64)).abs().max() < 2e-6
    assert (dl.ravel() - torch.tensor(kat["forward"][which]["delta_logp"], dtype=torch.float64)).abs().max() < 2e-6
    zz = torch.tensor(kat["reverse"][which]["z"], dtype=torch.float64).reshape(-1, 1)
    with torch.no_grad():
        xx, dl2 = sample_fn(zz, torch.zeros_like(zz))
    assert (xx.ravel() - torch.tensor(kat["reverse"][which]["x"], dtype=torch.float64)).abs().max() < 2e-5
    assert (dl2.ravel() - torch.tensor(kat["reverse"][which]["delta_logp"], dtype=torch.float64)).abs().max() < 2e-5


@pytest.mark.parametrize("which,lo,hi,kw", [
    ("unimodal", 3.5, 6.5, dict(n_mixture=1, mean_dist=5, sigma=0.1, unif_intsect=5, unif_ratio=3)),
    ("multimodal", -10.0, 100.0, dict(n_mixture=8, mean_dist=10, sigma=2, unif_intsect=1.5, unif_ratio=1.0)),
])
def test_model_density_normalises_and_matches_training_density(which, lo, hi, kw):
    m = oracle_fixture_model(which)
    _, density_fn = oracle.get_transforms(m)
    xs = torch.linspace(lo, hi, 2001).reshape(-1, 1)
    with torch.no_grad():
        z, dl = density_fn(xs, torch.zeros_like(xs))
    p = torch.exp(oracle.standard_normal_logprob(z) - dl).ravel().double().numpy()
    grid = xs.detach().ravel().double().numpy()
    integral = np.trapz(p, grid)
    assert abs(integral - 1.0) < 2e-3
    true = GausUniffMixture(**kw).density(grid)
    l1 = np.trapz(np.abs(p - true), grid)
    assert l1 < 0.15      # SURVEY probe P1: 0.034 (unimodal), 0.109 (multimodal)


def test_rk4_fixed_grid_matches_dopri5():
    m = oracle_fixture_model("unimodal")
    x = torch.linspace(4.6, 5.4, 9).reshape(-1, 1)
    with torch.no_grad():
        z_ref, l_ref = m(x, torch.zeros_like(x))
        blk = m.chain[0]
        blk.test_solver = "rk4"
        z4, l4 = oracle.odeint(blk.odefunc, (x, torch.zeros_like(x)), torch.tensor([0.0, 0.5]), method="rk4",
                               options=dict(step_size=0.01))
    assert (z4[1] - z_ref).abs().max() < 1e-3 and (l4[1] - l_ref).abs().max() < 1e-3


def test_adjoint_gradients_match_autograd_through_unrolled_rk4():
    """Appendix D has no reference artefact: check the adjoint against plain backprop through a fine fixed grid."""
    torch.manual_seed(0)
    m = oracle.build_cnf(2).double()
    x = torch.randn(8, 2, dtype=torch.float64) * 0.5
    blk = m.chain[0]
    blk.atol = blk.rtol = 1e-9

    def loss_of(z, dl):
        return -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()

    m.train()
    z, dl = m(x, torch.zeros(8, 1, dtype=torch.float64))
    m.zero_grad()
    loss_of(z, dl).backward()
    g_adj = torch.cat([p.grad.reshape(-1) for p in blk.odefunc.parameters()])
    m.zero_grad()
    zs, ls = oracle.odeint(blk.odefunc, (x, torch.zeros(8, 1, dtype=torch.float64)),
                           torch.tensor([0.0, 0.5], dtype=torch.float64), method="rk4", options=dict(step_size=0.005))
    loss_of(zs[1], ls[1]).backward()
    g_bp = torch.cat([p.grad.reshape(-1) for p in blk.odefunc.parameters()])
    assert (g_adj - g_bp).abs().max() / g_bp.abs().max() < 1e-5


@pytest.mark.parametrize("solver,step,bar", [("rk4", 0.02, 1e-6), ("midpoint", 0.005, 2e-4), ("euler", 0.001, 5e-3)])
def test_fixed_grid_adjoint_converges_to_backprop_through_the_same_solver(solver, step, bar):
    """The oracle's fixed-grid adjoint (odeint_adjoint with method = euler | midpoint | rk4: what the kernels' fixed-grid
    adjoint is held against) has no reference artefact either.  It integrates the adjoint ODE with the forward's method, which
    is not the exact gradient of the discretised forward solve but converges to it with the method's order: float64, the
    same grid, adjoint gradients against plain backprop through the unrolled solver."""
    torch.manual_seed(2)
    m = oracle.build_cnf(2, solver=solver).double()
    x = torch.randn(8, 2, dtype=torch.float64) * 0.5
    blk = m.chain[0]
    blk.solver_options = {"step_size": step}

    def loss_of(z, dl):
        return -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()

    m.train()
    xa = x.clone().requires_grad_(True)
    z, dl = m(xa, torch.zeros(8, 1, dtype=torch.float64))
    m.zero_grad()
    loss_of(z, dl).backward()
    g_adj = torch.cat([p.grad.reshape(-1) for p in blk.odefunc.parameters()])
    gx_adj = xa.grad.clone()
    assert blk.stats["backward"]["n_steps"] == blk.stats["forward"]["n_steps"] == int(np.ceil(0.5 / step - 1e-9))
    m.zero_grad()
    xb = x.clone().requires_grad_(True)
    zs, ls = oracle.odeint(blk.odefunc, (xb, torch.zeros(8, 1, dtype=torch.float64)),
                           torch.tensor([0.0, float(blk.sqrt_end_time * blk.sqrt_end_time)]).double(), method=solver,   # fp32 end time, as CNF.forward forms it
                           options=dict(step_size=step))
    loss_of(zs[1], ls[1]).backward()
    g_bp = torch.cat([p.grad.reshape(-1) for p in blk.odefunc.parameters()])
    assert (z - zs[1]).abs().max() < 1e-12 and (dl - ls[1]).abs().max() < 1e-12
    assert (g_adj - g_bp).abs().max() / g_bp.abs().max() < bar
    assert (gx_adj - xb.grad).abs().max() / xb.grad.abs().max() < bar


def test_adjoint_error_control_modes():
    """SURVEY Appendix D.3's switch in the oracle: 'state_only' (CNF's length-2 tolerance lists, zipped with the state
    tuple), 'rows' (+ a_z, a_logp; the kernels' CNF_FLAG_ADJ_CONTROL_ALL) and 'all' (+ a_t, a_theta; what scalar tolerances
    would do).  All three integrate the same adjoint ODE: gradients agree to solver accuracy; what changes is the step
    sequence.  For 'rows' / 'all' the initial-step heuristic meets a_logp's identically-zero dynamics (d0 / d1 = inf) and,
    with python's NaN-skipping max(), leaves (0.01 / max d1)^(1/5) as the first step -- checked here against that formula.
    Parity unpinned: no file of the reference records backward step counts."""
    torch.manual_seed(2)
    m = oracle.build_cnf(2).train()
    blk = m.chain[0]
    x = torch.randn(64, 2)
    grads, stats = {}, {}
    for mode in ("state_only", "rows", "all"):
        blk.adjoint_error_control = mode
        m.zero_grad()
        z, dl = m(x, torch.zeros(64, 1))
        (-(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()).backward()
        grads[mode] = torch.cat([p.grad.reshape(-1) for p in blk.odefunc.parameters()])
        stats[mode] = dict(blk.stats["backward"])
    for mode in ("rows", "all"):
        assert (grads[mode] - grads["state_only"]).abs().max() < 2e-4 * grads["state_only"].abs().max()
        assert stats[mode]["n_accepted"] >= stats["state_only"]["n_accepted"]
    # first step of 'rows': (0.01 / max(d1z, d1l, d1a))^(1/5), scales atol + rtol |y|, rms over each tensor
    with torch.no_grad():
        z1, dl1 = z.detach(), dl.detach()
        a_z, a_l = z1 / 64, torch.full((64, 1), 1.0 / 64)
    fn = blk.odefunc
    fn.before_odeint()
    with torch.enable_grad():
        zz = z1.clone().requires_grad_(True)
        f, nd = fn(torch.tensor(0.5), (zz, dl1.clone().requires_grad_(True)))
        vz, = torch.autograd.grad((f, nd), zz, (-a_z, -a_l))
    rms = lambda v, y: float((v / (1e-5 + 1e-5 * y.abs())).pow(2).mean().sqrt())   # noqa: E731
    d1 = max(rms(-f.detach(), z1), rms(-nd.detach(), dl1), rms(-vz, a_z))   # reversed time: slopes change sign
    assert abs(stats["rows"]["dts"][0] - (0.01 / d1) ** 0.2) < 1e-5 * stats["rows"]["dts"][0]
    assert stats["rows"]["dts"][0] != stats["state_only"]["dts"][0]


def test_regularised_dynamics_oracle_is_self_consistent():
    """RegularizedODEfunc with the l1 / l2 kinetic regularisers (oracle.l1_regularzation_fn / l2_regularzation_fn, restated as
    recalled; parity unpinned): the regularisation states are integrals of batch means, and odeint_adjoint carries their
    constant adjoints.  float64, tolerances 1e-12: (i) the states equal a fine quadrature of the integrands along the
    solution; (ii) d loss / d x from the adjoint equals central finite differences to 1e-3 (1e-8 on most rows, 2e-4 on a row
    that crosses a kink); (iii) d loss / d theta agrees with backprop through the same solve to a few 1e-3 -- a_z and a_theta are not under error control (the tolerance lists cover
    (z, logp, r) only) and the regularisers' integrands have kinks, so the adjoint's quadrature is only as good as the steps
    the smooth state needed (without regularisers the same comparison gives 1e-8)."""
    from _helpers import oracle_random_model
    fns = (oracle.l1_regularzation_fn, oracle.l2_regularzation_fn)
    m = oracle_random_model(2, seed=3, scale=1.5, regularization_fns=fns).double()
    m.train()
    x = torch.randn(32, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    blk = m.chain[0]
    blk.rtol = blk.atol = 1e-12
    T = float(blk.sqrt_end_time.detach() ** 2)
    zero = torch.zeros(32, 1, dtype=torch.float64)

    def loss_of(xv):
        z, dl = m(xv, zero)
        r = blk.get_regularization_states()
        return -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean() + 0.3 * r[0] + 0.7 * r[1], r

    m.zero_grad()
    xx = x.clone().requires_grad_(True)
    loss, regs = loss_of(xx)
    loss.backward()
    params = list(blk.odefunc.odefunc.diffeq.parameters())
    g_adj = [p.grad.clone() for p in params]
    # (i) quadrature of the integrands along a fine rk4 solution
    fn = blk.odefunc.odefunc
    n = 400
    h = T / n
    z = x.clone()
    acc = torch.zeros(2, dtype=torch.float64)
    with torch.no_grad():
        def field(t, zz):
            with torch.enable_grad():
                return fn(torch.tensor(t, dtype=torch.float64), (zz.clone(), zero))[0].detach()
        def integrands(dx):
            return torch.stack([dx.abs().mean(), (dx.norm(dim=1) / 2 ** 0.5).mean()])
        for i in range(n):
            t = i * h
            k1 = field(t, z); k2 = field(t + h / 2, z + h / 2 * k1); k3 = field(t + h / 2, z + h / 2 * k2); k4 = field(t + h, z + h * k3)
            zn = z + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
            acc += h / 6 * (integrands(k1) + 4 * integrands(field(t + h / 2, (z + zn) / 2 + h / 8 * (k1 - k4))) + integrands(k4))
            z = zn
    assert abs(float(regs[0]) - acc[0].item()) < 1e-6 and abs(float(regs[1]) - acc[1].item()) < 1e-6
    # (ii) finite differences in x
    eps = 1e-4
    with torch.no_grad():
        for (i, j) in ((0, 0), (7, 1)):
            xp, xm = x.clone(), x.clone()
            xp[i, j] += eps
            xm[i, j] -= eps
            fd = (loss_of(xp)[0] - loss_of(xm)[0]).item() / (2 * eps)
            assert abs(fd - xx.grad[i, j].item()) < 1e-3 * abs(fd), (i, j, fd, xx.grad[i, j].item())
    # (iii) backprop through the same adaptive solve
    m.zero_grad()
    blk.odefunc.before_odeint()
    st = oracle.odeint(blk.odefunc, (x, zero, torch.tensor(0.0, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64)),
                       torch.tensor([0.0, T], dtype=torch.float64), rtol=[1e-12, 1e-12, 1e20, 1e20], atol=[1e-12, 1e-12, 1e20, 1e20],
                       method="dopri5")
    l2 = -(oracle.standard_normal_logprob(st[0][1]).sum(1, keepdim=True) - st[1][1]).mean() + 0.3 * st[2][1] + 0.7 * st[3][1]
    assert abs(l2.item() - loss.item()) < 1e-7
    l2.backward()
    for p, ga in zip(params, g_adj):
        assert (p.grad - ga).abs().max() < 1e-2 * p.grad.abs().max()
