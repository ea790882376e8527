"""GPU parity tests of the hidden-width-256 forward path (csrc/cnf_wide.hip; BASELINE config 4: stacked CNF blocks,
H = (256, 256, 256)).  Same bars as tests/test_gpu_forward.py: one evaluation 1e-5 relative to the oracle, rk4
z / logp 1e-5 relative, dopri5 identical accepted / rejected counts.  All calls go through the C ABI."""
import pytest
import torch

import oracle
from _helpers import oracle_random_model, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = (256, 256, 256)


def _native():
    from gan_flow_amd import _native
    _native.load_library()
    return _native


def _fn(m, i=0):
    return m.chain[i].odefunc.odefunc


def _flat(m, i=0):
    return torch.cat([p.detach().reshape(-1) for p in _fn(m, i).diffeq.parameters()]).float()


def _model(dim, hutch, seed=3, scale=1.5, blocks=1):
    m = oracle_random_model(dim, seed=seed, divergence_fn="approximate" if hutch else "brute_force", scale=scale,
                            hidden_dims=H, num_blocks=blocks)
    m.train()
    return m


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 2048), (1, False, 100), (2, True, 77), (2, False, 1)])
def test_wide_field_matches_oracle(dim, hutch, rows):
    nat = _native()
    m = _model(dim, hutch)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(rows, dim, generator=g) * 2
    e = torch.randn(rows, dim, generator=g) if hutch else None
    net = nat.net_desc(dim, 256, 3, hutch)
    params = _flat(m).to(DEV)
    assert params.numel() == nat.param_count(net)
    for t in (0.0, 0.37):
        fn = _fn(m)
        fn.before_odeint(e=e)
        with torch.enable_grad():
            f_ref, nd_ref = fn(torch.tensor(t), (z.clone().requires_grad_(True), torch.zeros(rows, 1)))
        f, nd = nat.eval_field(net, params, t, z.to(DEV), None if e is None else e.to(DEV))
        print("wide field d=%d hutch=%s t=%.2f: f %.1e -div %.1e" % (dim, hutch, t, rel_err(f, f_ref), rel_err(nd, nd_ref)))
        assert rel_err(f, f_ref) < 1e-5
        assert rel_err(nd, nd_ref) < 2e-5


def _solve_pair(m, x, solver, step_size=None, tile_rows=0, hutch=False, e=None, reverse=False):
    blk = m.chain[0]
    T = float(blk.sqrt_end_time.detach() * blk.sqrt_end_time.detach())
    ts = torch.tensor([T, 0.0] if reverse else [0.0, T])
    stats = {}
    blk.odefunc.before_odeint(e=e)
    opts = dict(stats=stats)
    if step_size:
        opts["step_size"] = step_size
    y0 = (x.clone(), torch.zeros(x.shape[0], 1))
    with torch.no_grad():
        if solver == "dopri5":
            z_ref, l_ref = oracle.odeint(blk.odefunc, y0, ts, rtol=[1e-5, 1e-5], atol=[1e-5, 1e-5], method="dopri5", options=opts)
        else:
            z_ref, l_ref = oracle.odeint(blk.odefunc, y0, ts, method=solver, options=opts)
    nat = _native()
    net = nat.net_desc(x.shape[1], 256, 3, hutch)
    o = nat.make_opts(solver, 1e-5, 1e-5, step_size, tile_rows)
    z, l, st = nat.solve_fwd(net, _flat(m).to(DEV), x.to(DEV), None, float(ts[0]), float(ts[1]), o, None if e is None else e.to(DEV))
    torch.cuda.synchronize()
    st = nat.Stats(st)
    st.raise_for_status()
    return (z_ref[1], l_ref[1], stats), (z.cpu(), l.cpu(), st)


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 1000), (2, True, 50), (1, False, 17)])
def test_wide_rk4_parity(dim, hutch, rows):
    m = _model(dim, hutch, seed=1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    (z_ref, l_ref, _), (z, l, st) = _solve_pair(m, x, "rk4", step_size=0.05, hutch=hutch, e=e)
    print("wide rk4 d=%d: z %.1e logp %.1e" % (dim, rel_err(z, z_ref), rel_err(l, l_ref)))
    assert rel_err(z, z_ref) < 1e-5 and rel_err(l, l_ref) < 1e-5
    assert st.nfe == 4 * st.n_accepted == 40


@pytest.mark.parametrize("dim,hutch,rows,reverse", [(2, False, 2048, False), (2, False, 333, True), (2, True, 96, False),
                                                    (1, False, 4096, False)])
def test_wide_dopri5_parity_batch_global(dim, hutch, rows, reverse):
    m = _model(dim, hutch, seed=1, scale=2.0)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    (z_ref, l_ref, so), (z, l, st) = _solve_pair(m, x, "dopri5", hutch=hutch, e=e, reverse=reverse)
    print("wide dopri5 d=%d rows=%d: hip %d+%d oracle %d+%d | max abs err z %.1e logp %.1e"
          % (dim, rows, st.n_accepted, st.n_rejected, so["n_accepted"], so["n_rejected"],
             (z - z_ref).abs().max().item(), (l - l_ref).abs().max().item()))
    assert (st.n_accepted, st.n_rejected) == (so["n_accepted"], so["n_rejected"])
    assert st.nfe == 2 + 6 * (st.n_accepted + st.n_rejected)
    assert (z - z_ref).abs().max() < 2e-4 and (l - l_ref).abs().max() < 2e-4


def test_wide_dopri5_tile_local_equals_oracle_per_tile():
    """tile_rows=16: every 16-row tile is its own control group == the oracle run on each tile separately."""
    m = _model(2, False, seed=4, scale=2.5)
    g = torch.Generator().manual_seed(9)
    rows = 16 * 5 + 7
    x = torch.randn(rows, 2, generator=g) * 1.5
    (_, _, _), (z, l, st) = _solve_pair(m, x, "dopri5", tile_rows=16)
    nfe_max, row_evals = 0, 0
    for lo in range(0, rows, 16):
        (z_ref, l_ref, so), _unused = (None, None, None), None
        blk = m.chain[0]
        T = float(blk.sqrt_end_time.detach() ** 2)
        stats = {}
        blk.odefunc.before_odeint()
        xt = x[lo:lo + 16]
        with torch.no_grad():
            zr, lr = oracle.odeint(blk.odefunc, (xt.clone(), torch.zeros(xt.shape[0], 1)), torch.tensor([0.0, T]),
                                   rtol=[1e-5, 1e-5], atol=[1e-5, 1e-5], method="dopri5", options=dict(stats=stats))
        nfe = 2 + 6 * (stats["n_accepted"] + stats["n_rejected"])
        nfe_max = max(nfe_max, nfe)
        row_evals += nfe * xt.shape[0]
        assert (z[lo:lo + 16] - zr[1]).abs().max() < 2e-4 and (l[lo:lo + 16] - lr[1]).abs().max() < 2e-4
    assert st.nfe == nfe_max and st.row_evals == row_evals


def test_wide_stacked_blocks_module_api_and_limits():
    """BASELINE config 4 shape: 4 CNF blocks of width 256 through the reference's module API (density and sampling
    directions), against the oracle; training entry points say clearly that the adjoint is not built at this width."""
    from gan_flow_amd import gu_ffjord, _native as nat
    mo = _model(2, False, seed=6, scale=1.5, blocks=4)
    mo.eval()
    mp = gu_ffjord.build_model(2, dims=H, num_blocks=4)
    mp.load_state_dict({k: v.float() for k, v in mo.state_dict().items()})
    mp = mp.to(DEV).eval()
    x = torch.randn(500, 2, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        z_ref, dl_ref = mo(x, torch.zeros(500, 1))
        z, dl = mp(x.to(DEV), torch.zeros(500, 1, device=DEV))
        s_ref = mo(z_ref, reverse=True)
        s = mp(z, reverse=True)
    print("wide 4 blocks: z %.1e dlogp %.1e round trip %.1e" % (rel_err(z, z_ref), rel_err(dl, dl_ref), (s.cpu() - x).abs().max().item()))
    assert rel_err(z, z_ref) < 2e-4 and rel_err(dl, dl_ref) < 2e-4
    assert rel_err(s, s_ref) < 2e-4 and (s.cpu() - x).abs().max() < 1e-3
    mp.train()
    with pytest.raises(nat.CnfError, match="not built for hidden width 256"):
        gu_ffjord.compute_loss(mp, x.to(DEV)).backward()
    net = nat.net_desc(2, 256, 3, False)
    with pytest.raises(nat.CnfError, match="at most 4096 rows"):
        nat.solve_fwd(net, _flat(mo).to(DEV), torch.zeros(5000, 2, device=DEV), None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5))
