"""GPU parity of the adjoint (backward) HIP path against the CPU oracle, through the C ABI.

  * one augmented-dynamics evaluation vs torch.autograd on the oracle's ODEfunc (vjp wrt z and theta, including
    the double-backward through the divergence): 2e-5 relative (max-norm) -- a pure fp32 formula check;
  * the whole adjoint solve vs the oracle's odeint_adjoint: same attempt counts, gradients within 2e-4 relative
    (both integrate at rtol = atol = 1e-5; the parameter gradient is a quadrature outside error control);
  * the module surface: loss.backward() through SequentialFlow/CNF fills .grad of every parameter.
"""
import pytest
import torch

import oracle
from _helpers import oracle_fixture_model, oracle_random_model, product_like, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _native():
    from gan_flow_amd import _native
    return _native


def _odefunc(m):
    return m.chain[0].odefunc.odefunc


def _flat(m):
    return torch.cat([p.detach().reshape(-1) for p in _odefunc(m).diffeq.parameters()]).float()


def _oracle_aug(m, t, z, a_z, a_l, e=None):
    """What OdeintAdjointMethod.backward's augmented_dynamics returns for (z, logp, a_z, a_logp)."""
    fn = _odefunc(m)
    fn.before_odeint(e=e)
    params = tuple(fn.diffeq.parameters())
    with torch.enable_grad():
        tt = torch.tensor(t, requires_grad=True)
        zz = z.clone().requires_grad_(True)
        ll = torch.zeros(z.shape[0], 1, requires_grad=True)
        f, nd = fn(tt, (zz, ll))
        grads = torch.autograd.grad((f, nd), (zz,) + params, (-a_z, -a_l), allow_unused=True)
    vz = grads[0]
    vp = torch.cat([torch.zeros_like(p).reshape(-1) if g is None else g.reshape(-1) for g, p in zip(grads[1:], params)])
    return f.detach(), nd.detach(), vz, vp


@pytest.mark.parametrize("case", ["fixture_uni", "fixture_multi", "rand_d1", "rand_d2", "rand_d2_ragged", "rand_d2_hutch",
                                  "rand_d2_big", "rand_d16_hutch", "rand_d16_ragged_hutch"])
def test_eval_aug_matches_autograd(case):
    nat = _native()
    g = torch.Generator().manual_seed(17)
    hutch = case.endswith("hutch")
    if case.startswith("fixture"):
        which = "unimodal" if case.endswith("uni") else "multimodal"
        m = oracle_fixture_model(which)
        dim, rows = 1, 300
        z = torch.rand(rows, 1, generator=g) * (1.0 if which == "unimodal" else 76.0) + (4.5 if which == "unimodal" else 7.0)
    else:
        dim = 16 if "d16" in case else (1 if "d1" in case else 2)
        rows = 77 if "ragged" in case else (4500 if "big" in case else 256)   # > 4096 rows: one-wave-per-tile kernels
        m = oracle_random_model(dim, seed=5, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
        z = torch.randn(rows, dim, generator=g) * 1.5
    m.train()
    a_z = torch.randn(rows, dim, generator=g)
    a_l = torch.randn(rows, 1, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    net = nat.net_desc(dim, 64, 3, hutch)
    params = _flat(m).to(DEV)
    for t in (0.0, 0.31):
        f_ref, nd_ref, vz_ref, vp_ref = _oracle_aug(m, t, z, a_z, a_l, e)
        f, nd, vz, vp = nat.eval_aug(net, params, t, z.to(DEV), a_z.to(DEV), a_l.to(DEV), None if e is None else e.to(DEV))
        torch.cuda.synchronize()
        errs = (rel_err(f, f_ref), rel_err(nd, nd_ref), rel_err(vz, vz_ref), rel_err(vp, vp_ref))
        print("aug %s t=%.2f: rel err f %.1e -div %.1e vjp_z %.1e vjp_theta %.1e" % ((case, t) + errs))
        assert errs[0] < 1e-5 and errs[1] < 2e-5 and errs[2] < 2e-5 and errs[3] < 2e-5
        # per-parameter-block check (a block that is all wrong can hide behind a large neighbour in a max-norm)
        off = 0
        for name, p in _odefunc(m).diffeq.named_parameters():
            n = p.numel()
            blk, ref = vp[off:off + n].cpu(), vp_ref[off:off + n]
            assert (blk - ref).abs().max() <= 5e-5 * max(ref.abs().max().item(), 1e-3 * vp_ref.abs().max().item()), name
            off += n


def _loss(z, dl):
    return -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()


def _oracle_train_grads(m, x):
    m.train()
    m.zero_grad()
    xx = x.clone().requires_grad_(True)
    z, dl = m(xx, torch.zeros(x.shape[0], 1, dtype=x.dtype))
    loss = _loss(z, dl)
    loss.backward()
    blk = m.chain[0]
    gp = torch.cat([p.grad.reshape(-1) for p in _odefunc(m).diffeq.parameters()])
    return loss.item(), z.detach(), dl.detach(), xx.grad.clone(), gp, blk.stats


@pytest.mark.parametrize("case,rows", [("rand_d2", 2048), ("rand_d2", 100), ("rand_d1", 64), ("fixture_uni", 512), ("rand_d2_hutch", 256),
                                       ("rand_d2", 4200), ("fixture_uni", 4100), ("rand_d16_hutch", 2048),
                                       ("rand_d16_hutch", 50)])
def test_solve_bwd_matches_oracle_adjoint(case, rows):
    nat = _native()
    g = torch.Generator().manual_seed(23)
    hutch = case.endswith("hutch")
    if case == "fixture_uni":
        m = oracle_fixture_model("unimodal")
        dim = 1
        x = torch.rand(rows, 1, generator=g) + 4.5
    else:
        dim = 16 if "d16" in case else (1 if "d1" in case else 2)
        m = oracle_random_model(dim, seed=7, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
        x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    m.train()
    fn = _odefunc(m)
    if hutch:   # pin the probe: the oracle's CNF.forward resets it, so hand it over through before_odeint
        orig = fn.before_odeint
        fn.before_odeint = lambda e_=None: orig(e=e)
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(m, x)
    blk = m.chain[0]
    T = float(blk.sqrt_end_time.detach() * blk.sqrt_end_time.detach())
    net = nat.net_desc(dim, 64, 3, hutch)
    params = _flat(m).to(DEV)
    opts = nat.make_opts("dopri5", 1e-5, 1e-5)
    ed = None if e is None else e.to(DEV)
    z1, l1, s_f = nat.solve_fwd(net, params, x.to(DEV), None, 0.0, T, opts, ed)
    # dL/dz1 = z1 / B, dL/dlogp1 = 1 / B for loss = -mean(log N(z1) - logp1)
    gz1 = z1 / rows
    gl1 = torch.full((rows, 1), 1.0 / rows, device=DEV)
    gx, gl0, gp, s_b = nat.solve_bwd(net, params, z1, l1, gz1, gl1, 0.0, T, opts, ed)
    torch.cuda.synchronize()
    s_f, s_b = nat.Stats(s_f), nat.Stats(s_b)
    s_f.raise_for_status()
    s_b.raise_for_status()
    fo, bo = st_ref["forward"], st_ref["backward"]
    print("bwd %s rows=%d: fwd steps hip %d+%d oracle %d+%d | bwd steps hip %d+%d oracle %d+%d | evals run %d"
          % (case, rows, s_f.n_accepted, s_f.n_rejected, fo["n_accepted"], fo["n_rejected"],
             s_b.n_accepted, s_b.n_rejected, bo["n_accepted"], bo["n_rejected"], s_b.evals_run))
    e_gx, e_gp = rel_err(gx, gx_ref), rel_err(gp, gp_ref)
    print("   rel err grad_x %.2e grad_theta %.2e ; grad_logp0 max dev %.1e" % (e_gx, e_gp, (gl0.cpu() - 1.0 / rows).abs().max().item()))
    assert (s_f.n_accepted, s_f.n_rejected) == (fo["n_accepted"], fo["n_rejected"])
    assert (s_b.n_accepted, s_b.n_rejected) == (bo["n_accepted"], bo["n_rejected"])
    assert s_b.nfe == 2 + 6 * (s_b.n_accepted + s_b.n_rejected)
    assert torch.equal(gl0.cpu(), torch.full((rows, 1), 1.0 / rows))
    assert e_gx < 2e-4 and e_gp < 2e-4


_STIFF_ORACLE = {}


def _stiff_case(dim):
    """x + float64 tight-tolerance gradients (tests/golden/stiff_adjoint_d{1,2}.npz, make_stiff_adjoint_vectors.py) and the
    fp32 oracle's adjoint on the same rows (12 s of CPU per case, computed once per process)."""
    import numpy as np
    import os
    from _helpers import GOLDEN, embed_trained_multimodal, fixture_state_dict
    if dim not in _STIFF_ORACLE:
        a = np.load(os.path.join(GOLDEN, "stiff_adjoint_d%d.npz" % dim))
        m = oracle.build_cnf(dim)
        if dim == 1:
            m.load_state_dict(fixture_state_dict("multimodal"))
        else:
            embed_trained_multimodal(m)
        x = torch.from_numpy(a["x"])
        ref32 = _oracle_train_grads(m, x)
        _STIFF_ORACLE[dim] = (m, x, {k: torch.from_numpy(np.asarray(a[k])) for k in a.files}, ref32)
    return _STIFF_ORACLE[dim]


@pytest.mark.parametrize("dim,spread,train", [(1, False, False), (1, True, False), (2, False, False), (2, True, False),
                                              (2, True, True)])
def test_solve_bwd_stiff_trained_field(dim, spread, train):
    """The adjoint solve where BASELINE config 2 spends its time: the reference's trained gu_num=8 field (checkpoint
    weights; d = 2: embedded as bench.py's late_training leg does), NFE ~1.5-2k per solve, every seventh attempt rejected.
    Accept/reject decisions fall within fp32 rounding there, so the kernel's step sequence is not the oracle's.  Bars, as
    for the stiff forward solves: attempt counts within 3 % of the oracle's; gradients no further from the float64
    rtol = atol = 1e-9 adjoint (golden fixture) than 2x the fp32 oracle's own distance, per parameter block.
    Covers commit-on-accept across rejections, the FSAL theta-buffers, the last step's quadrature weights, the
    two-workgroups-per-tile split, and cnf_solve_bwd_train's in-kernel NLL cotangents."""
    nat = _native()
    m, x, gold, (loss_o, z_o, dl_o, gx_o, gp_o, st_o) = _stiff_case(dim)
    rows = x.shape[0]
    blk = m.chain[0]
    T = float(blk.sqrt_end_time.detach() * blk.sqrt_end_time.detach())
    net = nat.net_desc(dim, 64, 3, False)
    params = _flat(m).to(DEV)
    flags = nat.FLAG_SPREAD if spread else 0
    opts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags)
    z1, l1, s_f = nat.solve_fwd(net, params, x.to(DEV), None, 0.0, T, opts)
    if train:
        p_work = params.clone()
        mom, var = torch.zeros_like(params), torch.zeros_like(params)
        state = torch.zeros(2, dtype=torch.int32, device=DEV)
        gx, gl0, gp, loss, s_b = nat.solve_bwd_train(net, p_work, z1, l1, None, None, 0.0, T, opts, mom, var, state,
                                                     want_grad_params=True, lr=0.0)
        assert torch.equal(p_work, params) and int(state[0]) == 1        # lr = 0: a step was taken, nothing moved
        assert abs(loss.item() - float(gold["loss"])) < 2e-3 * abs(float(gold["loss"]))
    else:
        gx, gl0, gp, s_b = nat.solve_bwd(net, params, z1, l1, z1 / rows, torch.full_like(l1, 1.0 / rows), 0.0, T, opts)
    torch.cuda.synchronize()
    s_f, s_b = nat.Stats(s_f), nat.Stats(s_b)
    s_f.raise_for_status()
    s_b.raise_for_status()
    fo, bo = st_o["forward"], st_o["backward"]
    n_f, n_b = s_f.n_accepted + s_f.n_rejected, s_b.n_accepted + s_b.n_rejected
    n_fo, n_bo = fo["n_accepted"] + fo["n_rejected"], bo["n_accepted"] + bo["n_rejected"]
    print("stiff adjoint d=%d spread=%d train=%d: fwd attempts hip %d+%d oracle %d+%d | bwd hip %d+%d oracle %d+%d"
          % (dim, spread, train, s_f.n_accepted, s_f.n_rejected, fo["n_accepted"], fo["n_rejected"],
             s_b.n_accepted, s_b.n_rejected, bo["n_accepted"], bo["n_rejected"]))
    assert abs(n_f - n_fo) <= 0.03 * n_fo and abs(n_b - n_bo) <= 0.03 * n_bo
    assert s_b.nfe == 2 + 6 * n_b and s_b.evals_run == s_b.nfe       # FSAL: nothing re-evaluated
    # Regression pin of the KERNELS' own step sequences on this fixture (recorded on MI355X, round 3).  The controller's
    # arithmetic is restated, not shared, with the oracle (e.g. the fifth root of next_step_size is a seeded Newton iteration,
    # cnf_device.h, where the oracle calls pow): a change there that moves one accept / reject decision among these ~330
    # attempts shows up here even while the 3 % band above still holds.  Update deliberately, with the reason.
    pinned = {1: ((254, 87), (240, 81)), 2: ((254, 76), (247, 78))}[dim]
    assert ((s_f.n_accepted, s_f.n_rejected), (s_b.n_accepted, s_b.n_rejected)) == pinned
    assert torch.equal(gl0.cpu(), torch.full((rows, 1), 1.0 / rows))
    gx64, gp64 = gold["grad_x"], gold["grad_theta"]
    # grad_x: max-norm distance to the float64 solution, hip vs oracle
    e_h, e_o = (gx.cpu().double() - gx64).abs().max().item(), (gx_o.double() - gx64).abs().max().item()
    sx = gx64.abs().max().item()
    print("   grad_x     |hip-f64| %.2e |oracle-f64| %.2e (scale %.2e)" % (e_h, e_o, sx))
    assert e_h <= 2.0 * max(e_o, 2e-5 * sx)
    # grad_theta: per parameter block
    off, worst = 0, 0.0
    sp = gp64.abs().max().item()
    for name, p in _odefunc(m).diffeq.named_parameters():
        n = p.numel()
        ref = gp64[off:off + n]
        b_h = (gp[off:off + n].cpu().double() - ref).abs().max().item()
        b_o = (gp_o[off:off + n].double() - ref).abs().max().item()
        scale = max(ref.abs().max().item(), 1e-3 * sp)
        worst = max(worst, b_h / max(b_o, 2e-5 * scale))
        assert b_h <= 2.0 * max(b_o, 2e-5 * scale), (name, b_h, b_o, scale)
        off += n
    e_h, e_o = (gp.cpu().double() - gp64).abs().max().item(), (gp_o.double() - gp64).abs().max().item()
    print("   grad_theta |hip-f64| %.2e |oracle-f64| %.2e (scale %.2e); worst block ratio hip/oracle %.2f" % (e_h, e_o, sp, worst))


@pytest.mark.parametrize("case,rows,spread,zero_al", [
    ("rand_d2", 2048, False, False), ("rand_d2", 2048, True, False), ("rand_d1", 64, False, False),
    ("rand_d2", 4200, False, False), ("rand_d16_hutch", 512, False, False), ("rand_d16_hutch", 2048, True, False),
    ("rand_h256", 200, False, False), ("rand_d2", 300, False, True), ("stiff_d2", 0, True, False)])
def test_adjoint_control_all_matches_oracle(case, rows, spread, zero_al):
    """CNF_FLAG_ADJ_CONTROL_ALL (SURVEY.md Appendix D.3's switch): a_z and a_logp under the adjoint solve's step control.
    Against the oracle's odeint_adjoint(adjoint_control='rows') -- tolerance lists as long as (z, logp, a_z, a_logp) --
    in every adjoint kernel family: identical step counts and the usual gradient bar on the random-init fields; on the
    stiff trained field -- with the cotangents of the SUMMED loss (rows x those of the mean), so that a_z is large against
    atol and its error ratio decides: ~75 % more attempts than state_only -- the 3 % band around the oracle's count of THAT
    mode, and gradients no further from float64 than 2x the oracle's.  zero_al: cotangent of logp identically zero (a_logp's d0 / d1 is then 0 / 0 instead of inf: the heuristic's
    ordinary branch with the a_z terms).  Parity unpinned in both modes (no reference artefact holds backward NFE)."""
    nat = _native()
    g = torch.Generator().manual_seed(29)
    hutch = case.endswith("hutch")
    hidden = 256 if "h256" in case else 64
    if case == "stiff_d2":
        m, x, gold, _ = _stiff_case(2)
        dim, rows = 2, x.shape[0]
    else:
        dim = 16 if "d16" in case else (1 if "d1" in case else 2)
        kw = dict(hidden_dims=(256, 256, 256)) if hidden == 256 else {}
        m = oracle_random_model(dim, seed=7, divergence_fn="approximate" if hutch else "brute_force", scale=2.0 if hidden == 64 else 1.5, **kw)
        x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    m.train()
    fn = _odefunc(m)
    if hutch:
        orig = fn.before_odeint
        fn.before_odeint = lambda e_=None: orig(e=e)
    blk = m.chain[0]
    cot = float(rows) if case == "stiff_d2" else 1.0   # cotangent scale: 1 = mean loss, rows = summed loss

    def oracle_run(mode):
        blk.adjoint_error_control = mode
        m.zero_grad()
        xx = x.clone().requires_grad_(True)
        z, dl = m(xx, torch.zeros(rows, 1))
        loss = -oracle.standard_normal_logprob(z).sum(1, keepdim=True).mean() if zero_al else _loss(z, dl)
        (loss * cot).backward()
        gp = torch.cat([p.grad.reshape(-1) for p in fn.diffeq.parameters()])
        return xx.grad.clone(), gp, dict(blk.stats["backward"])

    try:
        gx_ref, gp_ref, bo = oracle_run("rows")
        _, _, bo_state = oracle_run("state_only") if case == "stiff_d2" else (None, None, None)
    finally:
        blk.adjoint_error_control = "state_only"
    T = float(blk.sqrt_end_time.detach() * blk.sqrt_end_time.detach())
    net = nat.net_desc(dim, hidden, 3, hutch)
    params = _flat(m).to(DEV)
    opts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_ADJ_CONTROL_ALL | (nat.FLAG_SPREAD if spread else 0))
    ed = None if e is None else e.to(DEV)
    z1, l1, s_f = nat.solve_fwd(net, params, x.to(DEV), None, 0.0, T, opts, ed)
    gl1 = torch.zeros(rows, 1, device=DEV) if zero_al else torch.full((rows, 1), cot / rows, device=DEV)
    gx, gl0, gp, s_b = nat.solve_bwd(net, params, z1, l1, z1 * (cot / rows), gl1, 0.0, T, opts, ed)
    torch.cuda.synchronize()
    s_b = nat.Stats(s_b)
    s_b.raise_for_status()
    e_gx, e_gp = rel_err(gx, gx_ref), rel_err(gp, gp_ref)
    print("adjoint control ALL %s rows=%d spread=%d zero_al=%d: bwd steps hip %d+%d oracle('rows') %d+%d first dt %.5g | rel err grad_x "
          "%.2e grad_theta %.2e" % (case, rows, spread, zero_al, s_b.n_accepted, s_b.n_rejected, bo["n_accepted"], bo["n_rejected"],
                                    bo["dts"][0], e_gx, e_gp))
    assert s_b.nfe == 2 + 6 * (s_b.n_accepted + s_b.n_rejected)
    if case == "stiff_d2":
        n_b, n_bo = s_b.n_accepted + s_b.n_rejected, bo["n_accepted"] + bo["n_rejected"]
        n_state = bo_state["n_accepted"] + bo_state["n_rejected"]
        print("   oracle('state_only') attempts %d" % n_state)
        assert abs(n_b - n_bo) <= 0.03 * n_bo
        assert n_bo > 1.3 * n_state        # the switch does something here
        gx64, gp64 = gold["grad_x"] * cot, gold["grad_theta"] * cot
        assert (gx.cpu().double() - gx64).abs().max() <= 2.0 * max((gx_ref.double() - gx64).abs().max().item(), 2e-5 * gx64.abs().max().item())
        assert (gp.cpu().double() - gp64).abs().max() <= 2.0 * max((gp_ref.double() - gp64).abs().max().item(), 2e-5 * gp64.abs().max().item())
    else:
        assert (s_b.n_accepted, s_b.n_rejected) == (bo["n_accepted"], bo["n_rejected"])
        assert e_gx < 2e-4 and e_gp < 2e-4


def test_module_training_step_fills_parameter_grads():
    from gan_flow_amd import gu_ffjord
    mo = oracle_random_model(2, seed=11, scale=1.5)
    mo.train()
    mp = product_like(mo, DEV)
    mp.train()
    x = torch.randn(512, 2, generator=torch.Generator().manual_seed(1))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    nfe_total_ref = mo.chain[0].num_evals()
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    nfe_fwd = gu_ffjord.count_nfe(mp)
    loss.backward()
    nfe_total = gu_ffjord.count_nfe(mp)
    gp = torch.cat([p.grad.reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()])
    print("module step: loss hip %.6f oracle %.6f | NFE fwd %d total %d (oracle total %d)" % (loss.item(), loss_ref, nfe_fwd, nfe_total, nfe_total_ref))
    assert abs(loss.item() - loss_ref) < 1e-4
    assert nfe_total == nfe_total_ref
    assert rel_err(gp, gp_ref) < 2e-4 and rel_err(xp.grad, gx_ref) < 2e-4
    assert mp.chain[0].sqrt_end_time.grad is None       # the end time is detached, as in the reference
    # the same step with two workgroups per tile in the adjoint (CNF.spread_adjoint): identical gradients
    mp.zero_grad(set_to_none=True)
    mp.chain[0].spread_adjoint = True
    xs = x.to(DEV).requires_grad_(True)
    gu_ffjord.compute_loss(mp, xs).backward()
    gs = torch.cat([p.grad.reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()])
    assert torch.equal(gs, gp) and torch.equal(xs.grad, xp.grad)


@pytest.mark.parametrize("layer_type", ["ignore", "concat", "concat_v2", "squash"])
def test_other_layer_types_run_on_the_same_kernels(layer_type):
    """the diffeq layers that are special cases of concatsquash reach the kernels through
    kernel_slices() (constants for the absent gate / time term); loss, step counts and the gradients of the layer's OWN
    parameters match the oracle's autograd through its adjoint.  (Layer definitions are UPSTREAM-RECALL: parity unpinned.)"""
    from gan_flow_amd import gu_ffjord
    mo = oracle_random_model(2, seed=13, scale=1.5, layer_type=layer_type)
    mo.train()
    mp = product_like(mo, DEV, layer_type=layer_type)
    mp.train()
    assert [k for k, _ in mp.named_parameters()] == [k for k, _ in mo.named_parameters()]
    x = torch.randn(512, 2, generator=torch.Generator().manual_seed(2))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()])
    assert abs(loss.item() - loss_ref) < 1e-4
    # `ignore` has no time dependence: the field is so smooth that the error estimate of a step is fp32 rounding (ratio
    # ~1e-5) and the growth of the step size follows that noise.  The forward kernel's bf16-split hidden layers round
    # less than the oracle's fp32 matmul (tools/bf16_split_layer.hip), so it may need one attempt fewer; with another
    # step sequence the gradients agree to the integrator's accuracy, not to rounding (as in test_other_nonlinearities).
    nfe, nfe_ref = gu_ffjord.count_nfe(mp), mo.chain[0].num_evals()
    assert abs(nfe - nfe_ref) <= 6
    bar = 2e-4 if nfe == nfe_ref else 2e-3
    assert gp.numel() == gp_ref.numel()
    assert rel_err(gp, gp_ref) < bar and rel_err(xp.grad, gx_ref) < bar
    if layer_type == "concat":   # its time column sits inside the weight matrix: no place of its own in the kernels' layout
        with pytest.raises(NotImplementedError):
            gu_ffjord.FusedTrainStep(mp, xp.detach(), graph=False)
    else:                        # aliased, constants listed (tests/test_gpu_train_step.py trains them)
        assert gu_ffjord.FusedTrainStep(mp, xp.detach(), graph=False).consts[0] is not None


def test_short_training_run_reduces_loss():
    """Ten Adam steps of the real driver loop on gu_num=8 data lower the negative log-likelihood."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    torch.manual_seed(1)
    data = make_gu(8, seed=1)
    model = gu_ffjord.build_model(2).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    losses = []
    for _ in range(10):
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(model, data.get_sample(2048).to(DEV))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("losses", ["%.3f" % v for v in losses])
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def test_graphed_train_step_matches_eager():
    """The hipGraph-captured iteration takes the same training trajectory as the eager loop (same kernels, same order)."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(8, seed=3)
    batches = [data.get_sample(2048).to(DEV) for _ in range(8)]

    def run(graphed):
        torch.manual_seed(5)
        model = gu_ffjord.build_model(2).to(DEV).train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True)
        losses = []
        if graphed:
            step = gu_ffjord.GraphedTrainStep(model, opt, batches[0], batches[:2])
            for x in batches[2:]:
                losses.append(step(x).item())
            step.check()
            assert step.nfe.tolist() == [14 * 6, 14 * 6] or min(step.nfe.tolist()) >= 8 * 6
        else:
            for i, x in enumerate(batches):
                opt.zero_grad(set_to_none=True)
                loss = gu_ffjord.compute_loss(model, x)
                loss.backward()
                opt.step()
                if i >= 2:
                    losses.append(loss.item())
        return losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()])

    le, pe = run(False)
    lg, pg = run(True)
    print("eager ", ["%.4f" % v for v in le])
    print("graph ", ["%.4f" % v for v in lg])
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(le, lg))
    assert rel_err(pg, pe) < 1e-6


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 2048), (2, False, 4500), (16, True, 512), (1, False, 300)])
def test_solves_do_not_depend_on_stale_lds(dim, hutch, rows):
    """LDS keeps the previous kernel's content.  Poison it with NaNs / huge values before the forward and adjoint
    solves (both kernel families, batch-global and tile-local control): results must be bit-identical to a clean run."""
    nat = _native()
    g = torch.Generator().manual_seed(31)
    m = oracle_random_model(dim, seed=9, divergence_fn="approximate" if hutch else "brute_force", scale=1.5)
    net = nat.net_desc(dim, 64, 3, hutch)
    params = _flat(m).to(DEV)
    x = torch.randn(rows, dim, generator=g).to(DEV)
    e = torch.randn(rows, dim, generator=g).to(DEV) if hutch else None
    outs = []
    for pattern in (0x00000000, 0x7FC00000, 0x7F7FFFFF, 0xFF800000):
        res = []
        for tile_rows in (0, 16):
            opts = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=tile_rows)
            nat.debug_fill_lds(pattern)
            z1, l1, s_f = nat.solve_fwd(net, params, x, None, 0.0, 0.5, opts, e)
            res += [z1, l1, s_f[:4]]
            if tile_rows == 0 and (dim <= 2 or rows <= 4096):
                nat.debug_fill_lds(pattern)
                gx, gl, gp, s_b = nat.solve_bwd(net, params, z1, l1, z1 / rows, torch.full_like(l1, 1.0 / rows), 0.0, 0.5, opts, e)
                res += [gx, gp, s_b[:4]]
        torch.cuda.synchronize()
        assert all(int(r[0]) == 0 for r in res if r.dtype == torch.int32), "solver status"
        outs.append([r.cpu() for r in res])
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dim,hidden,rows", [(2, 64, 2048), (2, 64, 100), (16, 64, 512), (2, 256, 300)])
def test_flag_clean_calls_match_zeroing_calls(dim, hidden, rows):
    """CNF_FLAG_CLEAN (include/cnf_hip.h): on a zero-initialised workspace / stats block the solves enqueue no memset
    and leave the workspace head clean again.  Three flagged forward + adjoint rounds on ONE workspace give exactly the
    results of ordinary (zeroing) calls, the head of the workspace is zero after every call, and counters of a later,
    shorter solve replace those of an earlier one (no stale maxima)."""
    nat = _native()
    hutch = dim == 16
    g = torch.Generator().manual_seed(41)
    m = oracle_random_model(dim, seed=9, divergence_fn="approximate" if hutch else "brute_force", scale=1.5,
                            hidden_dims=(hidden,) * 3)
    net = nat.net_desc(dim, hidden, 3, hutch)
    params = _flat(m).to(DEV)
    x = torch.randn(rows, dim, generator=g).to(DEV)
    e = torch.randn(rows, dim, generator=g).to(DEV) if hutch else None
    ws = torch.zeros(nat.workspace_bytes(net, rows), dtype=torch.uint8, device=DEV)
    st = torch.zeros(2, 8, dtype=torch.int32, device=DEV)
    head = 16384 + 64
    for T in (0.5, 0.5, 0.05):      # the last round is a much shorter solve: fewer steps than the rounds before
        ref_o = nat.make_opts("dopri5", 1e-5, 1e-5)
        z_r, l_r, sf_r = nat.solve_fwd(net, params, x, None, 0.0, T, ref_o, e)
        gx_r, gl_r, gp_r, sb_r = nat.solve_bwd(net, params, z_r, l_r, z_r / rows, torch.full_like(l_r, 1.0 / rows), 0.0, T, ref_o, e)
        o = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_CLEAN)
        z, l, sf = nat.solve_fwd(net, params, x, None, 0.0, T, o, e, stats=st[0], workspace=ws)
        torch.cuda.synchronize()
        assert int(ws[:head].max()) == 0
        gx, gl, gp, sb = nat.solve_bwd(net, params, z, l, z / rows, torch.full_like(l, 1.0 / rows), 0.0, T, o, e, stats=st[1], workspace=ws)
        torch.cuda.synchronize()
        assert int(ws[:head].max()) == 0
        for a, b in ((z, z_r), (l, l_r), (gx, gx_r), (gp, gp_r), (sf, sf_r), (sb, sb_r)):
            assert torch.equal(a.cpu(), b.cpu())
        assert nat.Stats(sf).status == 0 and nat.Stats(sb).status == 0
    assert nat.Stats(sb).n_accepted < 3


@pytest.mark.parametrize("dim,hutch,rows", [(2, False, 2048), (2, False, 100), (1, False, 1000), (2, True, 256), (2, False, 1),
                                            (2, False, 2064), (16, True, 2048), (16, True, 50)])
def test_spread_adjoint_equals_plain_adjoint(dim, hutch, rows):
    """CNF_FLAG_SPREAD (include/cnf_hip.h): two workgroups per 16-row tile, each keeping half of the d/dtheta sums.  The
    arithmetic per tile and the order of the additions over tiles are those of the plain launch, so everything the call
    returns is identical bit for bit (2064 rows = 129 tiles do not fit twice: the flag is ignored there; d = 16: see below)."""
    nat = _native()
    g = torch.Generator().manual_seed(31)
    m = oracle_random_model(dim, seed=9, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
    net = nat.net_desc(dim, 64, 3, hutch)
    params = _flat(m).to(DEV)
    x = torch.randn(rows, dim, generator=g).to(DEV)
    e = torch.randn(rows, dim, generator=g).to(DEV) if hutch else None
    plain = nat.make_opts("dopri5", 1e-5, 1e-5)
    z1, l1, _ = nat.solve_fwd(net, params, x, None, 0.0, 0.5, plain, e)
    gz1 = z1 / rows
    gl1 = torch.full((rows, 1), 1.0 / rows, device=DEV)
    outs = []
    for flags in (0, nat.FLAG_SPREAD, nat.FLAG_SPREAD):
        o = nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags)
        gx, gl0, gp, st = nat.solve_bwd(net, params, z1, l1, gz1, gl1, 0.0, 0.5, o, e)
        torch.cuda.synchronize()
        st = nat.Stats(st)
        st.raise_for_status()
        outs.append((gx.clone(), gl0.clone(), gp.clone(), (st.nfe, st.n_accepted, st.n_rejected, st.evals_run)))
    assert outs[2][3] == outs[1][3] and all(torch.equal(u, v) for u, v in zip(outs[1][:3], outs[2][:3]))   # repeatable
    if dim == 16:
        # d = 16: the spread kernel also reuses the seventh stage's d/dtheta as the next step's first (FSAL: six evaluations
        # per attempt, its half-size accumulators leave room for the buffers); the plain kernel re-evaluates it at a time
        # that may differ in the last bit -> same steps, gradients equal to rounding
        assert outs[1][3][:3] == outs[0][3][:3]
        assert outs[1][3][3] == outs[1][3][0] and outs[0][3][3] == outs[0][3][0] + outs[0][3][1] + outs[0][3][2]
        assert rel_err(outs[1][0], outs[0][0]) < 1e-5 and torch.equal(outs[1][1], outs[0][1])
        assert rel_err(outs[1][2], outs[0][2]) < 1e-5
    else:
        assert outs[1][3] == outs[0][3]
        assert all(torch.equal(u, v) for u, v in zip(outs[1][:3], outs[0][:3]))
    assert outs[0][2].abs().max().item() > 0


@pytest.mark.parametrize("dim,hutch,stiff", [(2, False, False), (1, False, False), (2, True, False), (2, False, True)])
def test_spread_slabs_equal_plain_slabs_per_tile(dim, hutch, stiff):
    """The finest form of the comparison above: every entry of every workgroup's gradient slab.  A two-workgroups-per-tile
    launch leaves, per tile, the lower half of the parameters in the odd workgroup's slab and the upper half in the even
    one's; each owned entry must carry the bits of the one-workgroup-per-tile launch's slab of that tile.  (A sum over 128
    tiles can hide a wrong contribution of one tile behind rounding; this cannot.  It is the check that found the packed-fp32
    fault described in DESIGN.md section 4 -- a gfx950 erratum of packed fp32 instructions beside an MFMA stream, removed
    from the eight-wave kernel's translation unit by csrc/pk_hazard_pass.py.)  stiff: the rows of the stiff golden fixture
    on the reference's trained gu_num=8 field -- hundreds of commits with rejections in between, the regime config 2 lives in."""
    nat = _native()
    rows, H, gran = 2048, 64, 16448      # gran: bytes of reduction granules in front of the slabs (kGranBytes, cnf_api.hip)
    g = torch.Generator().manual_seed(31)
    if stiff:
        import numpy as np
        import os
        from _helpers import GOLDEN, embed_trained_multimodal
        m = oracle.build_cnf(dim)
        embed_trained_multimodal(m)
        x = torch.from_numpy(np.load(os.path.join(GOLDEN, "stiff_adjoint_d%d.npz" % dim))["x"]).to(DEV)
        rows = x.shape[0]
        assert rows % 16 == 0
    else:
        m = oracle_random_model(dim, seed=9, divergence_fn="approximate" if hutch else "brute_force", scale=2.0)
        x = torch.randn(rows, dim, generator=g).to(DEV)
    net = nat.net_desc(dim, H, 3, hutch)
    params = _flat(m).to(DEV)
    P = params.numel()
    e = torch.randn(rows, dim, generator=g).to(DEV) if hutch else None
    z1, l1, _ = nat.solve_fwd(net, params, x, None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5), e)
    gz1, gl1 = z1 / rows, torch.full((rows, 1), 1.0 / rows, device=DEV)
    tiles = rows // 16
    w3 = dim * H + 4 * H + H * H + 4 * H          # first parameter of the upper half (W3)
    slabs = {}
    for flags, name, n in ((0, "plain", tiles), (nat.FLAG_SPREAD, "spread", 2 * tiles)):
        ws = torch.full((gran + 4 * P * 2 * tiles + 4096,), 0x7F, dtype=torch.uint8, device=DEV)
        st = nat.solve_bwd(net, params, z1, l1, gz1, gl1, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags), e, workspace=ws)[3]
        torch.cuda.synchronize()
        st = nat.Stats(st)
        st.raise_for_status()
        if stiff:
            assert st.n_accepted > 100 and st.n_rejected > 10      # the regime the case is there for
        slabs[name] = ws[gran:gran + 4 * P * n].view(torch.float32).view(n, P).cpu()
    plain, spread = slabs["plain"], slabs["spread"]
    lower, upper = spread[1::2], spread[0::2]
    assert plain.abs().max().item() > 0
    assert torch.equal(lower[:, :w3].view(torch.int32), plain[:, :w3].view(torch.int32))
    assert torch.equal(upper[:, w3:].view(torch.int32), plain[:, w3:].view(torch.int32))


@pytest.mark.parametrize("hidden", [(32, 32, 32), (48, 64, 16), (100, 100, 100), (7, 130, 64)])
def test_other_hidden_widths_run_zero_padded(hidden):
    """Hidden widths other than 64 / 256 (ODEnet.kernel_hidden): the kernels see the net zero-padded to the next built
    width; loss, solver steps and the gradients of the real parameters match the oracle at those widths."""
    from gan_flow_amd import gu_ffjord
    mo = oracle_random_model(2, seed=17, scale=1.5, hidden_dims=hidden)
    mo.train()
    mp = gu_ffjord.build_model(2, dims=hidden)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    net = mp.chain[0].odefunc.odefunc.diffeq
    assert net.is_padded() and net.kernel_hidden() == (64 if max(hidden) <= 64 else 256)
    x = torch.randn(300, 2, generator=torch.Generator().manual_seed(2))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert gp.numel() == gp_ref.numel()
    st = mp.chain[0].last_stats
    assert (st["forward"].n_accepted, st["forward"].n_rejected) == (st_ref["forward"]["n_accepted"], st_ref["forward"]["n_rejected"])
    assert (st["backward"].n_accepted, st["backward"].n_rejected) == (st_ref["backward"]["n_accepted"], st_ref["backward"]["n_rejected"])
    assert abs(loss.item() - loss_ref) < 1e-4
    assert rel_err(gp, gp_ref) < 2e-4 and rel_err(xp.grad, gx_ref) < 2e-4
    assert gu_ffjord.FusedTrainStep(mp, xp.detach(), graph=False).consts[0] is not None   # the padding, restored after every update


@pytest.mark.parametrize("dim,hidden,hutch,rows", [(2, 64, False, 512), (1, 64, False, 5000), (16, 64, True, 300), (2, 256, False, 200),
                                                   (2, 64, True, 2048)])
def test_residual_dynamics(dim, hidden, hutch, rows):
    """ODEfunc(residual=True) (UPSTREAM-RECALL, parity unpinned): dy = f(t, y) - y, divergence - d.
    Every kernel family applies it where it consumes an evaluation (CNF_DIV_RESIDUAL); training loss, solver steps and
    gradients against the oracle, plus a fixed-grid evaluation pass."""
    from gan_flow_amd import gu_ffjord
    div = "approximate" if hutch else "brute_force"
    mo = oracle_random_model(dim, seed=19, scale=1.5, divergence_fn=div, hidden_dims=(hidden,) * 3, residual=True)
    mo.train()
    mp = gu_ffjord.build_model(dim, dims=(hidden,) * 3, divergence_fn=div, residual=True)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    if hutch:
        fo, fp = _odefunc(mo), mp.chain[0].odefunc.odefunc
        orig = fo.before_odeint
        fo.before_odeint = lambda e_=None: orig(e=e)
        origp = fp.before_odeint
        fp.before_odeint = lambda e_=None: origp(e=e.to(DEV))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()])
    st = mp.chain[0].last_stats
    assert (st["forward"].n_accepted, st["forward"].n_rejected) == (st_ref["forward"]["n_accepted"], st_ref["forward"]["n_rejected"])
    assert (st["backward"].n_accepted, st["backward"].n_rejected) == (st_ref["backward"]["n_accepted"], st_ref["backward"]["n_rejected"])
    assert abs(loss.item() - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))
    assert rel_err(gp, gp_ref) < 2e-4 and rel_err(xp.grad, gx_ref) < 2e-4
    # the residual term matters: the same weights without it give another loss
    mq = gu_ffjord.build_model(dim, dims=(hidden,) * 3, divergence_fn=div, residual=False)
    mq.load_state_dict(mo.state_dict())
    mq = mq.to(DEV).train()
    if hutch:
        fq = mq.chain[0].odefunc.odefunc
        origq = fq.before_odeint
        fq.before_odeint = lambda e_=None: origq(e=e.to(DEV))
    assert abs(gu_ffjord.compute_loss(mq, x.to(DEV)).item() - loss_ref) > 1e-2
    if not hutch:
        # fixed-grid rk4 in eval mode (the one-tile-per-wave / wide rk4 kernels)
        mo.eval()
        mp.eval()
        for blk in (mo.chain[0], mp.chain[0]):
            blk.test_solver, blk.solver_options = "rk4", {}
        with torch.no_grad():
            z_o, l_o = mo(x[:100], torch.zeros(100, 1))
        z_p, l_p = mp(x[:100].to(DEV), torch.zeros(100, 1, device=DEV))
        assert rel_err(z_p, z_o) < 1e-5 and rel_err(l_p, l_o) < 1e-4


def test_flow_with_batch_norm_layers_trains_and_inverts():
    """build_model(batch_norm=True): MovingBatchNorm1d layers around the CNF blocks (tensor ops) compose with the HIP
    solves through autograd; sampling inverts the density direction."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    torch.manual_seed(8)
    model = gu_ffjord.build_model(2, num_blocks=2, batch_norm=True, bn_lag=0.1).to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    data = make_gu(8, seed=5)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(model, data.get_sample(512).to(DEV))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad and p.dim() > 0)
    assert min(losses[-3:]) < losses[0] and gu_ffjord.count_nfe(model) > 0
    model.eval()
    x = data.get_sample(200).to(DEV)
    z, dl = model(x, torch.zeros(200, 1, device=DEV))
    xb, dlb = model(z, dl, reverse=True)
    assert rel_err(xb, x) < 1e-3 and dlb.abs().max().item() < 5e-3


@pytest.mark.parametrize("act", ["softplus", "elu", "relu"])
@pytest.mark.parametrize("dim,hidden,hutch,rows", [(2, 64, False, 512), (1, 64, False, 5000), (16, 64, True, 300), (2, 256, False, 200),
                                                   (2, (40, 64, 24), False, 100)])
def test_other_nonlinearities(act, dim, hidden, hutch, rows):
    """ODEnet(nonlinearity='softplus' | 'elu'): one kernel library per nonlinearity
    (libcnf_hip_<act>.so, csrc/cnf_device.h CNF_ACT), chosen by the net description.  One evaluation, the training loss,
    the solver steps and the adjoint gradients against the oracle, in every kernel family."""
    from gan_flow_amd import gu_ffjord
    nat = _native()
    div = "approximate" if hutch else "brute_force"
    dims = (hidden,) * 3 if isinstance(hidden, int) else hidden
    mo = oracle_random_model(dim, seed=23, scale=1.5, divergence_fn=div, hidden_dims=dims, nonlinearity=act)
    mo.train()
    mp = gu_ffjord.build_model(dim, dims=dims, divergence_fn=div, nonlinearity=act)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    fo, fp = _odefunc(mo), mp.chain[0].odefunc.odefunc
    assert fp.net_desc().act == act and nat.load_library(act).cnf_activation() == nat.ACTIVATIONS[act]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(rows, dim, generator=g) * 1.5
    e = torch.randn(rows, dim, generator=g) if hutch else None
    # one evaluation of (f, -div)
    fo.before_odeint(e=e)
    with torch.no_grad():
        pass
    f_o, nd_o = fo(torch.tensor(0.3), (x.clone(), torch.zeros(rows, 1)))[:2]
    f_p, nd_p = nat.eval_field(fp.net_desc(), fp.diffeq.flat_parameters().detach(), 0.3, x.to(DEV), None if e is None else e.to(DEV))
    assert rel_err(f_p, f_o) < 1e-5 and rel_err(nd_p, nd_o) < 2e-5
    if hutch:
        orig = fo.before_odeint
        fo.before_odeint = lambda e_=None: orig(e=e)
        origp = fp.before_odeint
        fp.before_odeint = lambda e_=None: origp(e=e.to(DEV))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in fp.diffeq.parameters()])
    st = mp.chain[0].last_stats
    assert (st["forward"].n_accepted, st["forward"].n_rejected) == (st_ref["forward"]["n_accepted"], st_ref["forward"]["n_rejected"])
    # the smooth softplus field makes the first adjoint step's error ratio ~6e-6, i.e. an error estimate of fp32 rounding
    # size: the growth factor lands on either side of its cap of 10 (2 steps here, 3 in the oracle at d=2); same solution
    nb, nb_ref = st["backward"].n_accepted + st["backward"].n_rejected, st_ref["backward"]["n_accepted"] + st_ref["backward"]["n_rejected"]
    assert abs(nb - nb_ref) <= 1
    assert abs(loss.item() - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))
    # same step sequence: rounding-level agreement (2e-4); one step more or fewer: agreement at the level the 1e-5
    # tolerances leave (2e-3).  A case outside its bar passes only if the float64 adjoint (same weights, tolerances
    # 1e-10) shows that the fp32 ORACLE is that far from the truth itself and the kernel no further than twice that:
    # elu at d = 16 (Hutchinson), whose act'' jumps at 0 -- rows that cross the kink inside a step pick up O(dt) differences
    # from where fp32 rounding puts the crossing (measured: kernel 3.2e-4 from the oracle; every other case < 1.1e-4).
    e_gp, e_gx = rel_err(gp, gp_ref), rel_err(xp.grad, gx_ref)
    print("%s d=%d: adjoint attempts hip %d oracle %d, grad_theta %.2e grad_x %.2e" % (act, dim, nb, nb_ref, e_gp, e_gx))
    bar = 2e-4 if nb == nb_ref else 2e-3
    if not (e_gp < bar and e_gx < bar):
        m64 = oracle_random_model(dim, seed=23, scale=1.5, divergence_fn=div, hidden_dims=dims, nonlinearity=act).double()
        if hutch:
            f64 = _odefunc(m64)
            orig64 = f64.before_odeint
            f64.before_odeint = lambda e_=None: orig64(e=e.double())
        m64.chain[0].rtol = m64.chain[0].atol = 1e-10
        _, _, _, gx64, gp64, _ = _oracle_train_grads(m64, x.double())
        d_k, d_o = rel_err(gp, gp64.float()), rel_err(gp_ref, gp64.float())
        dx_k, dx_o = rel_err(xp.grad, gx64.float()), rel_err(gx_ref, gx64.float())
        print("   against the float64 adjoint at 1e-10: grad_theta kernel %.2e oracle %.2e | grad_x kernel %.2e oracle %.2e" % (d_k, d_o, dx_k, dx_o))
        assert max(d_o, dx_o) > 0.5 * bar and d_k <= 2 * d_o and dx_k <= 2 * dx_o


@pytest.mark.parametrize("solver", ["rk4", "euler", "midpoint"])
@pytest.mark.parametrize("dim,hutch,rows,step", [(2, False, 300, 0.1), (1, False, 70, None), (2, True, 128, 0.07), (2, False, 5000, 0.2)])
def test_fixed_grid_adjoint_matches_oracle(solver, dim, hutch, rows, step):
    """Training with a fixed-grid solver (`--solver rk4 | euler | midpoint [--step_size h]`):
    odeint_adjoint integrates the augmented state backwards with the same method on the same kind of grid
    (fixed_bwd_kernel, csrc/cnf_bwd.hip).  Loss, grad_x and grad_theta against the oracle's adjoint on identical weights and
    inputs; both run the same arithmetic in fp32, so the bar is rounding level.  Parity unpinned: no reference artefact holds
    gradients, the oracle restates torchdiffeq's fixed-grid solvers and adjoint as recalled."""
    from gan_flow_amd import gu_ffjord
    div = "approximate" if hutch else "brute_force"
    mo = oracle_random_model(dim, seed=11, scale=1.5, divergence_fn=div, solver=solver)
    mo.train()
    mp = product_like(mo, DEV, divergence_fn=div, solver=solver)
    opts = {"step_size": step} if step else {}
    for blk in list(mo.chain) + list(mp.chain):
        blk.solver_options = dict(opts)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, dim, generator=g) * 1.2
    fo, fp = _odefunc(mo), mp.chain[0].odefunc.odefunc
    if hutch:
        e = torch.randn(rows, dim, generator=g)
        orig, origp = fo.before_odeint, fp.before_odeint
        fo.before_odeint = lambda e_=None: orig(e=e)
        fp.before_odeint = lambda e_=None: origp(e=e.to(DEV))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in fp.diffeq.parameters()])
    st = mp.chain[0].last_stats
    n_steps = st_ref["backward"]["n_steps"]
    per_step = {"rk4": 4, "euler": 1, "midpoint": 2}[solver]
    assert st["backward"].status == 0 and st["backward"].n_accepted == n_steps and st["backward"].nfe == per_step * n_steps
    assert st["forward"].n_accepted == st_ref["forward"]["n_steps"]
    e_gp, e_gx = rel_err(gp, gp_ref), rel_err(xp.grad, gx_ref)
    print("%s d=%d rows=%d: %d steps, loss %.6f / %.6f, grad_theta %.2e grad_x %.2e" % (solver, dim, rows, n_steps, loss.item(), loss_ref, e_gp, e_gx))
    assert abs(loss.item() - loss_ref) < 2e-5 * max(1.0, abs(loss_ref))
    assert e_gp < 5e-5 and e_gx < 5e-5


@pytest.mark.parametrize("solver", ["rk4", "euler", "midpoint"])
@pytest.mark.parametrize("dim,hidden,hutch,rows,step", [(16, 64, True, 200, 0.1), (16, 64, True, 2048, None), (2, 256, False, 100, 0.125),
                                                        (1, 256, False, 40, None), (2, 256, True, 64, 0.3)])
def test_fixed_grid_adjoint_other_kernel_families(solver, dim, hidden, hutch, rows, step):
    """The fixed-grid adjoint at d = 16 (fixed_bwd_split16_kernel, csrc/cnf_split16.hip) and at hidden width 256
    (wide_fixed_bwd_kernel, csrc/cnf_wide_bwd.hip) against the oracle's adjoint, as test_fixed_grid_adjoint_matches_oracle."""
    from gan_flow_amd import gu_ffjord
    div = "approximate" if hutch else "brute_force"
    dims = (hidden,) * 3
    mo = oracle_random_model(dim, seed=13, scale=1.5 if hidden == 64 else 1.0, divergence_fn=div, solver=solver, hidden_dims=dims)
    mo.train()
    mp = gu_ffjord.build_model(dim, dims=dims, divergence_fn=div, solver=solver)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    opts = {"step_size": step} if step else {}
    for blk in list(mo.chain) + list(mp.chain):
        blk.solver_options = dict(opts)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(rows, dim, generator=g) * 1.2
    fo, fp = _odefunc(mo), mp.chain[0].odefunc.odefunc
    if hutch:
        e = torch.randn(rows, dim, generator=g)
        orig, origp = fo.before_odeint, fp.before_odeint
        fo.before_odeint = lambda e_=None: orig(e=e)
        fp.before_odeint = lambda e_=None: origp(e=e.to(DEV))
    loss_ref, z_ref, dl_ref, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in fp.diffeq.parameters()])
    st = mp.chain[0].last_stats
    n_steps = st_ref["backward"]["n_steps"]
    per_step = {"rk4": 4, "euler": 1, "midpoint": 2}[solver]
    assert st["backward"].status == 0 and st["backward"].n_accepted == n_steps and st["backward"].nfe == per_step * n_steps
    e_gp, e_gx = rel_err(gp, gp_ref), rel_err(xp.grad, gx_ref)
    print("%s d=%d H=%d rows=%d: %d steps, loss %.6f / %.6f, grad_theta %.2e grad_x %.2e"
          % (solver, dim, hidden, rows, n_steps, loss.item(), loss_ref, e_gp, e_gx))
    assert abs(loss.item() - loss_ref) < 2e-5 * max(1.0, abs(loss_ref))
    assert e_gp < 5e-5 and e_gx < 5e-5


@pytest.mark.parametrize("solver", ["rk4", "euler", "midpoint"])
@pytest.mark.parametrize("case", ["d2", "d2_ragged", "d1", "d2_hutch", "d2_residual_reverse", "d2_one_step", "stiff_d1"])
def test_fixed_grid_one_tile_per_workgroup_kernels(solver, case):
    """Fixed-grid solves of up to 4096 rows run one tile per WORKGROUP (fixed_split_tw_kernel, csrc/cnf_split_fwd_tw.h;
    fixed_bwd_split_kernel, csrc/cnf_split.hip: the feature-split evaluations of the dopri5 kernels inside rk4_kernel's /
    fixed_bwd_kernel's loops).  Held here against the one-wave-per-tile kernels (CNF_FLAG_FWD_FOUR_WAVES keeps them for such
    batches), which test_fixed_grid_adjoint_matches_oracle's 5000-row case and test_rk4_parity_* hold against the oracle: same
    grid and step counts; states and gradients equal up to the hidden layers' arithmetic (three-piece bf16 products summed in
    fp32 against the fp32 MFMA chain).  The adjoint in both of its launches: four waves per tile, and CNF_FLAG_SPREAD's two
    eight-wave workgroups per tile, whose sums must equal the four-wave launch's exactly in the matrices' halves each keeps."""
    nat = _native()
    g = torch.Generator().manual_seed(23)
    dim = 1 if "d1" in case else 2
    hutch = "hutch" in case
    rows = {"d2": 2048, "d2_ragged": 203, "d1": 4096, "d2_hutch": 500, "d2_residual_reverse": 777, "d2_one_step": 64, "stiff_d1": 96}[case]
    if case == "stiff_d1":
        m = oracle_fixture_model("multimodal")
        x = torch.rand(rows, 1, generator=g) * 76.0 + 7.0
        step = {"rk4": 0.004, "midpoint": 0.002, "euler": 0.001}[solver]
    else:
        m = oracle_random_model(dim, seed=19, divergence_fn="approximate" if hutch else "brute_force", scale=1.5)
        x = torch.randn(rows, dim, generator=g) * 1.2
        step = None if case == "d2_one_step" else 0.07
    net = nat.net_desc(dim, 64, 3, hutch, residual="residual" in case)
    params = _flat(m).to(DEV)
    x = x.to(DEV)
    e = torch.randn(rows, dim, generator=g).to(DEV) if hutch else None
    t0, t1 = (0.5, 0.0) if "reverse" in case else (0.0, 0.5)
    gz = torch.randn(rows, dim, generator=g).to(DEV) / rows
    gl = torch.randn(rows, 1, generator=g).to(DEV) / rows
    out = {}
    for name, flags in (("split", 0), ("spread", nat.FLAG_SPREAD), ("wave", nat.FLAG_FWD_FOUR_WAVES)):
        opts = nat.make_opts(solver, step_size=step, flags=flags)
        z, lp, st = nat.solve_fwd(net, params, x, None, t0, t1, opts, e)
        gx, glp, gp, stb = nat.solve_bwd(net, params, z, lp, gz, gl, t0, t1, opts, e)
        torch.cuda.synchronize()
        st, stb = nat.Stats(st), nat.Stats(stb)
        st.raise_for_status()
        stb.raise_for_status()
        out[name] = (z.clone(), lp.clone(), gx.clone(), glp.clone(), gp.clone(),
                     (st.nfe, st.n_accepted, st.evals_run, stb.nfe, stb.n_accepted, stb.evals_run))
    per_step = {"rk4": 4, "euler": 1, "midpoint": 2}[solver]
    n_steps = out["wave"][5][1]
    assert out["split"][5] == out["wave"][5] == out["spread"][5] == (per_step * n_steps, n_steps, per_step * n_steps) * 2
    assert n_steps == 1 if step is None else n_steps >= 7
    # the forward solve does not depend on the adjoint's launch shape; the state half of the adjoint up to rounding at most
    assert torch.equal(out["split"][0], out["spread"][0]) and torch.equal(out["split"][1], out["spread"][1])
    assert rel_err(out["spread"][2], out["split"][2]) < 1e-6 and torch.equal(out["split"][3], out["spread"][3])
    errs = [rel_err(out["split"][i], out["wave"][i]) for i in range(5)]
    e_sp = rel_err(out["spread"][4], out["split"][4])
    print("%s %s rows=%d steps=%d: split vs one-wave-per-tile z %.1e logp %.1e grad_x %.1e grad_logp %.1e grad_theta %.1e | spread vs split grad_theta %.1e"
          % (solver, case, rows, n_steps, *errs, e_sp))
    bar = 2e-4 if case == "stiff_d1" else 5e-6   # the stiff field amplifies rounding differences over its 125..500 steps
    assert errs[0] < bar and errs[1] < bar and errs[2] < 4 * bar and errs[4] < 4 * bar
    assert torch.equal(out["split"][3], out["wave"][3])   # a_logp is constant
    assert e_sp < 1e-6


def test_fixed_grid_adjoint_limits():
    """What the fixed-grid adjoint does not cover is refused with a message, before anything is launched."""
    nat = _native()
    net = nat.net_desc(16, 64, 3, True)
    p = torch.zeros(nat.param_count(net), device=DEV)
    z = torch.zeros(5000, 16, device=DEV)
    l = torch.zeros(5000, 1, device=DEV)
    with pytest.raises(nat.CnfError, match="at most"):      # d = 16: one workgroup per tile, <= 4096 rows
        nat.solve_bwd(net, p, z, l, z, l, 0.0, 0.5, nat.make_opts("rk4", step_size=0.1), e=z)
    net = nat.net_desc(2, 64, 3, False)
    p = torch.zeros(nat.param_count(net), device=DEV)
    z = torch.zeros(70000, 2, device=DEV)
    l = torch.zeros(70000, 1, device=DEV)
    with pytest.raises(nat.CnfError, match="fixed-grid adjoint supports at most"):
        nat.solve_bwd(net, p, z, l, z, l, 0.0, 0.5, nat.make_opts("euler"))


@pytest.mark.parametrize("solver,step", [("rk4", 0.004), ("midpoint", 0.002)])
def test_fixed_grid_adjoint_on_the_trained_stiff_field(solver, step):
    """The fixed-grid adjoint over MANY steps on the reference's trained gu_num=8 field (checkpoint weights, d = 1; the
    regime config 2 lives in): 125 / 250 steps, sums accumulated in fp32 over all of them.  Bar as in the stiff dopri5 tests:
    grad_theta no further from the float64 oracle (same method, same grid) than 2x the fp32 oracle is; grad_x (per row) by the
    percentiles of its row errors against the fp32 oracle's own."""
    import copy
    from gan_flow_amd import gu_ffjord
    mo = oracle_fixture_model("multimodal")
    mo.train()
    mp = product_like(mo, DEV)
    m64 = copy.deepcopy(mo).double()
    for m_ in (mo, mp, m64):
        for blk in m_.chain:
            blk.solver = solver
            blk.solver_options = {"step_size": step}
    g = torch.Generator().manual_seed(9)
    x = torch.rand(96, 1, generator=g) * 76.0 + 7.0
    loss_ref, _, _, gx_ref, gp_ref, st_ref = _oracle_train_grads(mo, x)
    m64.zero_grad()
    x64 = x.double().requires_grad_(True)
    z64, dl64 = m64(x64, torch.zeros(96, 1, dtype=torch.float64))
    _loss(z64, dl64).backward()
    gp64 = torch.cat([p.grad.reshape(-1) for p in _odefunc(m64).diffeq.parameters()])
    gx64 = x64.grad
    xp = x.to(DEV).requires_grad_(True)
    loss = gu_ffjord.compute_loss(mp, xp)
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()])
    st = mp.chain[0].last_stats["backward"]
    assert st.status == 0 and st.n_accepted == st_ref["backward"]["n_steps"]
    d_hip_p, d_or_p = rel_err(gp, gp64), rel_err(gp_ref, gp64)
    d_hip_x, d_or_x = rel_err(xp.grad, gx64), rel_err(gx_ref, gx64)
    print("%s h=%g: %d steps | grad_theta vs float64: hip %.2e oracle-fp32 %.2e | grad_x: hip %.2e oracle-fp32 %.2e | hip vs oracle-fp32 %.2e %.2e"
          % (solver, step, st.n_accepted, d_hip_p, d_or_p, d_hip_x, d_or_x, rel_err(gp, gp_ref), rel_err(xp.grad, gx_ref)))
    # grad_x is a per-row quantity on a field that amplifies rounding differently row by row: state the distribution
    sx = gx64.abs().max().item()
    eh = ((xp.grad.cpu().double() - gx64).abs() / sx).reshape(-1)
    eo = ((gx_ref.double() - gx64).abs() / sx).reshape(-1)
    qs = torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64)
    qh, qo = torch.quantile(eh, qs), torch.quantile(eo, qs)
    print("   grad_x row errors / max|grad_x|, 50/90/99th percentile and max: hip %s %.2e | oracle-fp32 %s %.2e | worst rows hip %s oracle %s"
          % (["%.1e" % v for v in qh.tolist()], eh.max().item(), ["%.1e" % v for v in qo.tolist()], eo.max().item(),
             eh.argsort(descending=True)[:3].tolist(), eo.argsort(descending=True)[:3].tolist()))
    assert d_hip_p <= 2 * d_or_p + 2e-6
    # same quantile within 1.5x (median, 90th), tail and worst row bounded by 2x / 4x the oracle's: the worst row is one draw
    # of rounding noise amplified by that row's trajectory (measured: rk4 5.3e-4 against 1.4e-4 in one row, midpoint 3.0e-4
    # against 3.4e-4), the body of the distribution is what shows a systematic difference
    assert qh[0] <= 1.5 * qo[0] + 2e-6 and qh[1] <= 1.5 * qo[1] + 2e-6 and qh[2] <= 2 * qo[2] + 2e-6
    assert eh.max() <= 4 * eo.max() + 2e-6
    assert abs(loss.item() - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))


@pytest.mark.parametrize("case", ["d2_l1", "d2_l2", "d2_l1_l2", "d1_l2_l1", "d2_hutch_l2", "d2_residual_l1_l2", "d2_l2_5000"])
def test_regularised_dynamics_match_oracle(case):
    """RegularizedODEfunc with l1 / l2 kinetic regularisers (cnf_solve_fwd_reg / cnf_solve_bwd_reg):
    every regulariser is one more state of the solve -- the batch mean of a function of dz/dt, tolerances 1e20, but part of
    the first-step heuristic of both solves -- and its constant adjoint one more cotangent on dz/dt in the backward sweep.
    Against the oracle's restatement (oracle.l1_regularzation_fn / l2_regularzation_fn through odeint_adjoint) on identical
    weights: the regularisation states, the solves' step counts, loss and gradients of nll + sum_k c_k r_k.
    Parity unpinned: the reference's drivers pass regularization_fns=() and no artefact records a regularised run."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.lib.layers.wrappers import cnf_regularization as creg
    dim = 1 if case.startswith("d1") else 2
    hutch, residual = "hutch" in case, "residual" in case
    kinds = [k for k in case.split("_") if k in ("l1", "l2")]
    rows = 5000 if case.endswith("5000") else 300
    div = "approximate" if hutch else "brute_force"
    o_fns = tuple({"l1": oracle.l1_regularzation_fn, "l2": oracle.l2_regularzation_fn}[k] for k in kinds)
    p_fns = tuple({"l1": creg.l1_regularzation_fn, "l2": creg.l2_regularzation_fn}[k] for k in kinds)
    mo = oracle_random_model(dim, seed=29, scale=1.5, divergence_fn=div, residual=residual, regularization_fns=o_fns)
    mo.train()
    mp = gu_ffjord.build_model(dim, divergence_fn=div, residual=residual, regularization_fns=p_fns)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(rows, dim, generator=g) * 1.5
    if residual:
        # dz/dt = f(z) - z is ~ -z here, and both integrands have a kink at dz/dt = 0 (|.| and the norm): a row that starts
        # within rounding of it gets an ill-conditioned cotangent (direction of a near-zero vector) in the oracle and the
        # kernel alike -- measured: one row of 300 at x = (-2e-3, 1e-4), 7 % off, every other row 1e-6.  Keep the rows off it.
        x = x + 4.0
    e = torch.randn(rows, dim, generator=g) if hutch else None
    coef = [0.5, 0.25][:len(kinds)]
    fo, fp = _odefunc(mo), mp.chain[0].odefunc.odefunc
    if hutch:
        orig = fo.before_odeint
        fo.before_odeint = lambda e_=None: orig(e=e)
        origp = fp.before_odeint
        fp.before_odeint = lambda e_=None: origp(e=e.to(DEV))

    def run(model, dev):
        model.zero_grad()
        xx = x.to(dev).clone().requires_grad_(True)
        z, dl = model(xx, torch.zeros(rows, 1, device=dev))
        regs = model.chain[0].get_regularization_states()
        assert len(regs) == len(kinds)
        loss = _loss(z, dl) + sum(c * r for c, r in zip(coef, regs))
        loss.backward()
        f = model.chain[0].odefunc.odefunc
        gp = torch.cat([p.grad.reshape(-1) for p in f.diffeq.parameters()])
        return loss.item(), z.detach().cpu(), dl.detach().cpu(), [float(r) for r in regs], xx.grad.cpu(), gp.cpu()

    lo, zo, dlo, ro, gxo, gpo = run(mo, "cpu")
    lp, zp, dlp, rp, gxp, gpp = run(mp, DEV)
    so, sp = mo.chain[0].stats, mp.chain[0].last_stats
    print("%s: regs hip %s oracle %s | steps fwd %d+%d / %d+%d bwd %d+%d / %d+%d | loss %.2e z %.1e logp %.1e grad_x %.2e grad_theta %.2e"
          % (case, ["%.6f" % r for r in rp], ["%.6f" % r for r in ro], sp["forward"].n_accepted, sp["forward"].n_rejected,
             so["forward"]["n_accepted"], so["forward"]["n_rejected"], sp["backward"].n_accepted, sp["backward"].n_rejected,
             so["backward"]["n_accepted"], so["backward"]["n_rejected"], abs(lp - lo), rel_err(zp, zo), rel_err(dlp, dlo),
             rel_err(gxp, gxo), rel_err(gpp, gpo)))
    assert (sp["forward"].n_accepted, sp["forward"].n_rejected) == (so["forward"]["n_accepted"], so["forward"]["n_rejected"])
    assert (sp["backward"].n_accepted, sp["backward"].n_rejected) == (so["backward"]["n_accepted"], so["backward"]["n_rejected"])
    for a_, b_ in zip(rp, ro):
        assert abs(a_ - b_) < 1e-5 * max(1.0, abs(b_))
    assert abs(lp - lo) < 1e-4 * max(1.0, abs(lo)) and rel_err(zp, zo) < 2e-5 and rel_err(dlp, dlo) < 1e-4
    assert rel_err(gxp, gxo) < 2e-4 and rel_err(gpp, gpo) < 2e-4
    # evaluation mode integrates the two states only, as the reference's test branch does
    mp.eval()
    with torch.no_grad():
        mp(x.to(DEV), torch.zeros(rows, 1, device=DEV))
    assert mp.chain[0].get_regularization_states() == ()
