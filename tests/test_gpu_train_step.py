"""GPU tests of the loss and optimizer kernels (cnf_nll_loss, cnf_adam_step) and of
gu_ffjord.FusedTrainStep, the training iteration built from library kernels only.  References: the oracle's loss
expression (oracle/layers.py, CPU) and torch.optim.Adam on the CPU; the fused iteration is compared with the
autograd + torch.optim.Adam loop over the same HIP solves."""
import pytest
import torch

import oracle
from _helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _native():
    from gan_flow_amd import _native
    _native.load_library()
    return _native


@pytest.mark.parametrize("rows,dim", [(2048, 2), (1, 1), (1000, 16), (70001, 2)])
def test_nll_loss_and_gradient(rows, dim):
    nat = _native()
    g = torch.Generator().manual_seed(rows + dim)
    z = (torch.randn(rows, dim, generator=g) * 1.7).requires_grad_(True)
    dl = torch.randn(rows, 1, generator=g).requires_grad_(True)
    ref = -torch.mean(oracle.layers.standard_normal_logprob(z).view(rows, -1).sum(1, keepdim=True) - dl)
    ref.backward()
    loss, gz, gl = nat.nll_loss(z.detach().to(DEV), dl.detach().to(DEV))
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
    assert rel_err(gz, z.grad) < 1e-6 and rel_err(gl, dl.grad) < 1e-6
    loss2, none_z, none_l = nat.nll_loss(z.detach().to(DEV), dl.detach().to(DEV), want_grads=False)
    assert none_z is None and none_l is None and loss2.item() == loss.item()      # deterministic


@pytest.mark.parametrize("n,wd", [(9224, 1e-5), (300, 0.0), (200003, 1e-2)])
def test_adam_step_matches_torch_optim(n, wd):
    nat = _native()
    g = torch.Generator().manual_seed(n)
    p_ref = torch.nn.Parameter(torch.randn(n, generator=g))
    opt = torch.optim.Adam([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    p = p_ref.detach().clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    for it in range(6):
        grad = torch.randn(n, generator=g) * (10.0 ** (it - 3))
        p_ref.grad = grad.clone()
        opt.step()
        nat.adam_step(p, grad.to(DEV), m, v, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    st = opt.state[p_ref]
    assert state.tolist() == [6, 0]
    assert torch.allclose(p.cpu(), p_ref.detach(), rtol=1e-6, atol=1e-7)
    assert rel_err(m, st["exp_avg"]) < 1e-6 and rel_err(v, st["exp_avg_sq"]) < 1e-6
    assert torch.allclose(v.cpu(), st["exp_avg_sq"], rtol=2e-6, atol=0)


def test_adam_step_rejects_bad_arguments():
    nat = _native()
    p = torch.zeros(8, device=DEV)
    st = torch.zeros(2, dtype=torch.int32, device=DEV)
    with pytest.raises(nat.CnfError):
        nat.adam_step(p, p.clone(), p.clone(), p.clone(), st, betas=(1.0, 0.999))
    with pytest.raises(nat.CnfError):
        nat.adam_step(p, torch.zeros(7, device=DEV), p.clone(), p.clone(), st)
    with pytest.raises(nat.CnfError):
        nat.adam_step(p, p.clone(), p.clone(), p.clone(), torch.zeros(2, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize("graph,blocks", [(False, 1), (True, 1), (False, 2)])
def test_fused_train_step_matches_autograd_loop(graph, blocks):
    """Same batches, same initial weights: the fused iteration follows the trajectory of compute_loss / backward /
    torch.optim.Adam (which run the same solves) to rounding, and keeps the module's parameters up to date."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(8, seed=3)
    batches = [data.get_sample(2048).to(DEV) for _ in range(8)]

    torch.manual_seed(5)
    ref = gu_ffjord.build_model(2, num_blocks=blocks).to(DEV).train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x in batches:
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(ref, x)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    nfe_ref = (sum(b.last_stats["forward"].nfe for b in ref.chain), sum(b.last_stats["backward"].nfe for b in ref.chain))

    torch.manual_seed(5)
    model = gu_ffjord.build_model(2, num_blocks=blocks).to(DEV).train()
    t_before = model.chain[0].sqrt_end_time.detach().clone()
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=graph)
    losses = [step(x).item() for x in batches]
    nfe = step.check()
    print("reference", ["%.5f" % v for v in ref_losses])
    print("fused    ", ["%.5f" % v for v in losses])
    assert step.step_count == len(batches)
    assert nfe == nfe_ref
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(ref_losses, losses))
    pr = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    pf = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    print("max |dp| %.3e (updates moved the weights by %.3e)" % ((pr - pf).abs().max().item(), 8e-3))
    assert (pr - pf).abs().max().item() < 5e-5
    assert torch.equal(model.chain[0].sqrt_end_time.detach(), t_before)       # T gets no gradient, as in the reference
    # the module evaluates with the trained weights (parameters are views of the updated buffer)
    model.eval()
    ref.eval()
    xe = batches[0][:256]
    z1, dl1 = model(xe, torch.zeros(256, 1, device=DEV))
    z2, dl2 = ref(xe, torch.zeros(256, 1, device=DEV))
    assert rel_err(z1, z2) < 1e-4 and rel_err(dl1, dl2) < 1e-3
    assert model.chain[0].num_evals() > 0


@pytest.mark.parametrize("layer_type,dims,act,graph", [("concatsquash", (32, 48, 32), "tanh", True), ("squash", (64, 64, 64), "tanh", False),
                                                       ("ignore", (20, 20, 20), "softplus", True), ("concat_v2", (40, 64, 64), "elu", False)])
def test_fused_train_step_other_layer_types_and_padded_widths(layer_type, dims, act, graph):
    """FusedTrainStep on nets whose flat kernel buffer holds more than parameters: zero padding (hidden widths below the
    kernel width) and the constants by which 'ignore' / 'concat_v2' / 'squash' layers present themselves as concatsquash.
    ODEnet.flatten_parameters() aliases every parameter to its place (a padded weight as a strided block) and lists the rest;
    the step restores those entries after each update (padded features of a softplus net have non-zero gradients, and weight
    decay would move the gate-open constant).  Same trajectory as the autograd + torch.optim.Adam loop, which slices the
    kernels' gradient back to the real parameters."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(8, seed=3)
    batches = [data.get_sample(1024).to(DEV) for _ in range(6)]

    def build():
        torch.manual_seed(7)
        return gu_ffjord.build_model(2, dims=dims, layer_type=layer_type, nonlinearity=act).to(DEV).train()

    ref = build()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x in batches:
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(ref, x)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    model = build()
    names = [k for k, _ in model.named_parameters()]
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=graph)
    net = model.chain[0].odefunc.odefunc.diffeq
    consts = net.flat_constants
    assert consts is not None and consts[0].numel() > 0
    losses = [step(x).item() for x in batches]
    step.check()
    print("reference", ["%.5f" % v for v in ref_losses])
    print("fused    ", ["%.5f" % v for v in losses])
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(ref_losses, losses))
    assert [k for k, _ in model.named_parameters()] == names
    for (k, a), (_, b) in zip(ref.named_parameters(), model.named_parameters()):
        assert a.shape == b.shape and (a - b).abs().max().item() < 5e-5, k
    # what is not a parameter is what it was: padding zero, constants constant
    assert torch.equal(step.flats[0][consts[0]], consts[1])
    # and the kernels still see the module's parameters: evaluation through the module equals the reference model's
    model.eval()
    ref.eval()
    xe = batches[0][:256]
    z1, dl1 = model(xe, torch.zeros(256, 1, device=DEV))
    z2, dl2 = ref(xe, torch.zeros(256, 1, device=DEV))
    assert rel_err(z1, z2) < 1e-4 and rel_err(dl1, dl2) < 1e-3


@pytest.mark.parametrize("graph,blocks,kinds", [(True, 1, ("l1", "l2")), (False, 2, ("l2",))])
def test_fused_train_step_with_kinetic_regularisers_matches_autograd_loop(graph, blocks, kinds):
    """`--l1int` / `--l2int` through FusedTrainStep: cnf_solve_fwd_reg, the loss nll + sum c_k r_k, cnf_solve_bwd_reg with the
    constant cotangents c_k, cnf_adam_step -- library kernels only, captured or eager -- against the autograd +
    torch.optim.Adam loop over the same solves (compute_loss + get_regularization)."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    from gan_flow_amd.lib.layers.wrappers import cnf_regularization as creg
    fns = tuple({"l1": creg.l1_regularzation_fn, "l2": creg.l2_regularzation_fn}[k] for k in kinds)
    coeffs = [0.5, 0.25][:len(kinds)]
    data = make_gu(8, seed=3)
    batches = [data.get_sample(1024).to(DEV) for _ in range(6)]

    def build():
        torch.manual_seed(5)
        return gu_ffjord.build_model(2, num_blocks=blocks, regularization_fns=fns).to(DEV).train()

    ref = build()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x in batches:
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(ref, x)
        loss = loss + gu_ffjord.get_regularization(ref, coeffs)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    model = build()
    with pytest.raises(ValueError, match="reg_coeffs"):
        gu_ffjord.FusedTrainStep(model, batches[0], graph=False)
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=graph, reg_coeffs=coeffs)
    losses = [step(x).item() for x in batches]
    nfe = step.check()
    print("reference", ["%.5f" % v for v in ref_losses])
    print("fused    ", ["%.5f" % v for v in losses], nfe)
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(ref_losses, losses))
    pr = torch.cat([p.detach().reshape(-1) for p in ref.parameters() if p.requires_grad])
    pf = torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])
    assert (pr - pf).abs().max().item() < 5e-5


@pytest.mark.parametrize("solver,step_size,graph,blocks", [("rk4", 0.05, True, 1), ("rk4", 0.125, False, 2), ("midpoint", 0.05, True, 1),
                                                           ("euler", None, False, 1)])
def test_fused_train_step_fixed_grid_matches_autograd_loop(solver, step_size, graph, blocks):
    """`--solver rk4 | midpoint | euler` through FusedTrainStep: the one-tile-per-workgroup fixed-grid kernels
    (fixed_split_tw_kernel, fixed_bwd_split_kernel with CNF_FLAG_SPREAD), cnf_nll_loss for the cotangents and
    cnf_solve_bwd_train's slab sum + Adam.  Same batches and initial weights as the autograd + torch.optim.Adam loop over the
    same solves: same trajectory to rounding; and the two-kernel tail (fuse_tail) gives the bits of cnf_solve_bwd +
    cnf_adam_step."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(8, seed=3)
    batches = [data.get_sample(2048).to(DEV) for _ in range(6)]

    def build():
        torch.manual_seed(5)
        m = gu_ffjord.build_model(2, num_blocks=blocks, solver=solver).to(DEV).train()
        for blk in m.chain:
            blk.solver_options = {"step_size": step_size} if step_size else {}
            blk.spread_adjoint = True
        return m

    ref = build()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x in batches:
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(ref, x)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    nfe_ref = (sum(b.last_stats["forward"].nfe for b in ref.chain), sum(b.last_stats["backward"].nfe for b in ref.chain))
    runs = []
    for fuse in (True, False):
        model = build()
        step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=graph and fuse, fuse_tail=fuse)
        losses = [step(x).item() for x in batches]
        nfe = step.check()
        runs.append((losses, [f.clone() for f in step.flats], [m.clone() for m in step.exp_avg]))
        assert step.step_count == len(batches) and nfe == nfe_ref
    losses, flats, _ = runs[0]
    print("reference", ["%.5f" % v for v in ref_losses])
    print("fused    ", ["%.5f" % v for v in losses])
    per_step = {"rk4": 4, "euler": 1, "midpoint": 2}[solver]
    assert nfe_ref[0] % (per_step * blocks) == 0 and nfe_ref[0] == nfe_ref[1]
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(ref_losses, losses))
    pr = torch.cat([p.detach().reshape(-1) for p in ref.parameters() if p.requires_grad])
    pf = torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])
    assert (pr - pf).abs().max().item() < 5e-5
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1] + runs[0][2], runs[1][1] + runs[1][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dim,blocks,rows,div", [(2, 1, 2048, "brute_force"), (2, 2, 1000, "brute_force"), (1, 1, 77, "brute_force"),
                                                 (16, 1, 2048, "approximate"), (2, 1, 300, "approximate")])
def test_solve_bwd_train_equals_separate_calls(dim, blocks, rows, div):
    """cnf_solve_bwd_train (loss + cotangents inside the adjoint kernel, Adam as the epilogue of the slab sum) against
    cnf_nll_loss / cnf_solve_bwd / cnf_adam_step: parameters and Adam moments identical bit for bit after every step,
    losses equal to the rounding of a different summation order."""
    from gan_flow_amd import gu_ffjord
    g = torch.Generator().manual_seed(9)
    batches = [(torch.randn(rows, dim, generator=g) * 1.5 + 0.3).to(DEV) for _ in range(4)]
    runs = []
    for fuse in (False, True):
        torch.manual_seed(6)
        model = gu_ffjord.build_model(dim, num_blocks=blocks, divergence_fn=div, rademacher=True).to(DEV).train()
        step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=False, fuse_tail=fuse)
        if div == "approximate":     # same Rademacher probes for both runs
            torch.manual_seed(77)
        losses = [step(x).item() for x in batches]
        nfe = step.check()
        runs.append((losses, nfe, [f.clone() for f in step.flats], [m.clone() for m in step.exp_avg],
                     [v.clone() for v in step.exp_avg_sq], step.step_count))
    (l0, n0, p0, m0, v0, c0), (l1, n1, p1, m1, v1, c1) = runs
    assert n0 == n1 and c0 == c1 == len(batches)
    assert all(abs(a - b) <= 2e-6 * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    for a, b in zip(p0 + m0 + v0, p1 + m1 + v1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("hidden,blocks,fail_in", [(64, 1, "forward"), (64, 1, "adjoint"), (64, 2, "forward"), (256, 1, "forward"),
                                                   (256, 1, "adjoint")])
def test_failed_solve_leaves_parameters_and_optimizer_untouched(hidden, blocks, fail_in):
    """A solve that fails (here: max_num_steps exceeded) leaves its outputs unwritten.  Every later solve of the iteration
    names the one before it (CnfSolverOpts.after) and exits at once with the same status, and the Adam kernels skip
    themselves on a non-zero status: parameters, moments and step counts keep their bits, .check() raises, and the step
    after the limit is lifted trains normally.  Both tails: cnf_solve_bwd_train (hidden 64) and cnf_solve_bwd +
    cnf_adam_step (hidden 256)."""
    from gan_flow_amd import _native, gu_ffjord
    g = torch.Generator().manual_seed(3)
    xs = [(torch.randn(512, 2, generator=g) * 1.5).to(DEV) for _ in range(3)]
    torch.manual_seed(4)
    model = gu_ffjord.build_model(2, dims=(hidden,) * 3, num_blocks=blocks).to(DEV).train()
    step = gu_ffjord.FusedTrainStep(model, xs[0], lr=1e-3, weight_decay=1e-5, graph=False)
    step(xs[0])
    assert step.check()[0] > 0 and step.step_count == 1
    before = [t.clone() for t in step.flats + step.exp_avg + step.exp_avg_sq + step.adam_state]
    # poison what a failed forward would leave unwritten, then make the first block's solves fail after one attempt
    blk = model.chain[0]
    if fail_in == "forward":
        blk.solver_options["max_num_steps"] = 1
        step(xs[1])
    else:
        # the adjoint alone: run the forward normally, then hand the adjoint a one-attempt limit through the raw calls
        nat = _native
        net, t0, t1, opts, e = blk.solve_plan(xs[1], reset_counter=False)
        z1, l1, sf = nat.solve_fwd(net, step.flats[0], xs[1], None, t0, t1, opts)
        lim = nat.make_opts("dopri5", 1e-5, 1e-5, max_steps=1, after=sf)
        st = torch.zeros(8, dtype=torch.int32, device=DEV)
        if hidden == 64:
            nat.solve_bwd_train(net, step.flats[0], z1, l1, None, None, t0, t1, lim, step.exp_avg[0], step.exp_avg_sq[0],
                                step.adam_state[0], stats=st, **step.hyper)
        else:
            gz, gl = z1 / 512, torch.full_like(l1, 1.0 / 512)
            gx, _, gp, st = nat.solve_bwd(net, step.flats[0], z1, l1, gz, gl, t0, t1, lim)
            nat.adam_step(step.flats[0], gp, step.exp_avg[0], step.exp_avg_sq[0], step.adam_state[0], guard=st, **step.hyper)
        torch.cuda.synchronize()
        assert nat.Stats(st).status == 3
    torch.cuda.synchronize()
    after = step.flats + step.exp_avg + step.exp_avg_sq + step.adam_state
    for a, b in zip(before, after):
        assert torch.equal(a, b)
    if fail_in == "forward":
        with pytest.raises(_native.CnfError, match="max_num_steps"):
            step.check()
        words = step.stats.cpu()
        assert all(int(words[i, k, 0]) == 3 for i in range(blocks) for k in (0, 1))     # the status travelled down the chain
        assert all(int(words[i, k, 1]) <= 8 for i in range(blocks) for k in (0, 1) if (i, k) != (0, 0))   # nothing else integrated
        blk.solver_options.pop("max_num_steps")
        step.stats.zero_()           # the status word is sticky under CNF_FLAG_CLEAN: the owner clears it after a failure
    step(xs[2])
    assert step.check()[0] > 0 and step.step_count == 2
    assert not torch.equal(before[0], step.flats[0])


def test_fused_train_step_two_blocks_hutchinson():
    """Stacked blocks + Hutchinson divergence: the loss gradient is chained through the blocks in reverse."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(8, seed=4)
    batches = [data.get_sample(1024).to(DEV) for _ in range(12)]
    torch.manual_seed(2)
    model = gu_ffjord.build_model(2, num_blocks=2, divergence_fn="approximate").to(DEV).train()
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=2e-3, graph=False)
    losses = [step(x).item() for x in batches]
    step.check()
    print("two-block losses", ["%.3f" % v for v in losses])
    assert all(torch.isfinite(torch.tensor(losses))) and min(losses[-3:]) < losses[0]


@pytest.mark.parametrize("graph", [False, True])
def test_fused_train_step_d16_hutchinson_gmm(graph):
    """BASELINE config 5 in small: d=16 Gaussian mixture, Hutchinson divergence (a fresh probe every iteration, fixed
    within it for the forward and the adjoint solve), batch 2048.  The loss moves toward the data entropy."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gmm import GaussianMixture
    data = GaussianMixture(16, device=DEV, sample_seed=3)
    floor = data.entropy_estimate(50000)
    torch.manual_seed(3)
    model = gu_ffjord.build_model(16, divergence_fn="approximate").to(DEV).train()
    step = gu_ffjord.FusedTrainStep(model, data.get_sample(2048), lr=1e-3, weight_decay=1e-5, graph=graph)
    losses = [step(data.get_sample(2048)).item() for _ in range(40)]
    nfe = step.check()
    print("d16 losses", ["%.2f" % v for v in losses[::5]], "floor %.2f" % floor, "NFE", nfe)
    assert all(torch.isfinite(torch.tensor(losses)))
    assert sum(losses[-5:]) / 5 < sum(losses[:5]) / 5 - 1.0 and min(losses) > floor
    e1 = model.chain[0].odefunc.odefunc._e
    step(data.get_sample(2048))
    e2 = model.chain[0].odefunc.odefunc._e
    if not graph:
        assert e1 is not e2 and not torch.equal(e1, e2)       # a new probe per iteration


@pytest.mark.parametrize("solver,step_size", [("rk4", 0.1), ("midpoint", 0.05)])
def test_fused_train_step_d16_fixed_grid_matches_autograd_loop(solver, step_size):
    """d = 16 with a fixed-grid solver through FusedTrainStep (gmm_ffjord --solver rk4): the unfused tail -- cnf_nll_loss,
    cnf_solve_bwd on fixed_bwd_split16_kernel, cnf_adam_step -- in one captured iteration, against the autograd +
    torch.optim.Adam loop over the same solves with the same Rademacher probes."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gmm import GaussianMixture
    data = GaussianMixture(16, device=DEV, sample_seed=3)
    batches = [data.get_sample(2048) for _ in range(5)]

    def build():
        torch.manual_seed(3)
        m = gu_ffjord.build_model(16, divergence_fn="approximate", solver=solver, rademacher=True).to(DEV).train()
        for blk in m.chain:
            blk.solver_options = {"step_size": step_size}
        return m

    ref = build()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    torch.manual_seed(77)
    ref_losses = []
    for x in batches:
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(ref, x)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    model = build()
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=False)
    torch.manual_seed(77)
    losses = [step(x).item() for x in batches]
    nfe = step.check()
    print("reference", ["%.5f" % v for v in ref_losses])
    print("fused    ", ["%.5f" % v for v in losses], nfe)
    assert nfe[0] == nfe[1] == {"rk4": 4, "midpoint": 2}[solver] * round(0.5 / step_size)
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(ref_losses, losses))
    pr = torch.cat([p.detach().reshape(-1) for p in ref.parameters() if p.requires_grad])
    pf = torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])
    assert (pr - pf).abs().max().item() < 5e-5


def test_concurrent_trials_match_sequential():
    """Two trainings replayed on two streams of one GPU (gu_ffjord.ConcurrentTrials) follow exactly the trajectories they
    follow alone: the kernels are deterministic and share nothing but the device."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    datas = [make_gu(8, seed=11), make_gu(8, seed=12)]
    batches = [[d.get_sample(2048).to(DEV) for _ in range(10)] for d in datas]

    def build(seed, bs, spread=False):
        torch.manual_seed(seed)
        model = gu_ffjord.build_model(2).to(DEV).train()
        return model, gu_ffjord.FusedTrainStep(model, bs[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=spread)

    alone = []
    for k in range(2):
        _, step = build(20 + k, batches[k], spread=True)   # alone: two workgroups per tile (CNF_FLAG_SPREAD) -- same trajectory
        alone.append([step(x).item() for x in batches[k]])
        step.check()
    pairs = [build(20 + k, batches[k]) for k in range(2)]
    both = gu_ffjord.ConcurrentTrials([p[1] for p in pairs])
    together = [[], []]
    for i in range(10):
        la, lb = both([batches[0][i], batches[1][i]])
        torch.cuda.synchronize()
        together[0].append(la.item())
        together[1].append(lb.item())
    both.check()
    assert together == alone
    with pytest.raises(ValueError):
        gu_ffjord.ConcurrentTrials([p[1] for p in pairs] * 2)        # 4 x 128 workgroups do not fit 256 compute units
    with pytest.raises(ValueError):
        gu_ffjord.ConcurrentTrials([pairs[0][1], step])              # a spread trial cannot share the GPU
