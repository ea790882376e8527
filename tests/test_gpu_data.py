"""GPU: the study's data and metric on the device.

  * cnf_gu_sample (the mixture of /root/reference/gu.py:35-68 drawn by a HIP kernel) against the analytic CDF of that
    mixture (Kolmogorov-Smirnov), against the bit-exact host restatement (two-sample KS), and against the moments of the
    vectors generated from the reference itself (tests/golden/gu_samples.npz).  The random stream differs by construction
    (numpy's global state cannot be matched on a device): the bar is equality IN DISTRIBUTION;
  * util.w_distance on device tensors against scipy.stats.wasserstein_distance (what /root/reference/util.py:16-20 calls);
  * FusedTrainStep fed by the device sampler inside its captured graph: a fresh batch per replay, training proceeds;
  * BASELINE configs[0] (gu_num = 1 data): the fused training step against the CPU oracle, iteration by iteration.
"""
import os

import numpy as np
import pytest
import torch
from scipy import stats

import oracle
from _helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mixture_cdf(gu, x):
    total = np.zeros_like(x, dtype=np.float64)
    for c in range(gu.n_mixture):
        lo, hi = gu._bounds(c)
        unif = np.clip((x - lo) / (hi - lo), 0.0, 1.0)
        gaus = stats.norm.cdf(x, loc=gu.mean[c], scale=gu.sigma[c])
        total += (gaus * (1 - gu.unif_p[c]) + unif * gu.unif_p[c]) / gu.n_mixture
    return total


@pytest.mark.parametrize("gu_num", [1, 8])
def test_device_sampler_draws_the_reference_distribution(gu_num):
    from gan_flow_amd.gu import make_gu
    gu = make_gu(gu_num, seed=3, extend_dim=True)
    smp = gu.device_sampler(DEV, seed=11)
    n = 400000
    a = smp.get_sample(n)
    b = smp.get_sample(n)
    torch.cuda.synchronize()
    assert a.shape == (n, 2) and not torch.equal(a, b)                      # every call draws a fresh batch
    again = gu.device_sampler(DEV, seed=11).get_sample(n)
    assert torch.equal(again, a)                                            # same seed, same call number: same batch
    assert smp.state.tolist() == [2, 0]
    x = a[:, 0].double().cpu().numpy()
    u = a[:, 1].double().cpu().numpy()
    # one-sample KS against the analytic mixture CDF and against U(0, 1) for the nuisance column
    ks = stats.kstest(x, lambda v: _mixture_cdf(gu, v))
    ku = stats.kstest(u, "uniform")
    print("gu_num %d: KS vs analytic CDF D=%.2e p=%.3f | nuisance column D=%.2e p=%.3f" % (gu_num, ks.statistic, ks.pvalue, ku.statistic, ku.pvalue))
    assert ks.pvalue > 1e-3 and ku.pvalue > 1e-3
    assert abs(np.corrcoef(x, u)[0, 1]) < 0.01 and u.min() >= 0.0 and u.max() < 1.0
    # two-sample KS against the bit-exact host restatement of the reference's sampler
    host = gu.get_sample(200000)[:, 0].double().numpy()
    k2 = stats.ks_2samp(x, host)
    assert k2.pvalue > 1e-3, k2
    # moments of the vectors generated from the reference itself (make_gu_vectors.py)
    gold = np.load(os.path.join(GOLDEN, "gu_samples.npz"))
    ref = np.concatenate([gold[("uni" if gu_num == 1 else "multi") + "_ext0_" + k].reshape(-1) for k in "ab"]).astype(np.float64)
    se = ref.std() / np.sqrt(ref.size)
    assert abs(x.mean() - ref.mean()) < 6 * se and abs(x.std() - ref.std()) < 0.1 * ref.std()
    assert stats.ks_2samp(x[:20000], ref).pvalue > 1e-3
    # analytic moments
    m_true = float(np.mean(gu.mean))
    assert abs(x.mean() - m_true) < 5 * x.std() / np.sqrt(n)


def test_device_sampler_rejects_bad_arguments():
    from gan_flow_amd import _native as nat
    from gan_flow_amd.gu import GausUniffMixture, make_gu
    smp = make_gu(8, extend_dim=False).device_sampler(DEV)
    assert smp.get_sample(100).shape == (100, 1)
    with pytest.raises(nat.CnfError):
        smp.fill(torch.empty(100, 2, device=DEV))
    with pytest.raises(nat.CnfError):
        smp.fill(torch.empty(100, 1))
    with pytest.raises(nat.CnfError):
        GausUniffMixture(17, 1.0, 1.0, 1.0).device_sampler(DEV)


def test_w_distance_on_device_matches_scipy():
    from gan_flow_amd.util import w_distance
    g = torch.Generator().manual_seed(2)
    for na, nb in ((100000, 100000), (5000, 3000)):
        a = torch.randn(na, generator=g) * 3 + 1
        b = torch.rand(nb, generator=g) * 8 - 2
        ref = float(np.round(stats.wasserstein_distance(a.numpy(), b.numpy()), 4))
        got = w_distance(a.to(DEV), b.to(DEV))
        assert abs(got - ref) <= 1.01e-4, (got, ref)


def test_fused_training_fed_by_the_device_sampler():
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2).to(DEV).train()
    smp = make_gu(8, seed=1).device_sampler(DEV)
    step = gu_ffjord.FusedTrainStep(model, torch.zeros(2048, 2, device=DEV), lr=1e-3, weight_decay=1e-5, graph=True, sampler=smp, warmup=2)
    seen, losses = [], []
    for _ in range(40):
        losses.append(step().item())
        seen.append(step.static_x[:4, 0].clone())
    nfe = step.check()
    assert step.step_count == 42 and nfe[0] >= 14
    assert all(not torch.equal(seen[i], seen[i + 1]) for i in range(len(seen) - 1))       # a fresh batch at every replay
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])
    with pytest.raises(ValueError):
        step(torch.zeros(2048, 2, device=DEV))
    # the evaluation protocol with nothing on the host: samples, real data and the sort all on the device
    wd = gu_ffjord.evaluate_w_distance(model, smp, eval_size=20000, device=torch.device(DEV))
    assert np.isfinite(wd) and wd > 0


def test_config1_unimodal_training_matches_the_oracle():
    """BASELINE configs[0]: `gu_maf --gu_num 1`, B = 2048, dopri5, default ODEnet.  Five training iterations on the same
    gu_num = 1 batches from the same seed-1 init: losses, evaluation counts and the trained weights against the CPU oracle."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.gu import make_gu
    data = make_gu(1, seed=1, extend_dim=True)
    xs = [data.get_sample(2048) for _ in range(5)]
    assert abs(float(xs[0][:, 0].mean()) - 5.0) < 0.05 and float(xs[0][:, 1].min()) >= 0.0
    torch.manual_seed(1)
    mo = oracle.build_cnf(2).train()
    mp = gu_ffjord.build_model(2)
    mp.load_state_dict(mo.state_dict())
    mp = mp.to(DEV).train()
    oopt = torch.optim.Adam(mo.parameters(), lr=1e-3, weight_decay=1e-5)
    step = gu_ffjord.FusedTrainStep(mp, xs[0].to(DEV), lr=1e-3, weight_decay=1e-5, graph=False)
    for i, x in enumerate(xs):
        oopt.zero_grad(set_to_none=True)
        z, dl = mo(x, torch.zeros(2048, 1))
        lo = -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()
        lo.backward()
        oopt.step()
        lp = step(x.to(DEV)).item()
        nf, nb = step.check()
        so = mo.chain[0].stats
        print("config 1 iteration %d: loss hip %.6f oracle %.6f | NFE hip %d/%d oracle %d/%d"
              % (i, lp, lo.item(), nf, nb, 2 + 6 * (so["forward"]["n_accepted"] + so["forward"]["n_rejected"]),
                 2 + 6 * (so["backward"]["n_accepted"] + so["backward"]["n_rejected"])))
        assert abs(lp - lo.item()) <= 2e-5 * abs(lo.item())
        assert nf == 2 + 6 * (so["forward"]["n_accepted"] + so["forward"]["n_rejected"])
        assert nb == 2 + 6 * (so["backward"]["n_accepted"] + so["backward"]["n_rejected"])
    po = torch.cat([p.detach().reshape(-1) for p in mo.chain[0].odefunc.odefunc.diffeq.parameters()])
    pp = torch.cat([p.detach().reshape(-1) for p in mp.chain[0].odefunc.odefunc.diffeq.parameters()]).cpu()
    assert (po - pp).abs().max().item() < 5e-5            # five Adam steps of 1e-3 each moved the weights by ~5e-3


@pytest.mark.gpu
def test_rows_by_global_index_do_not_depend_on_the_sharding():
    """cnf_gu_sample_rows (include/cnf_hip.h): the generator addressed by GLOBAL row index.  Shards of a data set hold the bits
    of the unsharded draw for every split (config 3 evaluates the same 2^24 rows at 1, 2, 4 and 8 GPUs); call c of n rows
    equals the c-th cnf_gu_sample call; the one-component form is a normal sample; and the sharded mean log-likelihood of
    the same rows, evaluated as one shard or as two, agrees to 1e-6 relative (tile-local step control is per 16-row tile
    and the shards here are tile-aligned, so only the summation order differs)."""
    import numpy as np
    from gan_flow_amd import gu_ffjord, parallel_eval
    from gan_flow_amd.gu import DeviceNormalRows, make_gu
    n = 100003
    for gu_num, ext in ((8, True), (1, False)):
        smp = make_gu(gu_num, device="cpu", seed=11, extend_dim=ext).device_sampler(DEV)
        full = smp.get_rows(0, n, call=3)
        for world in (2, 3, 8):
            parts = [smp.get_rows(*parallel_eval.shard_bounds(n, r, world), call=3) for r in range(world)]
            assert torch.equal(torch.cat(parts), full)
        first, second = smp.get_sample(4096), smp.get_sample(4096)          # the stateful entry point: calls 0 and 1
        assert torch.equal(first, smp.get_rows(0, 4096, call=0)) and torch.equal(second, smp.get_rows(0, 4096, call=1))
        assert smp.get_rows(5, 5).shape[0] == 0
    z = DeviceNormalRows(DEV, seed=7).get_rows(0, 400000, dim=2)
    assert z.shape == (400000, 2) and abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3
    assert abs(torch.corrcoef(z.T)[0, 1].item()) < 5e-3
    t = DeviceNormalRows(DEV, seed=7, mean=45.0, sigma=23.0).get_rows(1000, 201000)
    assert abs(t.mean().item() - 45.0) < 0.2 and abs(t.std().item() - 23.0) < 0.2
    # one shard vs two shards of the same rows through the trained d = 1 model
    a = np.load(os.path.join(GOLDEN, "ffjord_multimodal.npz"))
    em = gu_ffjord.build_model(1)
    em.load_state_dict({k: torch.from_numpy(a[k]) for k in a.files})
    em = em.to(DEV).eval()
    for b in em.chain:
        b.tile_rows = 16
    mix = make_gu(8, device="cpu", seed=1234, extend_dim=False).device_sampler(DEV)
    rows = 1 << 14
    with torch.no_grad():
        lp = lambda v: parallel_eval.model_log_prob(em, v)
        one, n1 = parallel_eval.sharded_mean_log_likelihood(lp, mix.get_rows(0, rows))
        halves = [lp(mix.get_rows(lo, hi)).double().sum().item() for lo, hi in ((0, rows // 2), (rows // 2, rows))]
    assert n1 == rows and abs(sum(halves) / rows - one) <= 1e-6 * abs(one)


@pytest.mark.gpu
def test_resumed_run_trains_on_fresh_batches(tmp_path, monkeypatch):
    """`gu_maf --resume` with the default device sampler: the captured graph holds the sampler's seed by value, so the resume
    path must move the seed BEFORE it rebuilds the fused step -- else the resumed run replays the batches the original run
    had already trained on.  The batches a run trains on are recorded off FusedTrainStep's static input."""
    from gan_flow_amd import gu_ffjord
    seen = []
    orig = gu_ffjord.FusedTrainStep.__call__

    def recording(self, x=None):
        out = orig(self, x)
        seen.append(self.static_x[:8].clone())
        return out

    monkeypatch.setattr(gu_ffjord.FusedTrainStep, "__call__", recording)
    common = ["--gu_num", "8", "--batch_size", "512", "--log_interval", "1000", "--eval_size", "2000", "--eval_logp_size", "512",
              "--save", str(tmp_path)]
    gu_ffjord.main(common + ["--niters", "3"])
    first = [t.cpu() for t in seen]
    del seen[:]
    gu_ffjord.main(common + ["--niters", "6", "--resume", str(tmp_path / "ffjord_checkpoint.pth")])
    resumed = [t.cpu() for t in seen]
    assert len(first) == 3 and len(resumed) == 3
    assert all(not torch.equal(a, b) for a in first for b in resumed)
    assert all(not torch.equal(resumed[i], resumed[j]) for i in range(3) for j in range(i))


@pytest.mark.gpu
def test_driver_trains_with_kinetic_regularisers(tmp_path):
    """`gu_maf --l1int c1 --l2int c2` (ffjord/train_misc.py's flags): the driver builds the regularised blocks, adds the
    weighted states to the loss and trains through the autograd loop; the loss it logs is the NLL plus the weighted states
    (checked against a second forward of the saved model), and the weights move."""
    from gan_flow_amd import gu_ffjord
    from gan_flow_amd.lib.layers.wrappers import cnf_regularization as creg
    gu_ffjord.main(["--gu_num", "8", "--batch_size", "512", "--log_interval", "1000", "--eval_size", "2000", "--eval_logp_size", "512",
                    "--save", str(tmp_path), "--niters", "4", "--l1int", "0.5", "--l2int", "0.25"])
    sd = torch.load(tmp_path / "ffjord_state_dict.pth", map_location="cuda:0", weights_only=True)
    torch.manual_seed(1)
    fresh = gu_ffjord.build_model(2, regularization_fns=(creg.l1_regularzation_fn, creg.l2_regularzation_fn))
    moved = max((sd[k].cpu() - v).abs().max().item() for k, v in fresh.state_dict().items())
    assert moved > 1e-4
    model = gu_ffjord.build_model(2, regularization_fns=(creg.l1_regularzation_fn, creg.l2_regularzation_fn)).to("cuda:0")
    model.load_state_dict(sd)
    model.train()
    x = torch.randn(512, 2, device="cuda:0") * 3 + 40
    nll = gu_ffjord.compute_loss(model, x)
    reg = gu_ffjord.get_regularization(model, [0.5, 0.25])
    assert torch.isfinite(nll) and torch.isfinite(reg) and reg.item() > 0
    (nll + reg).backward()
    net = model.chain[0].odefunc.odefunc.diffeq   # (sqrt_end_time never trains: the reference detaches the end time)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in net.parameters())


@pytest.mark.gpu
def test_driver_trains_with_a_fixed_grid_solver_and_test_solver_flags(tmp_path, monkeypatch):
    """`gu_maf --solver rk4 --step_size h` trains through FusedTrainStep (the one-tile-per-workgroup fixed-grid kernels in one
    captured graph) and `--test_solver / --test_atol / --test_rtol` (ffjord/train_misc.py::set_cnf_options) reach the blocks'
    evaluation-mode solves: the logged evaluation runs dopri5 at the test tolerances while training runs rk4."""
    from gan_flow_amd import gu_ffjord
    made = []
    orig = gu_ffjord.FusedTrainStep.__init__

    def recording(self, model, *a, **k):
        made.append(model)
        return orig(self, model, *a, **k)

    monkeypatch.setattr(gu_ffjord.FusedTrainStep, "__init__", recording)
    gu_ffjord.main(["--gu_num", "8", "--batch_size", "512", "--log_interval", "2", "--eval_size", "2000", "--eval_logp_size", "512",
                    "--save", str(tmp_path), "--niters", "4", "--solver", "rk4", "--step_size", "0.1", "--test_solver", "dopri5",
                    "--test_atol", "1e-4", "--test_rtol", "1e-4"])
    assert len(made) == 1                                   # the fused step took the fixed-grid configuration
    blk = made[0].chain[0]
    assert (blk.solver, blk.test_solver, blk.test_atol, blk.test_rtol) == ("rk4", "dopri5", 1e-4, 1e-4)
    assert blk.solver_options == {"step_size": 0.1}
    sd = torch.load(tmp_path / "ffjord_state_dict.pth", map_location="cpu", weights_only=True)
    torch.manual_seed(1)
    fresh = gu_ffjord.build_model(2, solver="rk4")
    assert max((sd[k] - v).abs().max().item() for k, v in fresh.state_dict().items()) > 1e-4
