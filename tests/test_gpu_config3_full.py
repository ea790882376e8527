"""BASELINE config 3 at its FULL size (2^24 points), through size-independent properties (no oracle can run this size):
sharding invariance of the tile-local evaluation -- every row's log p is the same bits whether one rank evaluates all rows
or eight ranks an eighth each, so the all-reduced mean is world-size invariant to summation order -- and the encode ->
decode round trip x -> z -> x' with its residual distribution (percentiles bounded; ~20 s on an MI355X)."""
import os

import numpy as np
import pytest
import torch

from _helpers import oracle_fixture_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1 << 24
LAUNCH = 1 << 22


def _model():
    from gan_flow_amd import gu_ffjord
    a = np.load(os.path.join(ROOT, "tests", "golden", "ffjord_multimodal.npz"))
    em = gu_ffjord.build_model(1)
    em.load_state_dict({k: torch.from_numpy(a[k]) for k in a.files})
    em = em.to(DEV).eval()
    for b in em.chain:
        b.tile_rows = 16
    return em


def _lp(z, dl):
    """parallel_eval.model_log_prob's expression on a solve's outputs"""
    import math
    return (-0.5 * math.log(2 * math.pi) - z.pow(2) / 2).reshape(z.shape[0], -1).sum(1, keepdim=True) - dl


def _points(lo, hi):
    """bench.py's config-3 points by GLOBAL row: rows [0, N/2) the training mixture, [N/2, N) N(45, 23^2)."""
    from gan_flow_amd.gu import DeviceNormalRows, make_gu
    mix = make_gu(8, device="cpu", seed=1234, extend_dim=False).device_sampler(DEV)
    tails = DeviceNormalRows(DEV, seed=4321, mean=45.0, sigma=23.0)
    h = N // 2
    return torch.cat([mix.get_rows(min(lo, h), min(hi, h)), tails.get_rows(max(lo, h) - h, max(hi, h) - h)])


def test_config3_full_size_sharding_invariance_and_round_trip():
    from gan_flow_amd import parallel_eval
    em = _model()
    with torch.no_grad():
        xs = _points(0, N)
        assert xs.shape == (N, 1)
        # (i) one rank: z and delta log p of all 2^24 rows
        zs, dls = [], []
        for vc in torch.split(xs, LAUNCH):
            z, dl = em(vc, torch.zeros(vc.shape[0], 1, device=DEV))
            zs.append(z)
            dls.append(dl)
        z1, dl1 = torch.cat(zs), torch.cat(dls)
        lp1 = _lp(z1, dl1)
        mean1 = lp1.double().mean().item()
        assert torch.isfinite(lp1).all()
        # (ii) eight ranks' shards, evaluated one after the other from their own draws of the same global rows: every row's
        # log p is the N = 1 value bit for bit (tile-local control, shard bounds are multiples of 16 rows), so the all-reduced
        # mean differs by summation order only
        tot, cnt = 0.0, 0
        for r in range(8):
            lo, hi = parallel_eval.shard_bounds(N, r, 8)
            xr = _points(lo, hi)
            assert torch.equal(xr, xs[lo:hi])
            lpr = parallel_eval.model_log_prob(em, xr, chunk=LAUNCH)   # what a rank of the sharded evaluation computes
            assert torch.equal(lpr.reshape(-1), lp1[lo:hi].reshape(-1)), r
            tot += lpr.double().sum().item()
            cnt += hi - lo
        assert cnt == N and abs(tot / N - mean1) <= 1e-10 * abs(mean1)
        print("config 3 full size: mean log-likelihood %.6f (N = 1) / %.6f (8 shards)" % (mean1, tot / N))
        # (iii) decode: z -> x' with the accumulated delta log p carried back to ~0
        res_x, res_l = [], []
        for zc, dc, xc in zip(torch.split(z1, LAUNCH), torch.split(dl1, LAUNCH), torch.split(xs, LAUNCH)):
            xb, lb = em(zc, dc, reverse=True)
            res_x.append((xb - xc).abs().reshape(-1))
            res_l.append(lb.abs().reshape(-1))
        ex, el = torch.cat(res_x), torch.cat(res_l)
        qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], device=DEV)
        # torch.quantile is limited to 16M elements: a strided subsample of 2^22 rows for the percentiles, the maxima over all
        qx, ql = torch.quantile(ex[::4].float(), qs), torch.quantile(el[::4].float(), qs)
        print("   round trip |x' - x| percentiles (50, 90, 99, 99.9, 99.99, max): %s %.2e" % (" ".join("%.2e" % v for v in qx.tolist()), ex.max().item()))
        print("   round trip |delta log p| percentiles: %s %.2e" % (" ".join("%.2e" % v for v in ql.tolist()), el.max().item()))
        # measured: x 1.8e-4 / 1.1e-3 / 5.7e-3 / 2.5e-2, log p 8.1e-4 / 4.4e-3 / 2.7e-2 / 6.3e-2 (dopri5 at 1e-5 in fp32 on a field
        # with near-jumps: test_round_trip_large_batch_property holds the same percentiles to the oracle's at 16 384 rows)
        for q, bx, bl in zip(range(4), (6e-4, 3e-3, 2e-2, 8e-2), (2.5e-3, 1.5e-2, 8e-2, 0.2)):
            assert qx[q].item() < bx and ql[q].item() < bl, (q, qx.tolist(), ql.tolist())
        # The worst rows (here 0.9 .. 5 in x) are NOT held to a bound: they sit at the edges of the mixture's uniform boxes, where
        # the trained density has near-jumps and dopri5's embedded error estimate (an RMS over the tile's 16 rows) lets a rare
        # inaccurate step through.  tools/dbg_tile_outliers.py (profiles/r03g_tile_outliers.txt) shows this is a property of
        # dopri5(1e-5) in fp32 on this field, not of a kernel: on 4096 tiles the tile-local kernel, the batch-global kernel run
        # per tile and the fp32 oracle are each more than 1e-3 from the float64 solve on their own ~0.5 % of the tiles, and
        # right where another one is off.  For the record, the worst tile here against the oracle on the same 16 rows:
        t = int(torch.argmax(ex).item()) // 16
        rows = slice(16 * t, 16 * t + 16)
        m = oracle_fixture_model("multimodal")
        xo = xs[rows].cpu()
        zo, dlo = m(xo.clone(), torch.zeros(16, 1))
        xb, lb = m(zo, dlo, reverse=True)
        print("   worst tile %d (x in [%.1f, %.1f]): round-trip residual hip %.3e, fp32 oracle on the same rows %.3e"
              % (t, xo.min().item(), xo.max().item(), ex[rows].max().item(), (xb - xo).abs().max().item()))
        # what IS bounded: how many rows are affected at all (measured: 2.5e-2 at the 99.9th percentile, 5.7e-2 at the 99.99th)
        assert qx[4].item() < 0.2 and ql[4].item() < 0.3, (qx.tolist(), ql.tolist())
