"""One 16-row tile of BASELINE config 3's points (global rows 16 t .. 16 t + 15, bench.py's generator): the density-direction
solve x -> z and the way back, by the HIP tile-local kernel, the HIP batch-global kernel on the same 16 rows (one tile: the
same control), the fp32 oracle and the float64 oracle at 1e-10 (the truth).  usage: dbg_config3_tile.py TILE [TILE ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from _helpers import oracle_fixture_model  # noqa: E402
from gan_flow_amd import _native as nat  # noqa: E402
from gan_flow_amd.gu import DeviceNormalRows, make_gu  # noqa: E402

DEV = "cuda:0"
N = 1 << 24
m = oracle_fixture_model("multimodal")
m64 = oracle_fixture_model("multimodal", torch.float64)
flat = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).float().to(DEV)
net = nat.net_desc(1, 64, 3, False)
T = float(m.chain[0].sqrt_end_time.detach() ** 2)
mix = make_gu(8, device="cpu", seed=1234, extend_dim=False).device_sampler(DEV)
tails = DeviceNormalRows(DEV, seed=4321, mean=45.0, sigma=23.0)


def points(lo, hi):
    h = N // 2
    return torch.cat([mix.get_rows(min(lo, h), min(hi, h)), tails.get_rows(max(lo, h) - h, max(hi, h) - h)])


for t in [int(v) for v in sys.argv[1:]]:
    x = points(16 * t, 16 * t + 16)
    xc = x.cpu()
    print("tile %d: x = %s" % (t, " ".join("%.3f" % v for v in xc.reshape(-1).tolist())))
    with torch.no_grad():
        z64, l64 = oracle.odeint(m64.chain[0].odefunc, (xc.double(), torch.zeros(16, 1, dtype=torch.float64)),
                                 torch.tensor([0.0, T], dtype=torch.float64), rtol=1e-10, atol=1e-10, method="dopri5")
        st = {}
        z32, l32 = oracle.odeint(m.chain[0].odefunc, (xc.clone(), torch.zeros(16, 1)), torch.tensor([0.0, T]), rtol=[1e-5] * 2,
                                 atol=[1e-5] * 2, method="dopri5", options=dict(stats=st))
    z64, l64, z32, l32 = z64[1], l64[1], z32[1], l32[1]
    print("  oracle fp32: %d accepted %d rejected | z vs f64 max %.3e (per row: %s)"
          % (st["n_accepted"], st["n_rejected"], (z32.double() - z64).abs().max(), " ".join("%.1e" % v for v in (z32.double() - z64).abs().reshape(-1).tolist())))
    for name, tile, flags in (("hip tile-local", 16, 0), ("hip batch-global (twelve waves)", 0, 0), ("hip batch-global (four waves)", 0, nat.FLAG_FWD_FOUR_WAVES)):
        o = nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=tile, flags=flags)
        z, l, s = nat.solve_fwd(net, flat, x, None, 0.0, T, o)
        xb, lb, s2 = nat.solve_fwd(net, flat, z, l, T, 0.0, o)
        torch.cuda.synchronize()
        s, s2 = nat.Stats(s), nat.Stats(s2)
        dz = (z.cpu().double() - z64).abs().reshape(-1)
        print("  %-34s %d accepted %d rejected | z vs f64 max %.3e logp %.3e (per row: %s) | way back: %d + %d, |x' - x| max %.3e"
              % (name, s.n_accepted, s.n_rejected, dz.max(), (l.cpu().double() - l64).abs().max(), " ".join("%.1e" % v for v in dz.tolist()),
                 s2.n_accepted, s2.n_rejected, (xb - x).abs().max()))
