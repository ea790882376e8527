"""Which fp32 dopri5 run is wrong where two of them disagree?  The first ROWS of BASELINE config 3's points through (a) the
tile-local kernel in one launch and (b) the batch-global one-tile-per-workgroup kernel launched once per 16-row tile (one tile
= the same control group, other arithmetic and summation order); tiles where a row differs by more than 1e-3 in z are solved
by the float64 oracle at 1e-10 (the truth) and by the fp32 oracle.  usage: dbg_tile_outliers.py [ROWS]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from _helpers import oracle_fixture_model  # noqa: E402
from gan_flow_amd import _native as nat  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

DEV = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = oracle_fixture_model("multimodal")
m64 = oracle_fixture_model("multimodal", torch.float64)
flat = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).float().to(DEV)
net = nat.net_desc(1, 64, 3, False)
T = float(m.chain[0].sqrt_end_time.detach() ** 2)
x = make_gu(8, device="cpu", seed=1234, extend_dim=False).device_sampler(DEV).get_rows(0, rows)
za, la, sa = nat.solve_fwd(net, flat, x, None, 0.0, T, nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16))
torch.cuda.synchronize()
nat.Stats(sa).raise_for_status()
og = nat.make_opts("dopri5", 1e-5, 1e-5)
zb = torch.empty_like(za)
lb = torch.empty_like(la)
for t in range(rows // 16):
    z, l, s = nat.solve_fwd(net, flat, x[16 * t:16 * t + 16].contiguous(), None, 0.0, T, og)
    zb[16 * t:16 * t + 16] = z
    lb[16 * t:16 * t + 16] = l
torch.cuda.synchronize()
dz = (za - zb).abs().reshape(-1)
print("%d rows, %d tiles: |z_tile_local - z_batch_global| 50 / 99 / 99.9 %% / max: %s %.2e; rows over 1e-3: %d, over 1e-2: %d"
      % (rows, rows // 16, " ".join("%.1e" % v for v in torch.quantile(dz, torch.tensor([0.5, 0.99, 0.999], device=DEV)).tolist()),
         dz.max().item(), int((dz > 1e-3).sum()), int((dz > 1e-2).sum())), flush=True)
bad = torch.unique(torch.nonzero(dz > 1e-3).reshape(-1) // 16).tolist()
wrong = {"tile-local": 0, "batch-global": 0, "oracle fp32": 0}
for t in bad[:40]:
    xc = x[16 * t:16 * t + 16].cpu()
    with torch.no_grad():
        z64, l64 = oracle.odeint(m64.chain[0].odefunc, (xc.double(), torch.zeros(16, 1, dtype=torch.float64)),
                                 torch.tensor([0.0, T], dtype=torch.float64), rtol=1e-10, atol=1e-10, method="dopri5")
        z32, l32 = oracle.odeint(m.chain[0].odefunc, (xc.clone(), torch.zeros(16, 1)), torch.tensor([0.0, T]), rtol=[1e-5] * 2,
                                 atol=[1e-5] * 2, method="dopri5")
    ea = (za[16 * t:16 * t + 16].cpu().double() - z64[1]).abs().reshape(-1)
    eb = (zb[16 * t:16 * t + 16].cpu().double() - z64[1]).abs().reshape(-1)
    eo = (z32[1].double() - z64[1]).abs().reshape(-1)
    i = int(torch.stack([ea, eb, eo]).max(0).values.argmax())
    wrong["tile-local"] += int(ea.max() > 1e-3)
    wrong["batch-global"] += int(eb.max() > 1e-3)
    wrong["oracle fp32"] += int(eo.max() > 1e-3)
    print("  tile %6d row %2d x = %8.4f: |z - z_f64| tile-local %.2e batch-global %.2e oracle fp32 %.2e" % (t, i, xc[i].item(), ea[i], eb[i], eo[i]), flush=True)
print("tiles (of %d examined) with a row more than 1e-3 from the float64 solve:" % min(len(bad), 40), wrong)
