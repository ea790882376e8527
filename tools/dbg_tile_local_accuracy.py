#!/usr/bin/env python3
"""Accuracy of the tile-local density evaluation on the trained fixtures against a float64 tight-tolerance solve
(oracle, rtol = atol = 1e-10), for whichever library CNF_HIP_LIB names: usage dbg_tile_local_accuracy.py [rows]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from _helpers import oracle_fixture_model  # noqa: E402
from gan_flow_amd import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.set_num_threads(16)
for which, lo, hi in (("unimodal", 4.5, 5.5), ("multimodal", 7.0, 83.0)):
    m = oracle_fixture_model(which)
    flat = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()])
    g = torch.Generator().manual_seed(5)
    x = torch.rand(rows, 1, generator=g) * (hi - lo) + lo
    blk = m.chain[0]
    T = float(blk.sqrt_end_time.detach() * blk.sqrt_end_time.detach())
    z, l, st = nat.solve_fwd(nat.net_desc(1, 64, 3, False), flat.to(dev), x.to(dev), None, 0.0, T,
                             nat.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16))
    torch.cuda.synchronize()
    st = nat.Stats(st)
    m64 = oracle_fixture_model(which, torch.float64)
    ts = torch.tensor([0.0, T], dtype=torch.float64)
    with torch.no_grad():
        z64, l64 = oracle.odeint(m64.chain[0].odefunc, (x.double(), torch.zeros(rows, 1, dtype=torch.float64)), ts,
                                 rtol=1e-10, atol=1e-10, method="dopri5")
    ez = (z.cpu().double() - z64[1]).abs().reshape(-1)
    el = (l.cpu().double() - l64[1]).abs().reshape(-1)
    print("%-10s rows %d nfe max %d row_evals %d | z err mean %.3e p99 %.3e max %.3e | logp err mean %.3e p99 %.3e max %.3e | mean logp %.6f (f64 %.6f)"
          % (which, rows, st.nfe, st.row_evals, ez.mean(), ez.quantile(0.99), ez.max(), el.mean(), el.quantile(0.99), el.max(),
             -l.double().mean().item(), -l64[1].mean().item()), flush=True)
