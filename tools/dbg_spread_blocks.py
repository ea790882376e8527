#!/usr/bin/env python3
"""Compares the gradients of a two-workgroups-per-tile adjoint launch (CNF_FLAG_SPREAD) with those of the plain launch,
parameter block by parameter block (they must be bit-identical), with and without CNF_FLAG_ADJ_CONTROL_ALL.
usage: dbg_spread_blocks.py [rows] [dim]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _helpers import oracle_random_model  # noqa: E402
from gan_flow_amd import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H = 64
g = torch.Generator().manual_seed(31)
m = oracle_random_model(dim, seed=9, divergence_fn="brute_force", scale=2.0)
net = nat.net_desc(dim, H, 3, False)
params = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).to(dev)
x = torch.randn(rows, dim, generator=g).to(dev)
z1, l1, _ = nat.solve_fwd(net, params, x, None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5))
gz1, gl1 = z1 / rows, torch.full((rows, 1), 1.0 / rows, device=dev)
# canonical parameter order: per layer W (out, in), b, hyper_bias W, hyper_gate W, hyper_gate b
dims = [dim, H, H, H, dim]
blocks, o = [], 0
for l in range(4):
    for name, n in (("W", dims[l] * dims[l + 1]), ("b", dims[l + 1]), ("hb", dims[l + 1]), ("gw", dims[l + 1]), ("gb", dims[l + 1])):
        blocks.append(("L%d.%s" % (l + 1, name), o, o + n))
        o += n
assert o == params.numel(), (o, params.numel())
ref_path = os.environ.get("DBG_REF")      # gradients saved by another build (DBG_SAVE) to compare the two launches with
saved = {}
for extra, label in ((0, "default"), (nat.FLAG_ADJ_CONTROL_ALL, "control_all")):
    res = []
    for flags in (0, nat.FLAG_SPREAD):
        gx, gl0, gp, st = nat.solve_bwd(net, params, z1, l1, gz1, gl1, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags | extra))
        torch.cuda.synchronize()
        st = nat.Stats(st)
        res.append((gx.clone(), gp.clone(), (st.status, st.nfe, st.n_accepted, st.n_rejected)))
    saved[label] = (res[0][1].cpu(), res[1][1].cpu())
    if ref_path:
        ref = torch.load(ref_path)[label]
        for which, k in (("plain", 0), ("spread", 1)):
            for kr, wr in ((0, "plain"), (1, "spread")):
                d = (ref[kr] - saved[label][k]).abs()
                print("  %s: this build's %s vs the reference build's %s: entries differing %d, max %.3e" % (label, which, wr, int((d > 0).sum()), d.max().item()))
    print(label, "stats plain", res[0][2], "spread", res[1][2], "grad_x equal", torch.equal(res[0][0], res[1][0]))
    for name, a, b in blocks:
        d = (res[0][1][a:b] - res[1][1][a:b]).abs()
        print("  %-6s max |plain| %.3e  max diff %.3e  entries differing %d / %d" % (name, res[0][1][a:b].abs().max().item(), d.max().item(),
                                                                                     int((d > 0).sum().item()), b - a))
        if 0 < int((d > 0).sum().item()) <= 16:
            for i in torch.nonzero(d > 0).flatten().tolist():
                print("      entry %d (row %d, col %d of %s): plain %.6e spread %.6e" % (i, i // dims[int(name[1]) - 1] if name.endswith("W") else i,
                      i % dims[int(name[1]) - 1] if name.endswith("W") else 0, name, res[0][1][a + i].item(), res[1][1][a + i].item()))
if os.environ.get("DBG_SAVE"):
    torch.save(saved, os.environ["DBG_SAVE"])
