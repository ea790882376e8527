#!/usr/bin/env python3
"""Looks at the per-workgroup gradient slabs a two-workgroups-per-tile adjoint launch (CNF_FLAG_SPREAD) leaves in the
workspace: every owned entry against the plain launch's slab of the same tile (bit for bit), and everything a workgroup
does not own against the zero the call's memset left there (a stray store shows up with its address).
usage: dbg_slab_scan.py [rows] [dim]"""
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
H, GRAN = 64, 16448
m = oracle_random_model(dim, seed=9, divergence_fn="brute_force", scale=2.0)
net = nat.net_desc(dim, H, 3, False)
params = torch.cat([p.detach().reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()]).to(dev)
P = params.numel()
x = torch.randn(rows, dim, generator=torch.Generator().manual_seed(31)).to(dev)
z1, l1, _ = nat.solve_fwd(net, params, x, None, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5))
gz1, gl1 = z1 / rows, torch.full((rows, 1), 1.0 / rows, device=dev)
tiles = (rows + 15) // 16
w3 = dim * H + 4 * H + H * H + 4 * H     # first parameter of the upper half (W3)
slabs = {}
for flags, name, n in ((0, "plain", tiles), (nat.FLAG_SPREAD, "spread", 2 * tiles)):
    ws = torch.full((GRAN + 4 * P * 2 * tiles + 4096,), 0x7F, dtype=torch.uint8, device=dev)
    nat.solve_bwd(net, params, z1, l1, gz1, gl1, 0.0, 0.5, nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags), workspace=ws)
    torch.cuda.synchronize()
    slabs[name] = ws[GRAN:GRAN + 4 * P * n].view(torch.float32).view(n, P).cpu()
pl, sp = slabs["plain"], slabs["spread"]
lo, hi = sp[1::2], sp[0::2]          # odd workgroups keep the lower half, even ones the upper half
bad = 0
for name, got, ref, a, b in (("lower half (odd workgroups)", lo, pl, 0, w3), ("upper half (even workgroups)", hi, pl, w3, P)):
    d = got[:, a:b].view(torch.int32) != ref[:, a:b].view(torch.int32)
    print("%s: %d of %d owned entries differ from the plain launch's slab" % (name, int(d.sum()), d.numel()))
    for t, i in torch.nonzero(d)[:24].tolist():
        print("    tile %d entry %d: plain %.6e spread %.6e" % (t, a + i, ref[t, a + i].item(), got[t, a + i].item()))
    bad += int(d.sum())
for name, got, a, b in (("odd workgroups, upper half", lo, w3, P), ("even workgroups, lower half", hi, 0, w3)):
    nz = got[:, a:b] != 0
    print("%s (not owned): %d nonzero entries" % (name, int(nz.sum())))
    for t, i in torch.nonzero(nz)[:24].tolist():
        print("    tile %d entry %d: %.6e (bits %08x)" % (t, a + i, got[t, a + i].item(), got[t, a + i].view(torch.int32).item() & 0xFFFFFFFF))
print("OK" if bad == 0 else "MISMATCH")
