#!/usr/bin/env python3
"""Determinism probe for a build variant of the tile-local dopri5 kernel (tools/build_variants.sh, SRC=cnf_fwd): the
per-tile arithmetic does not depend on how many waves share a SIMD or a workgroup, so any variant must return the bits of
the shipped build, run after run.  usage: CNF_HIP_LIB=.../libcnf_hip_X.so dbg_variant_bits.py out.pt [ref.pt]
Writes (z, logp, nfe-per-call) of a fixed 2^18-row problem (d = 1 and d = 2, trained stiff field) solved three times;
with ref.pt compares against another build's file and prints which rows / tiles differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gan_flow_amd import _native, gu_ffjord  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << 18
out = {}
for dim in (1, 2):
    if dim == 1:
        a = np.load(os.path.join(ROOT, "tests", "golden", "ffjord_multimodal.npz"))
        em = gu_ffjord.build_model(1)
        em.load_state_dict({k: torch.from_numpy(a[k]) for k in a.files})
    else:
        em = bench.embed_trained_multimodal(gu_ffjord.build_model(2))
    em = em.to(dev).eval()
    blk = em.chain[0]
    g = torch.Generator(device=dev).manual_seed(99)
    x = 7 + 76 * torch.rand(n, 1, generator=g, device=dev)
    if dim == 2:
        x = torch.cat([x, torch.rand(n, 1, generator=g, device=dev)], 1)
    from gan_flow_amd.lib.layers.cnf import _inner
    net = _inner(blk.odefunc).net_desc()
    flat = _inner(blk.odefunc).diffeq.flat_parameters().detach()
    opts = _native.make_opts("dopri5", 1e-5, 1e-5, tile_rows=16)
    runs = []
    for rep in range(3):
        z, lp, st = _native.solve_fwd(net, flat, x, None, 0.0, 0.5, opts, None)
        torch.cuda.synchronize()
        s = _native.Stats(st)
        s.raise_for_status()
        runs.append((z.cpu(), lp.cpu(), s.nfe, s.row_evals))
    for rep in (1, 2):
        same = torch.equal(runs[0][0], runs[rep][0]) and torch.equal(runs[0][1], runs[rep][1])
        bad = (runs[0][1] != runs[rep][1]).reshape(-1).nonzero().reshape(-1)
        print("d=%d run %d vs run 0: %s (max nfe %d / %d, row evals %d / %d)%s" % (
            dim, rep, "identical bits" if same else "DIFFERENT", runs[0][2], runs[rep][2], runs[0][3], runs[rep][3],
            "" if same else "; %d rows differ, tiles %s" % (bad.numel(), sorted(set((bad // 16).tolist()))[:12])), flush=True)
    out["z%d" % dim], out["lp%d" % dim] = runs[0][0], runs[0][1]
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k in sorted(out):
        bad = (out[k] != ref[k]).reshape(out[k].shape[0], -1).any(1).nonzero().reshape(-1)
        print("%s vs reference build: %s" % (k, "identical bits" if bad.numel() == 0 else
              "%d rows differ (tiles %s ...), max |diff| %.3e" % (bad.numel(), sorted(set((bad // 16).tolist()))[:8],
                                                                  (out[k] - ref[k]).abs().max().item())), flush=True)
