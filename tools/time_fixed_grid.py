#!/usr/bin/env python3
"""Fixed-grid solves (rk4 / midpoint / euler) at the training batch size: kernel time (HIP events inside the library) of the
forward and the adjoint solve on the one-tile-per-workgroup kernels (fixed_split_tw_kernel, fixed_bwd_split_kernel: four
waves per tile, and CNF_FLAG_SPREAD's two eight-wave workgroups per tile) against the one-wave-per-tile kernels
(CNF_FLAG_FWD_FOUR_WAVES keeps them for batches of up to 4096 rows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev)
fn = model.chain[0].odefunc.odefunc
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
nat.kernel_timing(True)
for B in (2048, 4096, 256):
    x = make_gu(8, seed=1).get_sample(B).to(dev)
    for solver, step in (("rk4", 0.05), ("midpoint", 0.05), ("euler", 0.05)):
        for name, flags in (("one tile per workgroup", 0), ("  + spread adjoint", nat.FLAG_SPREAD), ("one wave per tile", nat.FLAG_FWD_FOUR_WAVES)):
            o = nat.make_opts(solver, step_size=step, flags=flags)
            tf, tb = [], []
            for _ in range(8):
                z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, 0.5, o)
                tf.append(nat.last_kernel_ms())
                _, _, _, sb = nat.solve_bwd(net, params, z1, l1, z1 / B, torch.full_like(l1, 1.0 / B), 0.0, 0.5, o)
                tb.append(nat.last_kernel_ms())
            sf, sb = nat.Stats(sf), nat.Stats(sb)
            print("B=%-5d %-8s %-24s fwd %7.1f us (%.2f us / evaluation, nfe %d) | adjoint %7.1f us (%.2f us / evaluation, nfe %d)"
                  % (B, solver, name, 1e3 * min(tf), 1e3 * min(tf) / sf.nfe, sf.nfe, 1e3 * min(tb), 1e3 * min(tb) / sb.nfe, sb.nfe), flush=True)
