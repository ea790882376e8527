#!/usr/bin/env python3
"""Fixed cost of a solve kernel launch: kernel time (HIP events inside the library) of the forward and adjoint solves at
NFE 8 (max_steps = 1: one attempt, then the step limit ends the solve) and NFE 14 (the full early-training solve).  The
difference is six evaluations + one grid reduction; extrapolating to zero evaluations gives what a launch costs before and
after its integration loop (staging the weights, initial step, output, cleanup)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

dev = torch.device("cuda:0")
B = 2048
x = make_gu(8, seed=1).get_sample(B).to(dev)
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev)
fn = model.chain[0].odefunc.odefunc
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
full = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_SPREAD)
z1, l1, _ = nat.solve_fwd(net, params, x, None, 0.0, 0.5, full)
gz, gl = z1 / B, torch.full_like(l1, 1.0 / B)
nat.kernel_timing(True)
for name, ms in (("nfe8 ", 1), ("nfe14", 0)):
    o = nat.make_opts("dopri5", 1e-5, 1e-5, max_steps=ms, flags=nat.FLAG_SPREAD)
    tf = tb = 0.0
    reps = 20
    for _ in range(reps):
        _, _, sf = nat.solve_fwd(net, params, x, None, 0.0, 0.5, o)
        tf += nat.last_kernel_ms() / reps
        _, _, _, sb = nat.solve_bwd(net, params, z1, l1, gz, gl, 0.0, 0.5, o)
        tb += nat.last_kernel_ms() / reps
    sf, sb = nat.Stats(sf), nat.Stats(sb)
    print("%s fwd kernel %.1f us (nfe %d) | bwd kernel %.1f us (nfe %d)" % (name, 1e3 * tf, sf.nfe, 1e3 * tb, sb.nfe))
