#!/usr/bin/env python3
"""Kernel-only times (cnf_debug_kernel_timing) of the d=16 forward and adjoint solves, plain and CNF_FLAG_SPREAD, on the
config-5 problem (random init, B rows of the synthetic Gaussian mixture)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402
from gan_flow_amd.gmm import GaussianMixture  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = GaussianMixture(device=dev).get_sample(B)
torch.manual_seed(1)
model = gu_ffjord.build_model(16, divergence_fn="approximate").to(dev).train()
fn = model.chain[0].odefunc.odefunc
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
T = model.chain[0]._end_time()
e = torch.randn_like(x)
plain = nat.make_opts("dopri5", 1e-5, 1e-5)
z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, T, plain, e)
gz, gl = z1 / B, torch.full_like(l1, 1.0 / B)
nat.kernel_timing(True)
for name, flags in (("plain", 0), ("spread", nat.FLAG_SPREAD)):
    o = nat.make_opts("dopri5", 1e-5, 1e-5, flags=flags)
    ms = 0.0
    for _ in range(10):
        sb = nat.solve_bwd(net, params, z1, l1, gz, gl, 0.0, T, o, e)[3]
        ms += nat.last_kernel_ms() / 10
    sb = nat.Stats(sb)
    print("d16 bwd %-6s %8.2f us  nfe %d run %d  %6.2f us/eval-run" % (name, 1e3 * ms, sb.nfe, sb.evals_run, 1e3 * ms / sb.evals_run))
ms = 0.0
for _ in range(10):
    sf = nat.solve_fwd(net, params, x, None, 0.0, T, plain, e)[2]
    ms += nat.last_kernel_ms() / 10
sf = nat.Stats(sf)
print("d16 fwd        %8.2f us  nfe %d  %6.2f us/eval" % (1e3 * ms, sf.nfe, 1e3 * ms / sf.nfe))
