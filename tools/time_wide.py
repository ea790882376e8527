#!/usr/bin/env python3
"""Times the hidden-width-256 forward and adjoint solve kernels (library event bracket) at B rows, random init."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord
from gan_flow_amd.gu import make_gu
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
x = make_gu(8, seed=1).get_sample(B).to(dev)
torch.manual_seed(1)
model = gu_ffjord.build_model(2, dims=(256, 256, 256)).to(dev)
fn = model.chain[0].odefunc.odefunc
with torch.no_grad():
    for p in fn.diffeq.parameters():
        if p.dim() == 2 and p.shape[1] > 1:
            p.mul_(scale)
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
T = model.chain[0]._end_time()
opts = nat.make_opts("dopri5", 1e-5, 1e-5)
SPREAD = (sys.argv[3] if len(sys.argv) > 3 else "spread") == "spread"   # adjoint launched as the training step launches it
bopts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_SPREAD) if SPREAD else opts
nat.kernel_timing(True)
for rep in range(3):
    z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, T, opts)
    tf = nat.last_kernel_ms()
    _, _, _, sb = nat.solve_bwd(net, params, z1, l1, z1 / B, torch.full_like(l1, 1.0 / B), 0.0, T, bopts)
    tb = nat.last_kernel_ms()
sf, sb = nat.Stats(sf), nat.Stats(sb)
F = 790528
print("H=256 B=%d scale %.1f | fwd %.3f ms nfe %d  %.1f us/eval  %.1f TFLOP/s | bwd %.3f ms nfe %d steps %d+%d  %.1f us/eval  %.1f TFLOP/s"
      % (B, scale, tf, sf.nfe, 1e3 * tf / sf.nfe, B * sf.nfe * F / tf / 1e9, tb, sb.nfe, sb.n_accepted, sb.n_rejected,
         1e3 * tb / sb.evals_run, 3 * B * sb.nfe * F / tb / 1e9))

# fixed-grid adjoint (rk4, 5 steps) at the same width
ro = nat.make_opts("rk4", 1e-5, 1e-5, 0.1)
z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, T, ro)
for rep in range(3):
    _, _, _, sb = nat.solve_bwd(net, params, z1, l1, z1 / B, torch.full_like(l1, 1.0 / B), 0.0, T, ro)
    tb = nat.last_kernel_ms()
sb = nat.Stats(sb)
print("H=256 rk4 adjoint: %.3f ms, %d evaluations, %.1f us/eval" % (tb, sb.evals_run, 1e3 * tb / sb.evals_run))
