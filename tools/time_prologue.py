#!/usr/bin/env python3
"""Fixed cost of a solve launch: the forward and adjoint split kernels launched with CnfSolverOpts.after pointing at a
failed block exit right after their staging (no evaluation): the kernel time left is prologue + epilogue."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord
from gan_flow_amd.gu import make_gu
dev = torch.device("cuda:0")
B = 2048
x = make_gu(8, seed=1).get_sample(B).to(dev)
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev)
fn = model.chain[0].odefunc.odefunc
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
bad = torch.zeros(8, dtype=torch.int32, device=dev); bad[0] = 3
nat.load_library().cnf_debug_kernel_timing(1)
import ctypes
def kernel_ms():
    ms = ctypes.c_float()
    nat.load_library().cnf_debug_last_kernel_ms(ctypes.byref(ms))
    return ms.value
for name, after in (("full", None), ("staging only", bad)):
    for spread in (0, nat.FLAG_SPREAD):
        opts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=spread, after=after)
        tf, tb = [], []
        for rep in range(6):
            z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, 0.5, opts)
            torch.cuda.synchronize(); tf.append(kernel_ms())
            if after is not None:
                z1, l1 = x.clone(), torch.zeros(B, 1, device=dev)
            nat.solve_bwd(net, params, z1, l1, z1 / B, torch.full_like(l1, 1.0 / B), 0.0, 0.5, opts)
            torch.cuda.synchronize(); tb.append(kernel_ms())
        print("%-13s spread=%d: fwd kernel %.1f us, bwd kernel %.1f us (min of 6)" % (name, bool(spread), 1e3 * min(tf), 1e3 * min(tb)), flush=True)
