#!/usr/bin/env python3
"""Times the forward and adjoint solve kernels alone (HIP events around the C-ABI calls) on two fixed problems:
'early' = seed-1 random init (NFE ~14) and 'late' = the reference's trained gu_num=8 field embedded in d=2 (NFE ~1.7k),
both at B rows of gu_num=8 data.  Prints microseconds per evaluation -- the number kernel work tries to lower."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
SPREAD = (sys.argv[2] if len(sys.argv) > 2 else "spread") == "spread"   # adjoint launched as the training step launches it
reps = 5
x = make_gu(8, seed=1).get_sample(B).to(dev)
for name in ("early", "late"):
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2)
    if name == "late":
        model = bench.embed_trained_multimodal(model)
    model = model.to(dev)
    fn = model.chain[0].odefunc.odefunc
    net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
    T = model.chain[0]._end_time()
    opts = nat.make_opts("dopri5", 1e-5, 1e-5)
    z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, T, opts)
    gz, gl = z1 / B, torch.full_like(l1, 1.0 / B)
    if SPREAD:
        opts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_SPREAD)
    nat.solve_bwd(net, params, z1, l1, gz, gl, 0.0, T, opts)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(reps):
        ev[0].record()
        z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, T, opts)
        ev[1].record()
        _, _, _, sb = nat.solve_bwd(net, params, z1, l1, gz, gl, 0.0, T, opts)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    sf, sb = nat.Stats(sf), nat.Stats(sb)
    print("%-5s B=%d fwd %8.3f ms  nfe %5d  %6.2f us/eval | bwd %8.3f ms  nfe %5d run %5d  %6.2f us/eval-run | %s %s"
          % (name, B, tf / reps, sf.nfe, 1e3 * tf / reps / sf.evals_run, tb / reps, sb.nfe, sb.evals_run,
             1e3 * tb / reps / sb.evals_run, (sf.n_accepted, sf.n_rejected), (sb.n_accepted, sb.n_rejected)))
