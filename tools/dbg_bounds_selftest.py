#!/usr/bin/env python3
"""Self-test of the bounds-checking build (make -C gan-flow_amd/csrc debug; run with CNF_HIP_LIBDIR=gan-flow_amd/debug).
Pass 1: a forward + adjoint solve at hidden width 64 and 256 must come back clean.  Pass 2 (CNF_DEBUG_SHRINK_WS set by this
script's caller): the same calls with the checker told that the workspace ends early must come back with status 5
(CNF_ERR_DEBUG_BOUNDS) from the adjoint -- whose last workgroups' gradient slabs lie in the bytes cut off -- and print the
ranges; nothing faults, because the memory is really there."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402

dev = torch.device("cuda:0")
expect_report = bool(os.environ.get("CNF_DEBUG_SHRINK_WS"))
print("library:", nat.library_path(), "| expecting a report:", expect_report, flush=True)
ok = True
for hidden in (64, 256):
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2, dims=(hidden,) * 3).to(dev)
    fn = model.chain[0].odefunc.odefunc
    net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach()
    x = torch.randn(2048, 2, device=dev)
    opts = nat.make_opts("dopri5", 1e-5, 1e-5, flags=nat.FLAG_SPREAD)
    z1, l1, sf = nat.solve_fwd(net, params, x, None, 0.0, 0.5, opts)
    out = nat.solve_bwd(net, params, z1, l1, z1 / 2048, torch.full_like(l1, 1.0 / 2048), 0.0, 0.5, opts)
    torch.cuda.synchronize()
    sf, sb = nat.Stats(sf), nat.Stats(out[-1])
    print("hidden %d: forward status %d, adjoint status %d" % (hidden, sf.status, sb.status), flush=True)
    ok &= (sb.status == 5) if expect_report else (sf.status == 0 and sb.status == 0)
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
