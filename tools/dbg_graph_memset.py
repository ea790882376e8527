"""Does a captured, unflagged cnf_solve_fwd (two memset nodes + the solve kernel) survive eager solves between its replays?
Library only -- no autograd, no optimizer.  usage: dbg_graph_memset.py [rows of the eager solve]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native as nat, gu_ffjord  # noqa: E402

dev = torch.device("cuda:0")
eager_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev)
fn = model.chain[0].odefunc.odefunc
net, params = fn.net_desc(), fn.diffeq.flat_parameters().detach().clone()
x = torch.randn(2048, 2, device=dev)
xe = torch.randn(eager_rows, 2, device=dev)
opts = nat.make_opts("dopri5", 1e-5, 1e-5)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    nat.solve_fwd(net, params, x, None, 0.0, 0.5, opts)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    z1, l1, st = nat.solve_fwd(net, params, x, None, 0.0, 0.5, opts)
for phase in range(3):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("after replays, phase %d: captured stats %s z1 finite %s" % (phase, st.tolist(), bool(torch.isfinite(z1).all())))
    if phase == 0:
        ze, le, se = nat.solve_fwd(net, params, xe, None, 0.0, 0.5, opts)
        torch.cuda.synchronize()
        print("eager solve of %d rows: stats %s" % (eager_rows, se.tolist()))
    if phase == 1:
        y = torch.sort(torch.randn(100000, device=dev))[0]
        torch.cuda.synchronize()
        print("eager torch work done", float(y[0]))
