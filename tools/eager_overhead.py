#!/usr/bin/env python3
"""Where the eager (autograd + torch.optim.Adam) training loop spends its time: variants of the loop at B=2048."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_flow_amd import gu_ffjord
from gan_flow_amd.gu import make_gu
dev = torch.device("cuda:0")
data = make_gu(8, seed=1)
xs = [data.get_sample(2048).to(dev) for _ in range(60)]
def run(name, fused_adam, sync, flatten):
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2).to(dev).train()
    if flatten:
        model.chain[0].odefunc.odefunc.diffeq.flatten_parameters()
    for b in model.chain:
        b.sync_stats = sync
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=fused_adam)
    def step(x):
        opt.zero_grad(set_to_none=True)
        loss = gu_ffjord.compute_loss(model, x)
        loss.backward()
        opt.step()
    for x in xs[:10]:
        step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in xs[10:]:
        step(x)
    torch.cuda.synchronize()
    print("%-40s %.3f ms/iter" % (name, 1e3 * (time.perf_counter() - t0) / 50), flush=True)
run("foreach Adam, sync stats", False, True, False)
run("fused Adam, sync stats", True, True, False)
run("fused Adam, no stats sync", True, False, False)
run("fused Adam, no stats sync, flattened", True, False, True)
import cProfile, pstats
torch.manual_seed(1)
model = gu_ffjord.build_model(2).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
def step(x):
    opt.zero_grad(set_to_none=True)
    loss = gu_ffjord.compute_loss(model, x)
    loss.backward()
    opt.step()
for x in xs[:10]:
    step(x)
pr = cProfile.Profile(); pr.enable()
for x in xs[10:]:
    step(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
