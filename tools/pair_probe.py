#!/usr/bin/env python3
"""Does a kernel with a large private segment (register spills) cost later stream concurrency?  Times a ConcurrentTrials
pair on the stiff field before and after one d=16 training iteration (whose adjoint kernel spills) in the same process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gan_flow_amd import gu_ffjord  # noqa: E402
from gan_flow_amd.gmm import GaussianMixture  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

dev = torch.device("cuda:0")
trials, xs = [], []
for k in range(2):
    data = make_gu(8, device="cpu", seed=101 + k, extend_dim=True)
    x = data.get_sample(2048).to(dev)
    torch.manual_seed(101 + k)
    model = bench.embed_trained_multimodal(gu_ffjord.build_model(2)).to(dev).train()
    trials.append(gu_ffjord.FusedTrainStep(model, x, lr=1e-3, weight_decay=1e-5, graph=True, spread=False, warmup_batches=[x]))
    xs.append(x)
both = gu_ffjord.ConcurrentTrials(trials)


def timed(tag):
    print("   streams", [hex(st.cuda_stream) for st in both.streams])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        both(xs)
    torch.cuda.synchronize()
    print("%-58s %.2f ms per pair of iterations" % (tag, 1e3 * (time.perf_counter() - t0) / 3), flush=True)


timed("before")
g = GaussianMixture(16, device=dev, sample_seed=1)
torch.manual_seed(1)
m16 = gu_ffjord.build_model(16, divergence_fn="approximate").to(dev).train()
s16 = gu_ffjord.FusedTrainStep(m16, g.get_sample(2048), lr=1e-3, graph=False)
s16(g.get_sample(2048))
torch.cuda.synchronize()
timed("after a d=16 iteration")
both = gu_ffjord.ConcurrentTrials(trials)
timed("new streams")
trials = []
for k in range(2):
    torch.manual_seed(101 + k)
    model = bench.embed_trained_multimodal(gu_ffjord.build_model(2)).to(dev).train()
    trials.append(gu_ffjord.FusedTrainStep(model, xs[k], lr=1e-3, weight_decay=1e-5, graph=True, spread=False, warmup_batches=[xs[k]]))
both = gu_ffjord.ConcurrentTrials(trials)
timed("new trials (graphs captured after the d=16 kernel ran)")
both = gu_ffjord.ConcurrentTrials(trials)
timed("  same trials, other streams")
for t in trials:
    t.graph_saved, t.graph = t.graph, None
timed("  same trials, eager launches instead of graph replays")
for t in trials:
    t.graph = t.graph_saved
s16g = gu_ffjord.FusedTrainStep(m16, g.get_sample(2048), lr=1e-3, graph=True)
s16g(g.get_sample(2048))
torch.cuda.synchronize()
timed("after a d=16 GRAPH replay")
trials = []
for k in range(2):
    torch.manual_seed(101 + k)
    model = bench.embed_trained_multimodal(gu_ffjord.build_model(2)).to(dev).train()
    trials.append(gu_ffjord.FusedTrainStep(model, xs[k], lr=1e-3, weight_decay=1e-5, graph=True, spread=False, warmup_batches=[xs[k]]))
both = gu_ffjord.ConcurrentTrials(trials)
timed("new trials after that")
