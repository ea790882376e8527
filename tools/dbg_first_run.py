#!/usr/bin/env python3
"""Why is the first timed loop of a process slower than its repeats?  Times the bench's W + K loop (5 + 20 iterations,
fresh seed-1 model and graph each time) several times in one process, optionally after PRESPIN_MS of unrelated GPU work
(forward solves), and prints ms per step of every run plus the per-step times of the first.  usage: dbg_first_run.py [prespin_ms]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_flow_amd import _native, gu_ffjord  # noqa: E402
from gan_flow_amd.gu import make_gu  # noqa: E402

prespin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda:0")
_native.load_library()
W, K = 5, 20
data = make_gu(8, device="cpu", seed=1, extend_dim=True)
batches = [data.get_sample(2048).to(dev) for _ in range(W + K)]
if prespin > 0:
    torch.manual_seed(7)
    m = gu_ffjord.build_model(2).to(dev).eval()
    t0 = time.perf_counter()
    n = 0
    with torch.no_grad():
        while 1e3 * (time.perf_counter() - t0) < prespin:
            m(batches[0], torch.zeros(2048, 1, device=dev))
            n += 1
            if n % 16 == 0:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("prespin: %d forward solves in %.0f ms" % (n, 1e3 * (time.perf_counter() - t0)))
for run in range(4):
    torch.manual_seed(1)
    model = gu_ffjord.build_model(2).to(dev).train()
    step = gu_ffjord.FusedTrainStep(model, batches[0], lr=1e-3, weight_decay=1e-5, graph=True, spread=True, warmup_batches=batches[:W])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i, x in enumerate(batches[W:]):
        step(x)
        ev[i + 1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = [1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print("run %d: %.4f ms/step (host clock) | per step us: %s" % (run, 1e3 * el / K, " ".join("%.0f" % p for p in per)))
