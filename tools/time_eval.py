#!/usr/bin/env python3
"""Times the tile-local density evaluation (BASELINE config 3 kernel: dopri5_kernel<1,false,false>) on the reference's
trained gu_num=8 d=1 field: usage time_eval.py [log2 rows] [log2 rows per launch, ...] [dim].  Prints rows/s and TFLOP/s."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gan_flow_amd import gu_ffjord, parallel_eval  # noqa: E402

dev = torch.device("cuda:0")
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
chunks = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [18]
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if dim == 1:
    a = np.load(os.path.join(ROOT, "tests", "golden", "ffjord_multimodal.npz"))
    em = gu_ffjord.build_model(1)
    em.load_state_dict({k: torch.from_numpy(a[k]) for k in a.files})
else:
    em = bench.embed_trained_multimodal(gu_ffjord.build_model(2))   # the same field in the first coordinate of d=2
em = em.to(dev).eval()
for b in em.chain:
    b.tile_rows = 16
g = torch.Generator(device=dev).manual_seed(1234)
half = n // 2
comp = torch.randint(0, 8, (half,), generator=g, device=dev).float() * 10 + 10
xs = torch.cat([comp + (torch.rand(half, generator=g, device=dev) * 6 - 3),
                45 + 23 * torch.randn(n - half, generator=g, device=dev)]).reshape(-1, 1)
if dim == 2:
    xs = torch.cat([xs, torch.randn(n, 1, generator=g, device=dev)], 1)
F = bench.F_EVAL_D1 if dim == 1 else bench.F_EVAL_D2
with torch.no_grad():
    parallel_eval.model_log_prob(em, xs[: 1 << 14], chunk=1 << 14)
    for c in chunks:
        evals = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = 0.0
        for vc in torch.split(xs, 1 << c):
            lp = parallel_eval.model_log_prob(em, vc, chunk=1 << c)
            evals += em.chain[0].last_stats["forward"].row_evals
            tot += lp.double().sum().item()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("rows 2^%d, 2^%d per launch: %.3f s  %.3f M rows/s  %.1f TFLOP/s  mean logp %.5f"
              % (n.bit_length() - 1, c, el, n / el / 1e6, evals * F / el / 1e12, tot / n), flush=True)
