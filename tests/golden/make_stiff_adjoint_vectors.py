#!/usr/bin/env python3
"""Float64 tight-tolerance adjoint gradients on the reference's trained STIFF field (gu_num=8 checkpoint weights).

The regime BASELINE config 2 spends > 99 % of its iterations in (NFE 300-1 700 per solve): the multimodal checkpoint's
field reaches |f| ~ 278 over T = 0.5 and dopri5(1e-5) rejects every seventh attempt.  The fp32 oracle and the fp32 HIP
kernel both decide accept/reject within rounding there, so their step sequences differ; what both must approximate is
the SOLUTION.  This script computes it once, on the CPU, with the oracle in float64 at rtol = atol = 1e-9 (about an hour of
one core for both cases), and stores inputs + gradients as data:

    tests/golden/stiff_adjoint_d1.npz   d = 1 checkpoint model (tests/golden/ffjord_multimodal.npz), x ~ U(7, 83)
    tests/golden/stiff_adjoint_d2.npz   the same weights embedded in the d = 2 net as bench.embed_trained_multimodal does
                                        (nuisance coordinate decoupled), x = [U(7, 83), U(0, 1)]
    keys: x (rows, d) fp32; loss, z, dlogp, grad_x, grad_theta (float64, loss = -mean(log N(z) - dlogp));
          nfe_forward, nfe_backward of the float64 solve; check_* = the same quantities at rtol = atol = 1e-10 on the first
          64 rows' loss share (how converged the reference itself is).

Weights come from the committed fixture (cut out of /root/reference/model_results/multimodal/ffjord.pth by
make_fixtures.py); nothing of the reference is read here.  Run from the repo root:
    python tests/golden/make_stiff_adjoint_vectors.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from _helpers import embed_trained_multimodal, fixture_state_dict  # noqa: E402

ROWS = 384


def model_for(dim, dtype, tol):
    m = oracle.build_cnf(dim, atol=tol, rtol=tol)
    if dim == 1:
        m.load_state_dict(fixture_state_dict("multimodal"))
    else:
        embed_trained_multimodal(m)
    return m.to(dtype).train()


def grads(m, x):
    m.zero_grad()
    xx = x.clone().requires_grad_(True)
    z, dl = m(xx, torch.zeros(x.shape[0], 1, dtype=x.dtype))
    loss = -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()
    loss.backward()
    gp = torch.cat([p.grad.reshape(-1) for p in m.chain[0].odefunc.odefunc.diffeq.parameters()])
    st = m.chain[0].stats
    return loss.item(), z.detach(), dl.detach(), xx.grad.clone(), gp, st


def main():
    torch.set_num_threads(4)
    for dim in (1, 2):
        g = torch.Generator().manual_seed(2024 + dim)
        x = torch.rand(ROWS, 1, generator=g) * 76.0 + 7.0
        if dim == 2:
            x = torch.cat([x, torch.rand(ROWS, 1, generator=g)], 1)
        t0 = time.time()
        loss, z, dl, gx, gp, st = grads(model_for(dim, torch.float64, 1e-9), x.double())
        nf = 2 + 6 * (st["forward"]["n_accepted"] + st["forward"]["n_rejected"])
        nb = 2 + 6 * (st["backward"]["n_accepted"] + st["backward"]["n_rejected"])
        print("d=%d float64 1e-9: loss %.9f NFE %d / %d (%.0f s)" % (dim, loss, nf, nb, time.time() - t0), flush=True)
        # convergence of the reference itself: a subset at 1e-10
        sub = x[:64].double()
        a = grads(model_for(dim, torch.float64, 1e-9), sub)
        b = grads(model_for(dim, torch.float64, 1e-10), sub)
        dgx = float((a[3] - b[3]).abs().max() / b[3].abs().max())
        dgp = float((a[4] - b[4]).abs().max() / b[4].abs().max())
        print("   1e-9 vs 1e-10 on 64 rows: grad_x %.2e grad_theta %.2e (max-norm relative)" % (dgx, dgp), flush=True)
        np.savez(os.path.join(HERE, "stiff_adjoint_d%d.npz" % dim), x=x.numpy(), loss=np.float64(loss), z=z.numpy(),
                 dlogp=dl.numpy(), grad_x=gx.numpy(), grad_theta=gp.numpy(), nfe_forward=np.int64(nf),
                 nfe_backward=np.int64(nb), check_grad_x=np.float64(dgx), check_grad_theta=np.float64(dgp))


if __name__ == "__main__":
    main()
