#!/usr/bin/env python3
"""H=256 adjoint, the Hutchinson test case of tests/test_gpu_wide.py: distance of the kernel's and of the fp32 oracle's
gradients from a float64 adjoint solved at 1e-10, and the step sizes both took.  CNF_HIP_LIB selects the library."""
import os, sys, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from _helpers import oracle_random_model, rel_err
from gan_flow_amd import _native as nat
DEV = "cuda:0"
cases = [(2, True, 256), (2, False, 100), (2, False, 2048)]
for dim, hutch, rows in cases:
    g = torch.Generator().manual_seed(23)
    m = oracle_random_model(dim, seed=7, divergence_fn="approximate" if hutch else "brute_force", scale=2.0, hidden_dims=(256, 256, 256), num_blocks=1)
    m.train()
    m64 = copy.deepcopy(m).double()
    x = torch.randn(rows, dim, generator=g)
    e = torch.randn(rows, dim, generator=g) if hutch else None
    def grads(model, xx, ee, tol):
        fn = model.chain[0].odefunc.odefunc
        if ee is not None:
            orig = fn.before_odeint
            fn.before_odeint = lambda e=None: orig(e=ee)
        model.chain[0].rtol = model.chain[0].atol = tol
        model.zero_grad()
        xx = xx.clone().requires_grad_(True)
        z, dl = model(xx, torch.zeros(rows, 1, dtype=xx.dtype))
        loss = -(oracle.standard_normal_logprob(z).sum(1, keepdim=True) - dl).mean()
        loss.backward()
        return torch.cat([p.grad.reshape(-1) for p in fn.diffeq.parameters()]), xx.grad, model.chain[0].stats
    gp32, gx32, st32 = grads(m, x, e, 1e-5)
    gp64, gx64, st64 = grads(m64, x.double(), None if e is None else e.double(), 1e-10)
    fn = m.chain[0].odefunc.odefunc
    T = float(m.chain[0].sqrt_end_time.detach() ** 2)
    net = nat.net_desc(dim, 256, 3, hutch)
    params = torch.cat([p.detach().reshape(-1) for p in fn.diffeq.parameters()]).float().to(DEV)
    opts = nat.make_opts("dopri5", 1e-5, 1e-5)
    ed = None if e is None else e.to(DEV)
    z1, l1, s_f = nat.solve_fwd(net, params, x.to(DEV), None, 0.0, T, opts, ed)
    gx, gl0, gp, s_b = nat.solve_bwd(net, params, z1, l1, z1 / rows, torch.full((rows, 1), 1.0 / rows, device=DEV), 0.0, T, opts, ed)
    torch.cuda.synchronize()
    s_b = nat.Stats(s_b)
    print("d=%d hutch=%s rows=%d lib=%s" % (dim, hutch, rows, os.environ.get("CNF_HIP_LIB", "default")))
    print("  steps: kernel %d+%d last_dt %.6f | oracle32 %s" % (s_b.n_accepted, s_b.n_rejected, s_b.last_dt, st32["backward"]))
    print("  grad_theta: kernel vs oracle32 %.2e | kernel vs f64 %.2e | oracle32 vs f64 %.2e" % (rel_err(gp, gp32), rel_err(gp, gp64.float()), rel_err(gp32, gp64.float())))
    print("  grad_x    : kernel vs oracle32 %.2e | kernel vs f64 %.2e | oracle32 vs f64 %.2e" % (rel_err(gx, gx32), rel_err(gx, gx64.float()), rel_err(gx32, gx64.float())))
