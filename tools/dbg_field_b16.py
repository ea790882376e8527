#!/usr/bin/env python3
"""Debugging aid for the bf16-split layers: one field evaluation (cnf_eval_field) of the trained fixtures against the
float64 evaluation of the same weights (oracle), with whichever library CNF_HIP_LIB names.  Build the variant whose
cnf_eval_field runs field_fwd_b16 with `SRC=cnf_fwd tools/build_variants.sh fieldb16 -DCNF_DBG_FIELD_B16`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _helpers import oracle_fixture_model  # noqa: E402
from gan_flow_amd import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
for which, lo, hi in (("unimodal", 4.5, 5.5), ("multimodal", 7.0, 83.0)):
    m = oracle_fixture_model(which)
    fn = m.chain[0].odefunc.odefunc
    flat = torch.cat([p.detach().reshape(-1) for p in fn.diffeq.parameters()])
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4096, 1, generator=g) * (hi - lo) + lo
    m64 = oracle_fixture_model(which).double()
    fn64 = m64.chain[0].odefunc.odefunc
    for t in (0.0, 0.2, 0.45):
        f, nd = nat.eval_field(nat.net_desc(1, 64, 3, False), flat.to(dev), t, x.to(dev))
        fn64.before_odeint()
        with torch.enable_grad():
            xr = x.double().requires_grad_(True)
            fr, ndr = fn64(torch.tensor(t, dtype=torch.float64), (xr, torch.zeros(4096, 1, dtype=torch.float64)))[:2]
        ef = (f.cpu().double() - fr.detach()).abs()
        ed = (nd.cpu().double().reshape(-1) - ndr.detach().reshape(-1)).abs()
        print("%-10s t=%.2f  f: max err %.2e (max |f| %.2e)   -div: max err %.2e (max |div| %.2e)"
              % (which, t, ef.max(), fr.abs().max(), ed.max(), ndr.abs().max()))
