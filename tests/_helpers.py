"""Shared helpers for the tests: fixtures, paired oracle/product models."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def fixture_state_dict(which):
    a = np.load(os.path.join(GOLDEN, "ffjord_%s.npz" % which))
    return {k: torch.from_numpy(a[k]) for k in a.files}


def appendix_c():
    with open(os.path.join(GOLDEN, "appendix_c.json")) as f:
        return json.load(f)


def oracle_fixture_model(which, dtype=torch.float32):
    import oracle
    m = oracle.build_cnf(1)
    m.load_state_dict(fixture_state_dict(which))
    return m.to(dtype).eval()


def oracle_random_model(dim, seed=1, divergence_fn="brute_force", scale=1.0, **kw):
    """Random-init model (nn.Linear default init under a seed); `scale` > 1 stiffens the field."""
    import oracle
    torch.manual_seed(seed)
    m = oracle.build_cnf(dim, divergence_fn=divergence_fn, **kw)
    if scale != 1.0:
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 2 and p.shape[1] > 1:
                    p.mul_(scale)
    return m


def product_like(oracle_model, device, divergence_fn="brute_force", **kw):
    """The product model (HIP path) carrying the oracle model's weights."""
    from gan_flow_amd import gu_ffjord
    dim = oracle_model.chain[0].odefunc.odefunc.diffeq.layers[-1]._layer.out_features
    blocks = len(oracle_model.chain)
    m = gu_ffjord.build_model(dim, num_blocks=blocks, divergence_fn=divergence_fn, **kw)
    m.load_state_dict({k: v.float() for k, v in oracle_model.state_dict().items()})
    m = m.to(device)
    m.train(oracle_model.training)
    return m


def embed_trained_multimodal(model):
    """The reference's trained gu_num=8 weights (d=1 checkpoint fixture) placed in a d=2 net, nuisance coordinate
    decoupled: bench.py's late-training stand-in, for oracle and product models alike."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from bench import embed_trained_multimodal as embed
    return embed(model)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_err(a, b, floor=1e-3):
    """Per-element relative error with an absolute floor: max |a - b| / max(|b|, floor * max|b|).  rel_err divides by the
    LARGEST element, so rows with small |b| can be far off in their own scale unnoticed; this one looks at every element in
    its own scale, down to `floor` of the largest (below that an fp32 result carries no relative information)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.abs().clamp_min(floor * b.abs().max().clamp_min(1e-30))
    return float(((a - b).abs() / den).max())
