"""Batch-sharded log-likelihood / sampling evaluation over the GPUs of one node (BASELINE config 3).

The path shards by rows: every rank integrates its contiguous shard of the points with tile-local step control
(no cross-rank traffic during the solve) and the only exchange is ONE all-reduce of two float64 numbers
{sum of log p, row count} -- 16 bytes over RCCL/xGMI (backend "nccl" on ROCm), or gloo in the CPU tests.
Training does not shard (batch-global step control would need a cross-GPU reduction per Runge-Kutta step):
independent trials run one per GPU, as the reference's own runs on cuda:0..7 did.
"""
import math

import torch
import torch.distributed as dist


def shard_bounds(n_rows, rank, world):
    """Contiguous, near-equal split: the first n_rows % world ranks get one extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def model_log_prob(model, x, chunk=1 << 16):
    """log p(x) per row under the flow: sum_d log N(z_d; 0, 1) - delta_logp."""
    out = []
    for xc in torch.split(x, chunk):
        z, dl = model(xc, torch.zeros(xc.shape[0], 1, device=xc.device, dtype=xc.dtype))
        logpz = (-0.5 * math.log(2 * math.pi) - z.pow(2) / 2).reshape(z.shape[0], -1).sum(1, keepdim=True)
        out.append(logpz - dl)
    return torch.cat(out, 0)


def sharded_mean_log_likelihood(log_prob_fn, x_local, group=None):
    """Mean log-likelihood over ALL ranks' rows.  `x_local` is this rank's shard (may be empty);
    `log_prob_fn(x) -> (rows, 1)` is the per-row log density (the HIP path in production; the CPU tests inject the
    oracle).  Returns (mean, total_rows).  One collective: all_reduce(sum) of [sum_logp, rows] in float64."""
    if x_local.shape[0] > 0:
        lp = log_prob_fn(x_local)
        acc = torch.stack([lp.double().sum(), torch.tensor(float(x_local.shape[0]), dtype=torch.float64, device=lp.device)])
    else:
        acc = torch.zeros(2, dtype=torch.float64, device=x_local.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    total = acc[1].item()
    return (acc[0] / acc[1]).item() if total > 0 else float("nan"), int(total)


def sharded_sample(sample_fn, n_total, dim, rank, world, device, seed=0, chunk=1 << 16):
    """This rank's shard of n_total model samples: z ~ N(0, I) drawn from a per-rank stream, pushed through sample_fn.
    Samples stay sharded (the Wasserstein metric of the study only needs them gathered if the sort is centralised)."""
    lo, hi = shard_bounds(n_total, rank, world)
    g = torch.Generator(device=device).manual_seed(seed * 1000003 + rank)
    z = torch.randn(hi - lo, dim, generator=g, device=device)
    return torch.cat([sample_fn(zc) for zc in torch.split(z, chunk)], 0) if hi > lo else z
