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


def sharded_sample(sample_fn, n_total, dim, rank, world, device, seed=0, chunk=1 << 16, prior_rows=None):
    """This rank's shard of n_total model samples: z ~ N(0, I) pushed through sample_fn.  Samples stay sharded
    (sharded_w_distance gathers them for the study's metric).
    prior_rows(lo, hi) -> (hi - lo, dim): the prior draws of GLOBAL rows lo .. hi-1 (gu.DeviceNormalRows.get_rows on the
    HIP path): the n_total samples are then the same set for every world size.  Without it the draws come from a
    per-rank torch stream (the set depends on the world size)."""
    lo, hi = shard_bounds(n_total, rank, world)
    if prior_rows is not None:
        z = prior_rows(lo, hi)
    else:
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + rank)
        z = torch.randn(hi - lo, dim, generator=g, device=device)
    return torch.cat([sample_fn(zc) for zc in torch.split(z, chunk)], 0) if hi > lo else z


def gather_shards(x_local, n_total, group=None):
    """All ranks' contiguous shards (shard_bounds) of a 1-D tensor, concatenated in rank order, on every rank: ONE
    all_gather of the shards padded to the longest (the sizes follow from n_total: no size exchange)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x_local
    world = dist.get_world_size(group)
    sizes = [b - a for a, b in (shard_bounds(n_total, r, world) for r in range(world))]
    if x_local.numel() != sizes[dist.get_rank(group)]:
        raise ValueError("shard of %d rows, shard_bounds gives %d" % (x_local.numel(), sizes[dist.get_rank(group)]))
    pad = torch.zeros(max(sizes), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.numel()] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)])


def sharded_w_distance(real_local, fake_local, n_real, n_fake, group=None):
    """The study's evaluation metric over ALL ranks' rows: util.w_distance (scipy's 1-D Wasserstein distance rounded to four
    decimals, /root/reference/util.py:16-20) between the real and the generated sample of the protocol at
    /root/reference/model_plot.py:93-101, whose halves live sharded on the ranks.  Two all_gathers (4 bytes per row each;
    2^24 rows = 64 MB per tensor, a fraction of a second over xGMI), then every rank sorts and integrates the same arrays, so
    all ranks return the same number.  A rank-count-independent result needs rank-count-independent samples: see
    sharded_sample(prior_rows=...) and gu.DeviceSampler.get_rows."""
    from .util import w_distance
    real = gather_shards(real_local.reshape(-1).contiguous(), n_real, group)
    fake = gather_shards(fake_local.reshape(-1).contiguous(), n_fake, group)
    return w_distance(real, fake)
