"""Evaluation metric and small helpers of the study (reference: /root/reference/util.py:16-47)."""
import logging
import os

import numpy as np
import torch


def w_distance(real_tensor, fake_tensor):
    """1-D Wasserstein-1 distance between two samples, rounded to 4 decimals (util.py:16-20).  For equal-size
    samples (the study always uses 100 000 vs 100 000) W1 is the mean absolute difference of the order statistics;
    unequal sizes use the general CDF form.  Sorting runs on whatever device the tensors are on."""
    a = real_tensor.detach().reshape(-1).double()
    b = fake_tensor.detach().reshape(-1).double().to(a.device)
    if a.numel() == b.numel():
        return float(np.round((a.sort().values - b.sort().values).abs().mean().item(), 4))
    allv = torch.cat([a, b]).sort().values
    deltas = allv[1:] - allv[:-1]
    ca = torch.searchsorted(a.sort().values, allv[:-1], right=True).double() / a.numel()
    cb = torch.searchsorted(b.sort().values, allv[:-1], right=True).double() / b.numel()
    return float(np.round(((ca - cb).abs() * deltas).sum().item(), 4))


def count_parameters(net):
    return sum(p.numel() for p in net.parameters() if p.requires_grad)


def get_logger(log_path=None):
    logger = logging.getLogger("gan_flow_amd")
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        logger.addHandler(logging.StreamHandler())
        if log_path:
            logger.addHandler(logging.FileHandler(log_path, mode="a"))
    return logger


def makedirs(*dirnames):
    for d in dirnames:
        os.makedirs(d, exist_ok=True)
