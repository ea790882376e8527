"""Gaussian + uniform mixture: the training data of the study (reference: /root/reference/gu.py:8-98).

Same constructor, same `get_sample` / `density` results for the same seed (the sampler draws from numpy's global
RNG in the reference's order: component choice, then per component a binomial split, the uniform part, the
Gaussian part, and one shuffle at the end; checked against vectors generated from the reference in
tests/golden/gu_samples.npz)."""
import math

import numpy as np
import torch


def _per_component(value, n):
    return np.full(n, value) if isinstance(value, (int, float)) else np.asarray(value)


class GausUniffMixture:
    def __init__(self, n_mixture, mean_dist, sigma, unif_intsect, centralized=False, unif_ratio=1.0, device="cpu", seed=1,
                 extend_dim=False):
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.seed = seed
        self.device = torch.device(device)
        self.n_mixture = n_mixture
        self.mean = np.cumsum(tuple(_per_component(mean_dist, n_mixture)))
        if centralized:
            self.mean = self.mean - np.mean(self.mean)
        self.sigma = tuple(_per_component(sigma, n_mixture))
        self.unif_intsect = tuple(_per_component(unif_intsect, n_mixture))
        ratio = _per_component(unif_ratio, n_mixture)
        self.unif_p = tuple(ratio / (ratio + 1))
        self.extend_dim = extend_dim

    def _bounds(self, c):
        half = self.unif_intsect[c] * self.sigma[c]
        return self.mean[c] - half, self.mean[c] + half

    def get_sample(self, size):
        which = np.random.choice(self.n_mixture, size, p=[1.0 / self.n_mixture] * self.n_mixture)
        chunks = []
        for c in range(self.n_mixture):
            n_c = int(np.count_nonzero(which == c))
            n_unif = np.random.binomial(n=n_c, p=self.unif_p[c])
            lo, hi = self._bounds(c)
            chunks.append(np.random.uniform(lo, hi, n_unif))
            chunks.append(np.random.normal(loc=self.mean[c], scale=self.sigma[c], size=(n_c - n_unif)))
        points = np.concatenate(chunks)
        np.random.shuffle(points)
        dat = torch.from_numpy(points.astype("float32")).to(self.device).view(size, -1)
        if self.extend_dim:  # a U(0,1) nuisance column, so that 2-D flows have something to couple with
            dat = torch.cat((dat, torch.rand(size=dat.size(), device=self.device)), 1)
        return dat

    def device_sampler(self, device, seed=None):
        """The same distribution drawn ON the GPU by the library's kernel (cnf_gu_sample): no host sampler, no copy per
        iteration.  Another random stream than get_sample's (numpy's global state cannot be reproduced on a device)."""
        return DeviceSampler(self, device, self.seed if seed is None else seed)

    def density(self, x):
        total = 0.0
        for c in range(self.n_mixture):
            lo, hi = self._bounds(c)
            unif = 1 / (hi - lo) * (x <= hi) * (x >= lo)
            gaus = np.exp(-((x - self.mean[c]) / self.sigma[c]) ** 2 / 2) / (np.sqrt(2 * math.pi) * self.sigma[c])
            total = total + (gaus * (1 - self.unif_p[c]) + unif * self.unif_p[c]) / self.n_mixture
        return total


class DeviceSampler:
    """Batches of a GausUniffMixture drawn by the HIP kernel behind cnf_gu_sample (include/cnf_hip.h).
    `fill(out)` writes a fresh batch into a (rows, 1 | 2) fp32 tensor on the device -- inside a captured graph too: the call
    counter lives in device memory and the kernel advances it, so every replay draws new rows."""

    def __init__(self, mixture, device, seed):
        from . import _native
        self._native = _native
        self.device = torch.device(device)
        self.seed = int(seed)
        self.cols = 2 if mixture.extend_dim else 1
        self.desc = _native.gu_desc(mixture.mean, mixture.sigma, mixture.unif_intsect, mixture.unif_p, mixture.extend_dim)
        self.state = torch.zeros(2, dtype=torch.int32, device=self.device)

    def fill(self, out):
        return self._native.gu_sample(self.desc, self.seed, self.state, out)

    def get_sample(self, size):
        return self.fill(torch.empty(size, self.cols, dtype=torch.float32, device=self.device))

    def get_rows(self, lo, hi, call=0):
        """Rows lo .. hi-1 of draw number `call`, addressed by GLOBAL row index (cnf_gu_sample_rows): stateless, so the
        shards of a data set hold the same values for every number of shards (sharded evaluation, BASELINE config 3)."""
        out = torch.empty(max(0, hi - lo), self.cols, dtype=torch.float32, device=self.device)
        return self._native.gu_sample_rows(self.desc, self.seed, call, lo, out)


class DeviceNormalRows:
    """N(mean, sigma^2) rows of `cols` in {1, 2} columns by global row index, from the same counter-based generator
    (a one-component mixture without a uniform part; with two columns the second is U(0, 1), the data sets' nuisance
    column, so use cols = 1 per coordinate and stack for a d-dimensional normal)."""

    def __init__(self, device, seed, mean=0.0, sigma=1.0):
        from . import _native
        self._native = _native
        self.device = torch.device(device)
        self.seed = int(seed)
        self.desc = _native.gu_desc([mean], [sigma], [0.0], [0.0], False)

    def get_rows(self, lo, hi, dim=1, call=0):
        cols = [self._native.gu_sample_rows(self.desc, self.seed, call * 16 + c, lo,
                                            torch.empty(max(0, hi - lo), 1, dtype=torch.float32, device=self.device))
                for c in range(dim)]
        return cols[0] if dim == 1 else torch.cat(cols, 1)


def make_gu(gu_num, device="cpu", seed=1, extend_dim=True):
    """The two data sets of the study (/root/reference/model_plot.py:36-55, gu.py:105-116)."""
    if gu_num == 8:
        return GausUniffMixture(8, 10, 2, 1.5, unif_ratio=1.0, device=device, seed=seed, extend_dim=extend_dim)
    return GausUniffMixture(gu_num, 5, 0.1, 5, unif_ratio=3, device=device, seed=seed, extend_dim=extend_dim)
