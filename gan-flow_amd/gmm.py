"""Synthetic d-dimensional Gaussian mixture: the data of BASELINE config 5 / SURVEY.md section 8d ("d=16 GMM: 8 components,
means ~ N(0, 4^2 I) (seed 0), sigma = 1").  The reference has no such module (its data is gu.py's 1-D mixture); this is
the synthetic workload the survey defines for the Hutchinson-divergence path, generated on the device."""
import math

import torch


class GaussianMixture:
    def __init__(self, dim=16, n_components=8, mean_scale=4.0, sigma=1.0, seed=0, device="cpu", sample_seed=None):
        g = torch.Generator().manual_seed(seed)
        self.dim, self.n_components, self.sigma = dim, n_components, float(sigma)
        self.device = torch.device(device)
        self.means = (torch.randn(n_components, dim, generator=g) * mean_scale).to(self.device)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed if sample_seed is None else sample_seed)

    def get_sample(self, n):
        comp = torch.randint(0, self.n_components, (n,), generator=self._gen, device=self.device)
        noise = torch.randn(n, self.dim, generator=self._gen, device=self.device)
        return self.means[comp] + self.sigma * noise

    def log_prob(self, x):
        """Exact log-density (n,) of the mixture: what a trained flow's log-likelihood is compared with."""
        d2 = ((x[:, None, :] - self.means[None].to(x)) ** 2).sum(-1)
        logc = -0.5 * d2 / self.sigma ** 2 - self.dim * (math.log(self.sigma) + 0.5 * math.log(2 * math.pi))
        return torch.logsumexp(logc, dim=1) - math.log(self.n_components)

    def entropy_estimate(self, n=100000):
        """-E[log p(x)] by Monte Carlo: the floor of the negative log-likelihood loss."""
        return float(-self.log_prob(self.get_sample(n)).mean())
