"""Chain of flow blocks (reference: ffjord/lib/layers/container.py)."""
import torch.nn as nn


class SequentialFlow(nn.Module):
    """model(x, logpx) -> (z, delta_logp); `reverse=True` runs the chain backwards (sampling direction)."""

    def __init__(self, layersList):
        super().__init__()
        self.chain = nn.ModuleList(layersList)

    def forward(self, x, logpx=None, reverse=False, inds=None):
        if inds is None:
            inds = range(len(self.chain) - 1, -1, -1) if reverse else range(len(self.chain))
        if logpx is None:
            for i in inds:
                x = self.chain[i](x, reverse=reverse)
            return x
        for i in inds:
            x, logpx = self.chain[i](x, logpx, reverse=reverse)
        return x, logpx
