"""Time-conditioned linear layer of the ODEnet (reference: ffjord/lib/layers/diffeq_layers/basic.py, class path and
child names pinned by the checkpoints)."""
import torch
import torch.nn as nn


class ConcatSquashLinear(nn.Module):
    """Parameter container with the reference's child names (`_layer`, `_hyper_bias`, `_hyper_gate`), so that
    state_dicts interchange.  On the hot path (CNF.forward) its weights are consumed by the HIP kernels; calling
    the module directly evaluates the layer with ordinary tensor ops and is only meant for inspection."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(1, dim_out, bias=False)
        self._hyper_gate = nn.Linear(1, dim_out)

    def forward(self, t, x):
        tt = torch.as_tensor(t).to(x).reshape(1, 1)
        return self._layer(x) * torch.sigmoid(self._hyper_gate(tt)) + self._hyper_bias(tt)
