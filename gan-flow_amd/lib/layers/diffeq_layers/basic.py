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

    dim_in = property(lambda self: self._layer.in_features)
    dim_out = property(lambda self: self._layer.out_features)

    def kernel_slices(self):
        """(W, b, w_hb, w_g, b_g) as the kernels read a layer: y = (W x + b) * sigmoid(w_g t + b_g) + w_hb t."""
        return (self._layer.weight, self._layer.bias, self._hyper_bias.weight, self._hyper_gate.weight, self._hyper_gate.bias)

    def forward(self, t, x):
        tt = torch.as_tensor(t).to(x).reshape(1, 1)
        return self._layer(x) * torch.sigmoid(self._hyper_gate(tt)) + self._hyper_bias(tt)


# The other time-conditioned layers of ffjord/lib/layers/diffeq_layers/basic.py that are special cases of the form the
# kernels evaluate (class and child names UPSTREAM-RECALL: the submodule is not vendored and
# no checkpoint uses them, so their parity is unpinned).  Each one presents itself to the kernels as a concatsquash
# layer through kernel_slices(): a gate that is identically 1 is sigmoid(0 * t + 30) (== 1.0f in fp32, and its
# gradient factor g (1 - g) is exactly 0), an absent time term is a zero hyper-bias.  The constants are not
# parameters: autograd routes the kernels' gradient of the flat vector back to the real parameters only.
_GATE_OPEN = 30.0


def _const(like, n, value):
    return torch.full((n, 1), value, dtype=like.dtype, device=like.device)


class IgnoreLinear(nn.Module):
    """y = Linear(x): time is ignored."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self.dim_in, self.dim_out = dim_in, dim_out

    def kernel_slices(self):
        w, n = self._layer.weight, self.dim_out
        return (w, self._layer.bias, _const(w, n, 0.0), _const(w, n, 0.0), _const(w, n, _GATE_OPEN).reshape(-1))

    def forward(self, t, x):
        return self._layer(x)


class ConcatLinear(nn.Module):
    """y = Linear([t, x]): time enters as an extra FIRST input column."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in + 1, dim_out)
        self.dim_in, self.dim_out = dim_in, dim_out

    def kernel_slices(self):
        w, n = self._layer.weight, self.dim_out
        return (w[:, 1:], self._layer.bias, w[:, :1], _const(w, n, 0.0), _const(w, n, _GATE_OPEN).reshape(-1))

    def forward(self, t, x):
        tt = torch.ones_like(x[:, :1]) * torch.as_tensor(t).to(x)
        return self._layer(torch.cat([tt, x], 1))


class ConcatLinear_v2(nn.Module):
    """y = Linear(x) + hyper_bias(t)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(1, dim_out, bias=False)
        self.dim_in, self.dim_out = dim_in, dim_out

    def kernel_slices(self):
        w, n = self._layer.weight, self.dim_out
        return (w, self._layer.bias, self._hyper_bias.weight, _const(w, n, 0.0), _const(w, n, _GATE_OPEN).reshape(-1))

    def forward(self, t, x):
        return self._layer(x) + self._hyper_bias(torch.as_tensor(t).to(x).reshape(1, 1))


class SquashLinear(nn.Module):
    """y = Linear(x) * sigmoid(hyper(t))."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper = nn.Linear(1, dim_out)
        self.dim_in, self.dim_out = dim_in, dim_out

    def kernel_slices(self):
        w, n = self._layer.weight, self.dim_out
        return (w, self._layer.bias, _const(w, n, 0.0), self._hyper.weight, self._hyper.bias)

    def forward(self, t, x):
        return self._layer(x) * torch.sigmoid(self._hyper(torch.as_tensor(t).to(x).reshape(1, 1)))


LAYER_TYPES = {"ignore": IgnoreLinear, "concat": ConcatLinear, "concatcoord": ConcatLinear, "concat_v2": ConcatLinear_v2,
               "squash": SquashLinear, "concatsquash": ConcatSquashLinear}
