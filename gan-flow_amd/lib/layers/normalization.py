"""Moving batch normalisation between flow blocks (reference: ffjord/lib/layers/normalization.py, used by
`build_model_tabular(..., batch_norm=True)`).  UPSTREAM-RECALL: the submodule is not vendored
and no checkpoint of the reference uses it, so names and arithmetic follow the published FFJORD code as recalled --
parity unpinned.  An elementwise map over (rows, d) with d <= 16: ordinary tensor ops on the device, outside the
integrators; it is a flow layer like CNF (same call signature, contributes its log-determinant)."""
import torch
import torch.nn as nn


class MovingBatchNormNd(nn.Module):
    def __init__(self, num_features, eps=1e-4, decay=0.1, bn_lag=0.0, affine=True):
        super().__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        self.decay = decay
        self.bn_lag = bn_lag
        self.register_buffer("step", torch.zeros(1))
        if self.affine:
            self.weight = nn.Parameter(torch.zeros(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    @property
    def shape(self):
        raise NotImplementedError

    def reset_parameters(self):
        self.running_mean.zero_()
        self.running_var.fill_(1)
        if self.affine:
            self.weight.data.zero_()
            self.bias.data.zero_()

    def forward(self, x, logpx=None, reverse=False):
        return self._reverse(x, logpx) if reverse else self._forward(x, logpx)

    def _forward(self, x, logpx=None):
        c = x.size(1)
        used_mean = self.running_mean.clone().detach()
        used_var = self.running_var.clone().detach()
        if self.training:
            # batch statistics over everything but the feature axis
            x_t = x.transpose(0, 1).contiguous().view(c, -1)
            batch_mean = torch.mean(x_t, dim=1)
            batch_var = torch.var(x_t, dim=1)
            # with a lag, normalise by a de-biased mix of batch and running statistics (differentiable through the batch
            # part); without one (the default) by the running statistics as they were before this batch
            if self.bn_lag > 0:
                used_mean = batch_mean - (1 - self.bn_lag) * (batch_mean - used_mean.detach())
                used_mean = used_mean / (1.0 - self.bn_lag ** (self.step[0] + 1))
                used_var = batch_var - (1 - self.bn_lag) * (batch_var - used_var.detach())
                used_var = used_var / (1.0 - self.bn_lag ** (self.step[0] + 1))
            self.running_mean -= self.decay * (self.running_mean - batch_mean.data)
            self.running_var -= self.decay * (self.running_var - batch_var.data)
            self.step += 1
        used_mean = used_mean.view(*self.shape).expand_as(x)
        used_var = used_var.view(*self.shape).expand_as(x)
        y = (x - used_mean) * torch.exp(-0.5 * torch.log(used_var + self.eps))
        if self.affine:
            weight = self.weight.view(*self.shape).expand_as(x)
            bias = self.bias.view(*self.shape).expand_as(x)
            y = y * torch.exp(weight) + bias
        if logpx is None:
            return y
        return y, logpx - self._logdetgrad(x, used_var).view(x.size(0), -1).sum(1, keepdim=True)

    def _reverse(self, y, logpy=None):
        used_mean = self.running_mean
        used_var = self.running_var
        if self.affine:
            weight = self.weight.view(*self.shape).expand_as(y)
            bias = self.bias.view(*self.shape).expand_as(y)
            y = (y - bias) * torch.exp(-weight)
        used_mean = used_mean.view(*self.shape).expand_as(y)
        used_var = used_var.view(*self.shape).expand_as(y)
        x = y * torch.exp(0.5 * torch.log(used_var + self.eps)) + used_mean
        if logpy is None:
            return x
        return x, logpy + self._logdetgrad(x, used_var).view(x.size(0), -1).sum(1, keepdim=True)

    def _logdetgrad(self, x, used_var):
        logdetgrad = -0.5 * torch.log(used_var + self.eps)
        if self.affine:
            logdetgrad = logdetgrad + self.weight.view(*self.shape).expand(*x.size())
        return logdetgrad

    def __repr__(self):
        return ("{name}({num_features}, eps={eps}, decay={decay}, bn_lag={bn_lag}, affine={affine})"
                .format(name=self.__class__.__name__, **self.__dict__))


class MovingBatchNorm1d(MovingBatchNormNd):
    @property
    def shape(self):
        return [1, -1]
