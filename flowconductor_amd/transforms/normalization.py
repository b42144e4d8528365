"""Normalisation transforms (API of flowcon/transforms/normalization.py:72-218).

The point-wise maps run in ``fc_pointwise_affine``; batch statistics (BatchNorm in training
mode, ActNorm's one-off data-dependent initialisation) are cross-batch reductions outside the
bijector hot path and use torch reductions on the device.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import InverseNotAvailable, Transform, own_autograd_check
from flowconductor_amd.utils import typechecks as check


class BatchNorm(Transform):
    """Batch normalisation for 1-dim inputs; the inverse exists in eval mode only."""

    def __init__(self, features, eps=1e-5, momentum=0.1, affine=True):
        if not check.is_positive_int(features):
            raise TypeError("Number of features must be a positive integer.")
        super().__init__()
        self.momentum = momentum
        self.eps = eps
        constant = np.log(np.exp(1 - eps) - 1)
        self.unconstrained_weight = nn.Parameter(constant * torch.ones(features))
        self.bias = nn.Parameter(torch.zeros(features))
        self.register_buffer("running_mean", torch.zeros(features))
        self.register_buffer("running_var", torch.zeros(features))

    @property
    def weight(self):
        return F.softplus(self.unconstrained_weight) + self.eps

    @staticmethod
    def _check(inputs):
        if inputs.dim() != 2:
            raise ValueError("Expected 2-dim inputs, got inputs of shape: {}".format(inputs.shape))

    def _grad_needed(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _statistics(self, inputs, differentiable):
        """(mean, var) the map uses: the batch's own in training mode (unbiased variance, normalization.py:104-107; the
        running averages move by ``momentum`` towards them), the running averages otherwise."""
        if not self.training:
            return self.running_mean, self.running_var
        with torch.set_grad_enabled(differentiable):
            var, mean = torch.var_mean(inputs, dim=0)
        with torch.no_grad():
            self.running_mean.lerp_(mean.detach(), self.momentum)
            self.running_var.lerp_(var.detach(), self.momentum)
        return mean, var

    def _map(self, inputs, inverse):
        """Both directions, both execution modes.  With gradients required the map is plain torch arithmetic on the device
        (training is a cross-batch reduction, SURVEY 8e: not a row-wise kernel; gradients flow through the batch statistics);
        otherwise the point-wise kernel.  forward: y = w (x - mean) / std + b;  inverse: x = std (y - b) / w + mean."""
        self._check(inputs)
        differentiable = self._grad_needed(inputs)
        mean, var = self._statistics(inputs, differentiable) if not inverse else (self.running_mean, self.running_var)
        std = torch.sqrt(var + self.eps)
        weight, bias = (self.weight, self.bias) if differentiable else (self.weight.detach(), self.bias.detach())
        if differentiable:
            outputs = std * ((inputs - bias) / weight) + mean if inverse else weight * ((inputs - mean) / std) + bias
        else:
            outputs = ops.batchnorm_eval(inputs, mean, std, weight, bias, inverse=inverse)
        log_det = torch.sum(torch.log(weight) - 0.5 * torch.log(var + self.eps))
        return outputs, (-log_det if inverse else log_det) * inputs.new_ones(inputs.shape[0])

    @own_autograd_check
    def forward(self, inputs, context=None):
        return self._map(inputs, inverse=False)

    @own_autograd_check       # the reference's InverseNotAvailable comes first
    def inverse(self, inputs, context=None):
        if self.training:
            raise InverseNotAvailable(
                "Batch norm inverse is only available in eval mode, not in training mode.")
        return self._map(inputs, inverse=True)


class ActNorm(Transform):
    """Activation normalisation (Glow) for [N, D] and [N, C, H, W] inputs, per feature/channel.

    The first forward call in training mode initialises ``log_scale``/``shift`` from the batch so
    that outputs have zero mean and unit variance, and flips the ``initialized`` buffer."""

    _HIP_AUTOGRAD = True

    def __init__(self, features):
        if not check.is_positive_int(features):
            raise TypeError("Number of features must be a positive integer.")
        super().__init__()
        self.register_buffer("initialized", torch.tensor(False, dtype=torch.bool))
        self.log_scale = nn.Parameter(torch.zeros(features))
        self.shift = nn.Parameter(torch.zeros(features))

    @property
    def scale(self):
        return torch.exp(self.log_scale)

    def _broadcastable_scale_shift(self, inputs):
        if inputs.dim() == 4:
            return self.scale.view(1, -1, 1, 1), self.shift.view(1, -1, 1, 1)
        return self.scale.view(1, -1), self.shift.view(1, -1)

    def _run(self, inputs, inverse):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Expecting inputs to be a 2D or a 4D tensor.")
        if not inverse and self.training and not self.initialized:
            self._initialize(inputs)
        scale, shift = self._broadcastable_scale_shift(inputs)
        outputs = ops.pointwise_affine_autograd(inputs, scale[0], shift[0], inverse=inverse)
        total = torch.sum(self.log_scale)
        if inputs.dim() == 4:
            total = inputs.shape[2] * inputs.shape[3] * total
        logabsdet = total * outputs.new_ones(inputs.shape[0])
        return outputs, (-logabsdet if inverse else logabsdet)

    def forward(self, inputs, context=None):
        return self._run(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._run(inputs, inverse=True)

    def _initialize(self, inputs):
        """One-off data-dependent initialisation (normalization.py:206-218): per feature / channel, over every other
        position of the batch, log_scale = -log(std) with the unbiased std and shift = -mean(x / std)."""
        per_feature = inputs if inputs.dim() == 2 else inputs.movedim(1, -1).reshape(-1, inputs.shape[1])
        with torch.no_grad():
            std = per_feature.std(dim=0)
            self.log_scale.copy_(-std.log())
            self.shift.copy_(-(per_feature / std).mean(dim=0))
            self.initialized.fill_(True)
