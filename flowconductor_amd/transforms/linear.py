"""Linear transforms that parameterise a weight matrix (API of flowcon/transforms/linear.py:15-126).

``Linear`` keeps the reference's eval-mode cache semantics (``using_cache``, ``use_cache()``,
invalidation on ``train()``); the mat-vecs run in ``fc_linear``.
"""
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.utils import typechecks as check


class LinearCache:
    """Weight matrix, its inverse and its log|det|, filled lazily in eval mode."""

    def __init__(self):
        self.weight = None
        self.inverse = None
        self.logabsdet = None

    def invalidate(self):
        self.weight = None
        self.inverse = None
        self.logabsdet = None


class Linear(Transform):
    """Abstract base class for linear transforms that parameterize a weight matrix."""

    def __init__(self, features, using_cache=False):
        if not check.is_positive_int(features):
            raise TypeError("Number of features must be a positive integer.")
        super().__init__()
        self.features = features
        self.bias = nn.Parameter(torch.zeros(features))
        self.using_cache = using_cache
        self.cache = LinearCache()

    def _grad_needed(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, inputs, context=None):
        if not self.training and self.using_cache and not self._grad_needed(inputs):
            self._check_forward_cache()
            outputs = ops.linear(inputs, self.cache.weight, bias=self.bias, mode=ops.LINEAR_DENSE)
            return outputs, self.cache.logabsdet * outputs.new_ones(outputs.shape[0])
        return self.forward_no_cache(inputs)

    def _check_forward_cache(self):
        if self.cache.weight is None and self.cache.logabsdet is None:
            self.cache.weight, self.cache.logabsdet = self.weight_and_logabsdet()
        elif self.cache.weight is None:
            self.cache.weight = self.weight()
        elif self.cache.logabsdet is None:
            self.cache.logabsdet = self.logabsdet()

    def inverse(self, inputs, context=None):
        if not self.training and self.using_cache and not self._grad_needed(inputs):
            self._check_inverse_cache()
            # F.linear(inputs - bias, W^-1) == W^-1 inputs + (-(W^-1 bias)); the [D] constant is host-side
            shift = -(self.cache.inverse.detach() @ self.bias.detach())
            outputs = ops.linear(inputs, self.cache.inverse, bias=shift, mode=ops.LINEAR_DENSE)
            return outputs, (-self.cache.logabsdet) * outputs.new_ones(outputs.shape[0])
        return self.inverse_no_cache(inputs)

    def _check_inverse_cache(self):
        if self.cache.inverse is None and self.cache.logabsdet is None:
            self.cache.inverse, self.cache.logabsdet = self.weight_inverse_and_logabsdet()
        elif self.cache.inverse is None:
            self.cache.inverse = self.weight_inverse()
        elif self.cache.logabsdet is None:
            self.cache.logabsdet = self.logabsdet()

    def train(self, mode=True):
        if mode:
            # If training again, invalidate cache.
            self.cache.invalidate()
        return super().train(mode)

    def use_cache(self, mode=True):
        if not check.is_bool(mode):
            raise TypeError("Mode must be boolean.")
        self.using_cache = mode

    def weight_and_logabsdet(self):
        return self.weight(), self.logabsdet()

    def weight_inverse_and_logabsdet(self):
        return self.weight_inverse(), self.logabsdet()

    def forward_no_cache(self, inputs):
        """Applies `forward` method without using the cache."""
        raise NotImplementedError()

    def inverse_no_cache(self, inputs):
        """Applies `inverse` method without using the cache."""
        raise NotImplementedError()

    def weight(self):
        """Returns the weight matrix."""
        raise NotImplementedError()

    def weight_inverse(self):
        """Returns the inverse weight matrix."""
        raise NotImplementedError()

    def logabsdet(self):
        """Returns the log absolute determinant of the weight matrix."""
        raise NotImplementedError()
