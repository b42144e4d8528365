"""Element-wise bijectors and batch-shared spline CDFs (API of flowcon/transforms/nonlinearities.py)."""
import numpy as np
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import CompositeTransform, InverseTransform, Transform


class CompositeCDFTransform(CompositeTransform):
    """squash -> cdf -> unsquash (nonlinearities.py:239-243)."""

    def __init__(self, squashing_transform, cdf_transform):
        super().__init__([squashing_transform, cdf_transform, InverseTransform(squashing_transform)])


def _flatten_items(inputs):
    """[N, *shape] -> [N, prod(shape)] view for the [N, D] kernels."""
    return inputs.reshape(inputs.shape[0], -1)


class PiecewiseRationalQuadraticCDF(Transform):
    """RQ spline with learnable parameters shared across the batch (nonlinearities.py:429-487).

    Parameters ``unnormalized_widths/heights/derivatives`` have shape ``[*shape, K]`` /
    ``[*shape, K -/+ 1]``; the kernel reads one concatenated ``[prod(shape), 3K -/+ 1]`` row set
    staged once per workgroup in LDS instead of an ``expand``-ed ``[N, ...]`` view.
    """

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0, identity_init=False,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        if isinstance(shape, int):
            shape = (shape,)
        num_derivatives = (num_bins - 1) if self.tails == "linear" else (num_bins + 1)
        if identity_init:
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            constant = np.log(np.exp(1 - min_derivative) - 1)
            self.unnormalized_derivatives = nn.Parameter(constant * torch.ones(*shape, num_derivatives))
        else:
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, num_derivatives))

    def _spline(self, inputs, inverse=False):
        rows = torch.cat((self.unnormalized_widths.detach(), self.unnormalized_heights.detach(),
                          self.unnormalized_derivatives.detach()), dim=-1).reshape(-1)
        flat = _flatten_items(inputs)
        outputs, logabsdet = ops.rq_spline(
            flat, rows, None, num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
            min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative, inverse=inverse, shared_params=True)
        return outputs.reshape(inputs.shape), logabsdet

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)
