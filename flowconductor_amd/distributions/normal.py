"""Standard normal base distribution (API of flowcon/distributions/normal.py:11-50).

``_log_prob`` is one HIP row-reduction kernel; ``_sample`` is ``torch.randn`` on the device.
"""
import numpy as np
import torch

from flowconductor_amd import ops
from flowconductor_amd.distributions.base import Distribution
from flowconductor_amd.utils import torchutils


class StandardNormal(Distribution):
    """A multivariate Normal with zero mean and unit covariance."""

    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.register_buffer(
            "_log_z",
            torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
            persistent=False,
        )
        self._log_z_host = float(np.float32(0.5 * np.prod(shape) * np.log(2 * np.pi)))

    def _check_shape(self, inputs):
        if inputs.shape[1:] != self._shape:
            raise ValueError("Expected input of shape {}, got {}".format(self._shape, inputs.shape[1:]))

    def _log_prob(self, inputs, context):
        # Note: the context is ignored.
        self._check_shape(inputs)
        return ops.standard_normal_log_prob(inputs, self._log_z_host)

    def log_prob_plus(self, inputs, logabsdet):
        """``log_prob(inputs) + logabsdet`` in one pass (the tail of Flow._log_prob)."""
        self._check_shape(inputs)
        return ops.standard_normal_log_prob(inputs, self._log_z_host, add=logabsdet)

    def _sample(self, num_samples, context):
        if context is None:
            return torch.randn(num_samples, *self._shape, device=self._log_z.device)
        # The value of the context is ignored, only its size and device are taken into account.
        context_size = context.shape[0]
        samples = torch.randn(context_size * num_samples, *self._shape, device=context.device)
        return torchutils.split_leading_dim(samples, [context_size, num_samples])

    def _mean(self, context):
        if context is None:
            return self._log_z.new_zeros(self._shape)
        return context.new_zeros(context.shape[0], *self._shape)
