"""Normal base distributions (API of flowcon/distributions/normal.py:11-175).

``_log_prob`` is one HIP row-reduction kernel (after a point-wise / per-sample affine kernel for the diagonal forms);
``_sample`` is ``torch.randn`` on the device.
"""
import numpy as np
import torch
from torch import nn

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


def _log_z(shape):
    return float(np.float32(0.5 * np.prod(shape) * np.log(2 * np.pi)))


class ConditionalDiagonalNormal(Distribution):
    """A diagonal multivariate Normal whose parameters are functions of a context (normal.py:53-126): the context
    encoder returns ``[means | log_stds]`` per sample."""

    def __init__(self, shape, context_encoder=None):
        super().__init__()
        self._shape = torch.Size(shape)
        self._context_encoder = (lambda x: x) if context_encoder is None else context_encoder
        self.register_buffer("_log_z", torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
                             persistent=False)
        self._log_z_host = _log_z(shape)

    def _compute_params(self, context):
        if context is None:
            raise ValueError("Context can't be None.")
        params = self._context_encoder(context)
        if params.shape[-1] % 2 != 0:
            raise RuntimeError("The context encoder must return a tensor whose last dimension is even.")
        if params.shape[0] != context.shape[0]:
            raise RuntimeError("The batch dimension of the parameters is inconsistent with the input.")
        split = params.shape[-1] // 2
        means = params[..., :split].reshape(params.shape[0], *self._shape)
        log_stds = params[..., split:].reshape(params.shape[0], *self._shape)
        return means, log_stds

    def _log_prob(self, inputs, context):
        if inputs.shape[1:] != self._shape:
            raise ValueError("Expected input of shape {}, got {}".format(self._shape, inputs.shape[1:]))
        means, log_stds = self._compute_params(context)
        assert means.shape == inputs.shape and log_stds.shape == inputs.shape
        n = inputs.shape[0]
        # (x - mean) / std and -sum(log std) from the per-sample affine kernel in its inverse direction
        rows = torch.cat((means.reshape(n, -1), torch.exp(log_stds).reshape(n, -1)), dim=1)
        norm, neg_log_std_sum = ops.affine_coupling(inputs.reshape(n, -1), rows, None, activation=ops.AFFINE_SCALE_GIVEN,
                                                    inverse=True)
        return ops.standard_normal_log_prob(norm, self._log_z_host, add=neg_log_std_sum)

    def _sample(self, num_samples, context):
        means, log_stds = self._compute_params(context)
        means = torchutils.repeat_rows(means, num_samples)
        stds = torchutils.repeat_rows(torch.exp(log_stds), num_samples)
        context_size = context.shape[0]
        noise = torch.randn(context_size * num_samples, *self._shape, device=means.device)
        n = noise.shape[0]
        rows = torch.cat((means.reshape(n, -1), stds.reshape(n, -1)), dim=1)
        samples, _ = ops.affine_coupling(noise.reshape(n, -1), rows, None, activation=ops.AFFINE_SCALE_GIVEN)
        return torchutils.split_leading_dim(samples.reshape(n, *self._shape), [context_size, num_samples])

    def _mean(self, context):
        means, _ = self._compute_params(context)
        return means


class DiagonalNormal(Distribution):
    """A diagonal multivariate Normal with trainable parameters ``mean_`` / ``log_std_`` [1, D] (normal.py:129-175)."""

    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.mean_ = nn.Parameter(torch.zeros(shape).reshape(1, -1))
        self.log_std_ = nn.Parameter(torch.zeros(shape).reshape(1, -1))
        self.register_buffer("_log_z", torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
                             persistent=False)
        self._log_z_host = _log_z(shape)

    def _log_prob(self, inputs, context):
        if inputs.shape[1:] != self._shape:
            raise ValueError("Expected input of shape {}, got {}".format(self._shape, inputs.shape[1:]))
        n = inputs.shape[0]
        norm = ops.pointwise_affine_autograd(inputs.reshape(n, -1), torch.exp(self.log_std_[0]), self.mean_[0],
                                             inverse=True)
        return ops.standard_normal_log_prob(norm, self._log_z_host) - self.log_std_.sum()

    def _sample(self, num_samples, context):
        raise NotImplementedError()

    def _mean(self, context):
        return self.mean_
