"""Flow = transform + base distribution (API of flowcon/flows/base.py:11-119)."""
from inspect import signature

import torch.nn

from flowconductor_amd import ops
from flowconductor_amd.distributions.base import Distribution
from flowconductor_amd.distributions.normal import StandardNormal
from flowconductor_amd.utils import torchutils

__all__ = ["Flow"]


class Flow(Distribution):
    """Base class for all flow objects."""

    def __init__(self, transform, distribution, embedding_net=None):
        """
        Args:
            transform: A `Transform` object, it transforms data into noise.
            distribution: A `Distribution` object, the base distribution of the flow.
            embedding_net: A `nn.Module` encoding the context (trained jointly), or None.
        """
        super().__init__()
        self._transform = transform
        self._distribution = distribution
        self._context_used_in_base = "context" in signature(self._distribution.log_prob).parameters
        if embedding_net is not None:
            assert isinstance(embedding_net, torch.nn.Module), (
                "embedding_net is not a nn.Module. "
                "If you want to use hard-coded summary features, "
                "please simply pass the encoded features and pass "
                "embedding_net=None"
            )
            self._embedding_net = embedding_net
        else:
            self._embedding_net = torch.nn.Identity()

    def _log_prob(self, inputs, context):
        embedded_context = self._embedding_net(context)
        with ops.deferred_errors():
            noise, logabsdet = self._transform(inputs, context=embedded_context)
        if isinstance(self._distribution, StandardNormal):
            # the context value is ignored by StandardNormal; fold `+ logabsdet` into its kernel
            return self._distribution.log_prob_plus(noise, logabsdet)
        if self._context_used_in_base:
            log_prob = self._distribution.log_prob(noise, context=embedded_context)
        else:
            log_prob = self._distribution.log_prob(noise)
        return log_prob + logabsdet

    def _sample(self, num_samples, context):
        embedded_context = self._embedding_net(context)
        if self._context_used_in_base:
            noise = self._distribution.sample(num_samples, context=embedded_context)
        else:
            repeat_noise = self._distribution.sample(num_samples * embedded_context.shape[0])
            noise = torch.reshape(repeat_noise, (embedded_context.shape[0], -1, repeat_noise.shape[1]))

        if embedded_context is not None:
            # Merge the context dimension with sample dimension in order to apply the transform.
            noise = torchutils.merge_leading_dims(noise, num_dims=2)
            embedded_context = torchutils.repeat_rows(embedded_context, num_reps=num_samples)

        samples, _ = self._transform.inverse(noise, context=embedded_context)

        if embedded_context is not None:
            # Split the context dimension from sample dimension.
            samples = torchutils.split_leading_dim(samples, shape=[-1, num_samples])
        return samples

    def sample_and_log_prob(self, num_samples, context=None):
        """Samples from the flow together with their log probabilities (one inverse pass)."""
        embedded_context = self._embedding_net(context)
        if self._context_used_in_base:
            noise, log_prob = self._distribution.sample_and_log_prob(num_samples, context=embedded_context)
        else:
            noise, log_prob = self._distribution.sample_and_log_prob(num_samples)

        if embedded_context is not None:
            noise = torchutils.merge_leading_dims(noise, num_dims=2)
            embedded_context = torchutils.repeat_rows(embedded_context, num_reps=num_samples)

        samples, logabsdet = self._transform.inverse(noise, context=embedded_context)

        if embedded_context is not None:
            samples = torchutils.split_leading_dim(samples, shape=[-1, num_samples])
            logabsdet = torchutils.split_leading_dim(logabsdet, shape=[-1, num_samples])
        return samples, log_prob - logabsdet

    def transform_to_noise(self, inputs, context=None):
        """Transforms data `[batch, ...]` into noise (goodness-of-fit checks)."""
        noise, _ = self._transform(inputs, context=self._embedding_net(context))
        return noise
