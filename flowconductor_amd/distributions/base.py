"""The ``Distribution`` protocol of the flow's base densities.

Protocol (flowcon/distributions/base.py:16-128, restated from SURVEY.md section 8b): subclasses supply
``_log_prob(inputs, context)``, ``_sample(num_samples, context)`` and optionally ``_mean(context)``; the public
methods validate arguments and fix the shapes --

    log_prob(inputs [N, ...], context [N, ...] | None)      -> [N]
    sample(n, context [B, ...] | None, batch_size=None)     -> [n, ...]  or  [B, n, ...]
    sample_and_log_prob(n, context)                          -> the same draws + their log-densities
    mean(context)

A ``Distribution`` is an ``nn.Module`` only to own parameters: calling it raises.
"""
import torch
from torch import nn

from flowconductor_amd.utils import typechecks as check


class NoMeanException(Exception):
    """Exception to be thrown when a mean function doesn't exist."""


def _tensor_or_none(value):
    return None if value is None else torch.as_tensor(value)


class Distribution(nn.Module):
    """Base class for all distribution objects."""

    def forward(self, *args):
        raise RuntimeError("Forward method cannot be called for a Distribution object.")

    # -- density ------------------------------------------------------------------------------------------------
    def log_prob(self, inputs, context=None):
        inputs, context = torch.as_tensor(inputs), _tensor_or_none(context)
        if context is not None and context.shape[0] != inputs.shape[0]:
            raise ValueError("Number of input items must be equal to number of context items.")
        return self._log_prob(inputs, context)

    _log_prob = check.abstract("_log_prob", "(inputs [N, ...], context or None) -> log-density [N]")

    # -- sampling -----------------------------------------------------------------------------------------------
    def sample(self, num_samples, context=None, batch_size=None):
        check.need_positive_int(num_samples, "Number of samples")
        context = _tensor_or_none(context)
        if batch_size is None:        # one call
            return self._sample(num_samples, context)
        check.need_positive_int(batch_size, "Batch size")
        # draw in pieces of at most batch_size (bounds the activation memory of a deep inverse stack)
        sizes = [batch_size] * (num_samples // batch_size)
        if num_samples % batch_size:
            sizes.append(num_samples % batch_size)
        return torch.cat([self._sample(size, context) for size in sizes], dim=0)

    _sample = check.abstract("_sample", "(num_samples, context or None) -> [n, ...] or [B, n, ...] draws")

    def sample_and_log_prob(self, num_samples, context=None):
        """Generic version: draw, then evaluate (subclasses with a cheaper joint form override it)."""
        draws = self.sample(num_samples, context=context)
        if context is None:
            return draws, self.log_prob(draws)
        # [B, n, ...] draws -> B * n rows next to their (repeated) context rows, and back
        groups = draws.shape[0]
        rows = draws.reshape((groups * num_samples,) + tuple(draws.shape[2:]))
        rows_context = torch.as_tensor(context).repeat_interleave(num_samples, dim=0)
        if rows.shape[0] != rows_context.shape[0]:
            raise AssertionError("sample() returned %d groups for %d context rows" % (groups, len(context)))
        log_prob = self.log_prob(rows, context=rows_context)
        return draws, log_prob.reshape(groups, num_samples)

    # -- moments ------------------------------------------------------------------------------------------------
    def mean(self, context=None):
        return self._mean(_tensor_or_none(context))

    _mean = check.abstract("_mean", "(context or None) -> mean; absent by default", exception=NoMeanException)
