"""``Flow``: a bijector stack plus a base density, exposed as a ``Distribution``.

Protocol (flowcon/flows/base.py:11-119, restated from SURVEY.md section 3): data -> noise is the transform's
forward direction, so

    log p(x | c)          = log p_base(f(x; e(c)) [| e(c)]) + logabsdet f
    sample(n | c)         = f^-1(noise; e(c)),  noise ~ p_base, n draws for every context row
    sample_and_log_prob   = the same draws with  log p_base(noise) - logabsdet f^-1

with e = ``embedding_net`` (identity when None).  Whether the base density itself takes a context is decided once,
from the signature of its ``log_prob``.  Sub-module attribute names (``_transform``, ``_distribution``,
``_embedding_net``) are the reference's, so its checkpoints load.
"""
import inspect

from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.distributions.base import Distribution
from flowconductor_amd.distributions.normal import StandardNormal

__all__ = ["Flow"]


def _fold_draws(t, groups, draws):
    """[groups * draws, ...] -> [groups, draws, ...]."""
    return t.reshape((groups, draws) + tuple(t.shape[1:]))


class Flow(Distribution):
    """Base class for all flow objects."""

    def __init__(self, transform, distribution, embedding_net=None):
        super().__init__()
        if embedding_net is not None and not isinstance(embedding_net, nn.Module):
            raise AssertionError("embedding_net is not a nn.Module. If you want to use hard-coded summary features, "
                                 "please simply pass the encoded features and pass embedding_net=None")
        self._transform = transform
        self._distribution = distribution
        self._embedding_net = nn.Identity() if embedding_net is None else embedding_net
        self._context_used_in_base = "context" in inspect.signature(distribution.log_prob).parameters

    # -- pieces shared by the four entry points ------------------------------------------------------------------
    def _base_kwargs(self, embedded):
        return {"context": embedded} if self._context_used_in_base else {}

    def _noise_for(self, num_samples, embedded, with_log_prob):
        """Base draws laid out one row per (context row, draw): ``(noise [B * num_samples, ...], log_prob or None)``;
        without a context B == 1 and the rows are just the draws."""
        base = self._distribution
        if with_log_prob:
            noise, log_prob = base.sample_and_log_prob(num_samples, **self._base_kwargs(embedded))
        elif self._context_used_in_base or embedded is None:
            noise, log_prob = base.sample(num_samples, **self._base_kwargs(embedded)), None
        else:
            # a context-free base under a context: draw B * num_samples at once
            return base.sample(num_samples * embedded.shape[0]), None
        if embedded is not None:       # [B, num_samples, ...] -> rows
            noise = noise.reshape((-1,) + tuple(noise.shape[2:]))
        return noise, log_prob

    def _invert(self, noise, embedded, num_samples):
        """Noise rows -> data rows through the inverse stack, every context row repeated for its draws."""
        rows_context = None if embedded is None else embedded.repeat_interleave(num_samples, dim=0)
        return self._transform.inverse(noise, context=rows_context)

    # -- Distribution interface -----------------------------------------------------------------------------------
    def _log_prob(self, inputs, context):
        embedded = self._embedding_net(context)
        with ops.deferred_errors():       # one read of the device error word for the whole stack
            noise, logabsdet = self._transform(inputs, context=embedded)
        base = self._distribution
        if isinstance(base, StandardNormal):
            return base.log_prob_plus(noise, logabsdet)      # `+ logabsdet` folded into the reduction kernel
        return base.log_prob(noise, **self._base_kwargs(embedded)) + logabsdet

    def _sample(self, num_samples, context):
        embedded = self._embedding_net(context)
        noise, _ = self._noise_for(num_samples, embedded, with_log_prob=False)
        samples, _ = self._invert(noise, embedded, num_samples)
        return samples if embedded is None else _fold_draws(samples, embedded.shape[0], num_samples)

    def sample_and_log_prob(self, num_samples, context=None):
        """Draws and their log-densities from ONE pass through the inverse stack."""
        embedded = self._embedding_net(context)
        noise, base_log_prob = self._noise_for(num_samples, embedded, with_log_prob=True)
        samples, logabsdet = self._invert(noise, embedded, num_samples)
        if embedded is not None:
            groups = embedded.shape[0]
            samples = _fold_draws(samples, groups, num_samples)
            logabsdet = _fold_draws(logabsdet, groups, num_samples)
        return samples, base_log_prob - logabsdet

    def transform_to_noise(self, inputs, context=None):
        """``f(inputs; e(context))`` without the density (goodness-of-fit checks against the base)."""
        return self._transform(inputs, context=self._embedding_net(context))[0]
