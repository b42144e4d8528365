"""``Linear``: base of the transforms that parameterise an invertible [D, D] weight (LU, ...).

Protocol (flowcon/transforms/linear.py:15-126, restated from SURVEY.md 8a row U2): a subclass supplies
``weight()``, ``weight_inverse()``, ``logabsdet()`` and the uncached ``forward_no_cache`` / ``inverse_no_cache``.
With ``using_cache`` and the module in eval mode the dense weight (or its inverse) and log|det W| are computed once
and reused -- any ``train(True)`` drops them -- and the map itself is one ``fc_linear`` launch:

    forward:  y = W x + b,            logabsdet =  log|det W|   (the same value for every row)
    inverse:  y = W^-1 (x - b),       logabsdet = -log|det W|
"""
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.utils import typechecks as check


class LinearCache:
    """The three cached quantities (``None`` = not computed yet)."""

    weight = inverse = logabsdet = None

    def invalidate(self):
        self.weight = self.inverse = self.logabsdet = None


class Linear(Transform):
    """Abstract base class for linear transforms that parameterize a weight matrix."""

    def __init__(self, features, using_cache=False):
        check.need_positive_int(features, "Number of features")
        super().__init__()
        self.features, self.using_cache, self.cache = features, using_cache, LinearCache()
        self.bias = nn.Parameter(torch.zeros(features))     # state_dict key "bias", as in the reference

    # -- cache ----------------------------------------------------------------------------------------------------
    def use_cache(self, mode=True):
        self.using_cache = check.need_bool(mode, "Mode")

    def train(self, mode=True):
        if bool(mode):                    # parameters are about to change
            self.cache.invalidate()
        return super().train(mode)

    def _cache_active(self, inputs):
        wants_grad = torch.is_grad_enabled() and (inputs.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        return self.using_cache and not self.training and not wants_grad

    def _fill(self, slot, one, both):
        """Make ``cache.<slot>`` and ``cache.logabsdet`` present: the joint routine when both are missing (a
        subclass may share work between them), otherwise only the missing one."""
        cache = self.cache
        have_matrix, have_det = getattr(cache, slot) is not None, cache.logabsdet is not None
        if not have_matrix and not have_det:
            matrix, cache.logabsdet = both()
            setattr(cache, slot, matrix)
        elif not have_matrix:
            setattr(cache, slot, one())
        elif not have_det:
            cache.logabsdet = self.logabsdet()

    def _check_forward_cache(self):
        self._fill("weight", self.weight, self.weight_and_logabsdet)

    def _check_inverse_cache(self):
        self._fill("inverse", self.weight_inverse, self.weight_inverse_and_logabsdet)

    # -- the map --------------------------------------------------------------------------------------------------
    def forward(self, inputs, context=None):
        if not self._cache_active(inputs):
            return self.forward_no_cache(inputs)
        self._check_forward_cache()
        outputs = ops.linear(inputs, self.cache.weight, bias=self.bias, mode=ops.LINEAR_DENSE)
        return outputs, self.cache.logabsdet.expand(outputs.shape[0]).clone()

    def inverse(self, inputs, context=None):
        if not self._cache_active(inputs):
            return self.inverse_no_cache(inputs)
        self._check_inverse_cache()
        w_inv = self.cache.inverse
        # W^-1 (x - b) = W^-1 x - W^-1 b: the [D] constant is formed on the host side of the launch
        outputs = ops.linear(inputs, w_inv, bias=-(w_inv.detach() @ self.bias.detach()), mode=ops.LINEAR_DENSE)
        return outputs, (-self.cache.logabsdet).expand(outputs.shape[0]).clone()

    # -- what a subclass provides -----------------------------------------------------------------------------------
    def weight_and_logabsdet(self):
        """Joint form; a subclass whose two quantities share work overrides it."""
        return (self.weight(), self.logabsdet())

    def weight_inverse_and_logabsdet(self):
        return (self.weight_inverse(), self.logabsdet())

    forward_no_cache = check.abstract("forward_no_cache", "(inputs) -> (outputs, logabsdet) from the parameters")
    inverse_no_cache = check.abstract("inverse_no_cache", "(inputs) -> (outputs, logabsdet) from the parameters")
    weight = check.abstract("weight", "() -> dense [D, D] weight")
    weight_inverse = check.abstract("weight_inverse", "() -> dense [D, D] inverse weight")
    logabsdet = check.abstract("logabsdet", "() -> scalar log|det W|")
