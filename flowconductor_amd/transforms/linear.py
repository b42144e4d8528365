"""``Linear``: base of the transforms that parameterise an invertible [D, D] weight (LU, ...).

Protocol (flowcon/transforms/linear.py:15-126, restated from SURVEY.md 8a row U2): a subclass supplies
``weight()``, ``weight_inverse()``, ``logabsdet()`` and the uncached ``forward_no_cache`` / ``inverse_no_cache``.
With ``using_cache`` and the module in eval mode the dense weight (or its inverse) and log|det W| are computed once
and reused -- any ``train(True)`` drops them -- and the map itself is one ``fc_linear`` launch:

    forward:  y = W x + b,            logabsdet =  log|det W|   (the same value for every row)
    inverse:  y = W^-1 (x - b),       logabsdet = -log|det W|
"""
import numpy as np
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


class ScalarScale(Transform):
    """y = (exp(_scale) + eps) x with ONE scalar for the whole tensor (flowcon/transforms/linear.py:232-252; the
    reference's matrix/ helpers compose it).  The reference's log-determinant multiplies log(scale) by the SUM of the
    non-batch sizes (``np.sum(inputs.shape[1:])``), which equals the number of elements only for [N, D] inputs: kept."""

    _HIP_AUTOGRAD = True

    def __init__(self, scale=1.0, trainable=True, eps=1e-4):
        super().__init__()
        if not np.all(np.asarray(scale) > 1e-6):
            raise AssertionError("Scale too small..")
        self._scale = nn.Parameter(torch.log(torch.tensor(scale, dtype=torch.get_default_dtype())), requires_grad=trainable)
        self.eps = eps

    @property
    def scale(self):
        return torch.exp(self._scale) + self.eps

    def _map(self, inputs, inverse):
        scale = self.scale
        outputs = ops.pointwise_affine_autograd(inputs, scale.reshape(1), torch.zeros(1, device=inputs.device), inverse=inverse)
        logabsdet = inputs.new_ones(inputs.shape[0]) * torch.log(scale).sum() * float(np.sum(inputs.shape[1:]))
        return outputs, (-logabsdet if inverse else logabsdet)

    def forward(self, inputs, context=None):
        return self._map(inputs, False)

    def inverse(self, inputs, context=None):
        return self._map(inputs, True)


class ScalarShift(Transform):
    """y = x + shift with one scalar (flowcon/transforms/linear.py:255-266); logabsdet = 0."""

    _HIP_AUTOGRAD = True

    def __init__(self, shift=0.0, trainable=True):
        super().__init__()
        self.shift = nn.Parameter(torch.tensor(shift, dtype=torch.get_default_dtype()), requires_grad=trainable)

    def _map(self, inputs, inverse):
        # (x - b) / 1 is the kernel's inverse form: subtracting the shift
        outputs = ops.pointwise_affine_autograd(inputs, torch.ones(1, device=inputs.device), self.shift.reshape(1),
                                                inverse=inverse)
        return outputs, inputs.new_zeros(inputs.shape[0])

    def forward(self, inputs, context=None):
        return self._map(inputs, False)

    def inverse(self, inputs, context=None):
        return self._map(inputs, True)
