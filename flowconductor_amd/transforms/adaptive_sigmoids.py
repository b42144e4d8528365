"""Sum-of-sigmoids monotone transform (API of flowcon/transforms/adaptive_sigmoids.py:13-142)."""
import torch
import torch.nn as nn

from flowconductor_amd import ops
from flowconductor_amd.transforms.no_analytic_inv.base import MonotonicTransform
from flowconductor_amd.transforms.nonlinearities import ExtendedSoftplus


class SumOfSigmoids(MonotonicTransform):
    """Element-wise sum of S shifted/scaled sigmoids plus an extended softplus (linear far from the
    origin); inverse by bisection + Newton.  Both directions are one fused HIP kernel.

    Parameters keep the reference's names: ``shift_preact``, ``log_scale_preact``, ``raw_softmax``
    (``[1, F, S]``), ``extended_softplus.shift`` (``[1, F]``), ``log_scale_postact`` (``[1]``, frozen).
    With ``raw_params`` (``[N, F, 3S+1]``) the same tensors are per-sample views of it."""

    _HIP_AUTOGRAD = True    # forward kernel behind an autograd node (gradients from the same map in torch ops)

    PREACT_SCALE_MIN = .1
    PREACT_SCALE_MAX = 10.
    PREACT_SHIFT_MAX = 10

    def __init__(self, features, n_sigmoids=10, iterations_bisection_inverse=50, lim_bisection_inverse=120,
                 raw_params: torch.Tensor = None):
        self.n_sigmoids = n_sigmoids
        self.features = features
        super().__init__(num_iterations=iterations_bisection_inverse, lim=lim_bisection_inverse)
        self._raw = None
        if raw_params is None:
            self.shift_preact = nn.Parameter(torch.randn(1, features, self.n_sigmoids), requires_grad=True)
            self.log_scale_preact = nn.Parameter(torch.zeros(1, features, self.n_sigmoids), requires_grad=True)
            self.raw_softmax = nn.Parameter((torch.ones(1, features, self.n_sigmoids, requires_grad=False)))
            self.extended_softplus = ExtendedSoftplus(features=features)
        else:
            assert raw_params.shape[1:] == (features, 3 * self.n_sigmoids + 1)
            self.set_raw_params(features, raw_params)
        self.log_scale_postact = nn.Parameter(torch.log(torch.ones(1, device=self.shift_preact.device)),
                                              requires_grad=False)
        self.eps = 1e-6

    def get_raw_params(self):
        """All raw parameters concatenated: ``[-1, features, 3*n_sigmoids + 1]``."""
        return torch.cat((self.shift_preact.reshape(-1, self.features, self.n_sigmoids),
                          self.log_scale_preact.reshape(-1, self.features, self.n_sigmoids),
                          self.raw_softmax.reshape(-1, self.features, self.n_sigmoids),
                          self.extended_softplus.shift.reshape(-1, self.features, 1)), dim=-1)

    def set_raw_params(self, features, raw_params):
        vals = torch.split(raw_params, [self.n_sigmoids, self.n_sigmoids, self.n_sigmoids, 1], dim=-1)
        self.shift_preact, self.log_scale_preact, self.raw_softmax = vals[:3]
        self.extended_softplus = ExtendedSoftplus(features=features, shift=vals[3])
        self._raw = raw_params

    def _kernel(self, inputs, inverse, offset=0.0):
        raw = self._raw if self._raw is not None else self.get_raw_params()
        shared = raw.shape[0] == 1
        if not shared and raw.shape[0] != inputs.shape[0]:
            raise ValueError("raw_params batch %d != inputs batch %d" % (raw.shape[0], inputs.shape[0]))
        return ops.sum_of_sigmoids_autograd(inputs, raw, self.n_sigmoids, inverse=inverse, offset=offset,
                                            iterations=self.num_iterations, lim=self.lim, shared_params=shared)

    def forward(self, inputs, context=None):
        return self._kernel(inputs, inverse=False)

    def inverse(self, inputs, context=None, forward_function=None):
        if forward_function is not None:
            return super().inverse(inputs, context=context, forward_function=forward_function)
        return self._kernel(inputs, inverse=True)
