"""LU-parameterised linear transform (API of flowcon/transforms/lu.py:10-129)."""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn import init

from flowconductor_amd import ops
from flowconductor_amd.transforms.linear import Linear


class LULinear(Linear):
    """W = L U with unit-lower L and an upper U whose diagonal is softplus(.) + eps.

    Parameters ``lower_entries``, ``upper_entries`` (strict triangles, row-major), ``unconstrained_upper_diag``
    and ``bias`` keep the reference's names and shapes."""

    _HIP_AUTOGRAD = True

    def __init__(self, features, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__(features, using_cache)
        self.eps = eps
        self.lower_indices = np.tril_indices(features, k=-1)
        self.upper_indices = np.triu_indices(features, k=1)
        self.diag_indices = np.diag_indices(features)
        n_triangular_entries = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.upper_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        self._initialize(identity_init)

    def _initialize(self, identity_init):
        init.zeros_(self.bias)
        if identity_init:
            init.zeros_(self.lower_entries)
            init.zeros_(self.upper_entries)
            init.constant_(self.unconstrained_upper_diag, np.log(np.exp(1 - self.eps) - 1))
        else:
            stdv = 1.0 / np.sqrt(self.features)
            init.uniform_(self.lower_entries, -stdv, stdv)
            init.uniform_(self.upper_entries, -stdv, stdv)
            init.uniform_(self.unconstrained_upper_diag, -stdv, stdv)

    def _create_lower_upper(self):
        lower = self.lower_entries.new_zeros(self.features, self.features)
        lower[self.lower_indices[0], self.lower_indices[1]] = self.lower_entries
        lower[self.diag_indices[0], self.diag_indices[1]] = 1.0
        upper = self.upper_entries.new_zeros(self.features, self.features)
        upper[self.upper_indices[0], self.upper_indices[1]] = self.upper_entries
        upper[self.diag_indices[0], self.diag_indices[1]] = self.upper_diag
        return lower, upper

    def _needs_grad(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward_no_cache(self, inputs):
        """outputs = L (U x) + bias in one kernel; logabsdet = sum log diag(U)."""
        if self._needs_grad(inputs):    # training: same kernel behind an autograd node, L / U built differentiably
            lower, upper = self._create_lower_upper()
            outputs = ops.lu_linear_autograd(inputs, lower, upper, self.bias)
            return outputs, self.logabsdet() * inputs.new_ones(outputs.shape[0])
        with torch.no_grad():
            lower, upper = self._create_lower_upper()
            n = inputs.shape[0]
            if (inputs.dim() == 2 and inputs.is_cuda and ops.sylvester_mm_supported(n, self.features)
                    and n % ops.SYLVESTER_MM_ROWS == 0 and n >= 1024):
                # W = L U formed once in float64; the batch goes through the matrix cores
                weight = (lower.double() @ upper.double()).float()
                outputs = ops.dense_mm(inputs, weight, self.bias)
            else:
                outputs = ops.linear(inputs, upper, lower, self.bias, mode=ops.LINEAR_LU_FORWARD)
            logabsdet = self.logabsdet() * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet

    def inverse_no_cache(self, inputs):
        """outputs = U^-1 L^-1 (x - bias) by forward/back substitution in one kernel."""
        if self._needs_grad(inputs):
            lower, upper = self._create_lower_upper()
            outputs = ops.lu_linear_autograd(inputs, lower, upper, self.bias, inverse=True)
            return outputs, -self.logabsdet() * inputs.new_ones(outputs.shape[0])
        with torch.no_grad():
            lower, upper = self._create_lower_upper()
            outputs = ops.linear(inputs, upper, lower, self.bias, mode=ops.LINEAR_LU_INVERSE)
            logabsdet = -self.logabsdet() * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet

    def weight(self):
        lower, upper = self._create_lower_upper()
        return lower @ upper

    def weight_inverse(self):
        lower, upper = self._create_lower_upper()
        identity = torch.eye(self.features, self.features, device=self.lower_entries.device)
        lower_inverse = torch.linalg.solve_triangular(lower, identity, upper=False, unitriangular=True)
        return torch.linalg.solve_triangular(upper, lower_inverse, upper=True, unitriangular=False)

    @property
    def upper_diag(self):
        return F.softplus(self.unconstrained_upper_diag) + self.eps

    def logabsdet(self):
        return torch.sum(torch.log(self.upper_diag))
