"""LU-parameterised linear transform: class name, constructor arguments, parameter names and shapes of
flowcon/transforms/lu.py:10-129 (reference checkpoints load); the products run in the ``fc_linear`` / ``fc_dense_mm``
HIP kernels."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from flowconductor_amd import ops
from flowconductor_amd.transforms.linear import Linear


class LULinear(Linear):
    """``y = L (U x) + bias`` with ``L`` unit-lower and ``U`` upper triangular, ``diag U = softplus(.) + eps > 0``.

    Parameters: ``lower_entries`` / ``upper_entries`` (the strict triangles, row-major, ``D (D - 1) / 2`` values each),
    ``unconstrained_upper_diag`` [D], ``bias`` [D]."""

    _HIP_AUTOGRAD = True

    def __init__(self, features, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__(features, using_cache)
        self.eps = eps
        strict = features * (features - 1) // 2
        self.lower_entries = nn.Parameter(torch.zeros(strict))
        self.upper_entries = nn.Parameter(torch.zeros(strict))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        # (row, column) positions of the strict triangles in row-major order -- the order the entries are stored in
        self._triangles = {}      # device -> index tensors of the two strict triangles (row-major, the storage order)
        with torch.no_grad():
            self.bias.zero_()
            if identity_init:
                # softplus(c) + eps = 1  <=>  c = log(exp(1 - eps) - 1): W = I at initialisation
                self.unconstrained_upper_diag.fill_(math.log(math.expm1(1.0 - eps)))
            else:
                bound = features ** -0.5
                for p in (self.lower_entries, self.upper_entries, self.unconstrained_upper_diag):
                    p.uniform_(-bound, bound)

    @property
    def upper_diag(self):
        return F.softplus(self.unconstrained_upper_diag) + self.eps

    def _create_lower_upper(self):
        """Dense (L, U) from the stored triangles (differentiable: index_put of the parameters)."""
        d = self.features
        dev = self.lower_entries.device
        if dev not in self._triangles:
            self._triangles[dev] = (tuple(torch.tril_indices(d, d, offset=-1, device=dev)),
                                    tuple(torch.triu_indices(d, d, offset=1, device=dev)))
        below, above = self._triangles[dev]
        lower = torch.eye(d, dtype=self.lower_entries.dtype, device=dev).index_put(below, self.lower_entries)
        upper = torch.diag(self.upper_diag).index_put(above, self.upper_entries)
        return lower, upper

    def logabsdet(self):
        return self.upper_diag.log().sum()

    def _per_row(self, value, rows):
        return value * value.new_ones(rows)

    def _needs_grad(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward_no_cache(self, inputs):
        """``L (U x) + bias`` in one kernel; logabsdet = sum log diag(U) for every row."""
        if self._needs_grad(inputs):    # training: the same kernel behind an autograd node, L / U built differentiably
            lower, upper = self._create_lower_upper()
            outputs = ops.lu_linear_autograd(inputs, lower, upper, self.bias)
            return outputs, self._per_row(self.logabsdet(), outputs.shape[0])
        with torch.no_grad():
            lower, upper = self._create_lower_upper()
            rows = inputs.shape[0]
            wide = (inputs.dim() == 2 and inputs.is_cuda and rows >= 1024 and rows % ops.SYLVESTER_MM_ROWS == 0
                    and ops.sylvester_mm_supported(rows, self.features))
            if wide:     # W = L U formed once in float64; the batch goes through the matrix cores
                outputs = ops.dense_mm(inputs, (lower.double() @ upper.double()).float(), self.bias)
            else:
                outputs = ops.linear(inputs, upper, lower, self.bias, mode=ops.LINEAR_LU_FORWARD)
            return outputs, self._per_row(self.logabsdet(), rows)

    def inverse_no_cache(self, inputs):
        """``U^-1 L^-1 (x - bias)`` by forward / back substitution in one kernel."""
        if self._needs_grad(inputs):
            lower, upper = self._create_lower_upper()
            outputs = ops.lu_linear_autograd(inputs, lower, upper, self.bias, inverse=True)
            return outputs, self._per_row(-self.logabsdet(), outputs.shape[0])
        with torch.no_grad():
            lower, upper = self._create_lower_upper()
            rows = inputs.shape[0]
            wide = (inputs.dim() == 2 and inputs.is_cuda and rows >= 1024 and rows % ops.SYLVESTER_MM_ROWS == 0
                    and ops.sylvester_mm_supported(rows, self.features))
            if wide:
                # batch-independent parameters: W^-1 = U^-1 L^-1 formed once by two float64 triangular solves against the
                # identity (what lu.py:70-91 does per batch, in float32), the batch goes through the matrix cores as in the
                # forward direction:  W^-1 (x - b) = W^-1 x - W^-1 b.  (The per-row substitution kernel: two dependent sweeps
                # over D per row, 4.3 ms per 2^20 x 64 against 0.11 ms.)
                eye = torch.eye(self.features, dtype=torch.float64, device=lower.device)
                l_inv = torch.linalg.solve_triangular(lower.double(), eye, upper=False, unitriangular=True)
                w_inv = torch.linalg.solve_triangular(upper.double(), l_inv, upper=True)
                outputs = ops.dense_mm(inputs, w_inv.float(), -(w_inv @ self.bias.double()).float())
            else:
                outputs = ops.linear(inputs, upper, lower, self.bias, mode=ops.LINEAR_LU_INVERSE)
            return outputs, self._per_row(-self.logabsdet(), rows)

    def weight(self):
        lower, upper = self._create_lower_upper()
        return lower @ upper

    def weight_inverse(self):
        """``U^-1 L^-1`` by two triangular solves against the identity."""
        lower, upper = self._create_lower_upper()
        eye = torch.eye(self.features, dtype=lower.dtype, device=lower.device)
        l_inv = torch.linalg.solve_triangular(lower, eye, upper=False, unitriangular=True)
        return torch.linalg.solve_triangular(upper, l_inv, upper=True)
