"""Planar and Sylvester flows (API of flowcon/transforms/no_analytic_inv/planar.py:13-169).

Forward direction only, like the reference.  Each is one fused row-per-wavefront HIP kernel.
"""

import numpy as np
import torch
from torch import nn
from torch.nn import init

from flowconductor_amd import ops, options
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.transforms.orthogonal import HouseholderSequence


class PlanarTransform(Transform):
    """f(z) = z + u_hat * tanh(w.z + b) with u constrained so that w.u_hat >= -1 (invertibility)."""

    def __init__(self, features: int = 2, num_iterations=25, lim=50):
        super().__init__()
        self.w = nn.Parameter(torch.randn(1, features).normal_(0, 0.1))
        self.b = nn.Parameter(torch.randn(1).normal_(0, 0.1))
        self.u = nn.Parameter(torch.randn(1, features).normal_(0, 0.1))

    _HIP_AUTOGRAD = True

    def forward(self, inputs, context=None):
        return ops.planar_autograd(inputs, self.w, self.get_constrained_u(), self.b)

    def forward_logabsdet(self, inputs, context=None):
        return self.forward(inputs, context)[1].unsqueeze(-1)

    def get_constrained_u(self):
        """u + (softplus(w.u) - 1 - w.u) * w / |w|^2   (a [1, D] host-side expression)."""
        wtu = torch.mm(self.u, self.w.T)
        m_wtu = -1 + torch.nn.functional.softplus(wtu)
        w_direction = self.w / (torch.norm(self.w, p=2, dim=1) ** 2)
        return self.u + (m_wtu - wtu) * w_direction


class SylvesterTransform(Transform):
    """z + Q R2 tanh(R1 Q^T z + b), Q from ``num_householder`` reflections, R1/R2 upper triangular with
    tanh-squashed diagonals.  Parameter names follow the reference (``upper_entries1/2``,
    ``log_upper_diag1/2``, ``Q_orth.q_vectors``, ``bias``).  ``device`` defaults to "cuda" as there."""

    def __init__(self, features: int = 2, num_householder=None, device="cuda"):
        super().__init__()
        self.n_diag_entries = features
        self.n_triangular_entries = ((features - 1) * features) // 2
        self.features = features
        if num_householder is None:
            num_householder = self.features
        self.num_householder = num_householder
        self.upper_indices = np.triu_indices(features, k=1)
        self.diag_indices = np.diag_indices(features)
        self.upper_entries1 = nn.Parameter(torch.zeros(self.n_triangular_entries))
        self.log_upper_diag1 = nn.Parameter(torch.zeros(features))
        self.upper_entries2 = nn.Parameter(torch.zeros(self.n_triangular_entries))
        self.log_upper_diag2 = nn.Parameter(torch.zeros(features))
        self.Q_orth = HouseholderSequence(features=features, num_transforms=self.num_householder)
        if device is not None and (torch.device(device).type != "cuda" or torch.cuda.is_available()):
            self.Q_orth = self.Q_orth.to(device=device)
        self.bias = nn.Parameter(torch.zeros(features))
        self._initialize()

    def _initialize(self):
        stdv = 1.0 / np.sqrt(self.features)
        init.uniform_(self.upper_entries1, -stdv, stdv)
        init.uniform_(self.upper_entries2, -stdv, stdv)
        init.uniform_(self.log_upper_diag1, -stdv, stdv)
        init.uniform_(self.log_upper_diag2, -stdv, stdv)
        init.constant_(self.bias, 0.0)

    def _create_R(self, entries, log_diag):
        upper = entries.new_zeros(self.features, self.features)
        upper[self.upper_indices[0], self.upper_indices[1]] = entries
        upper[self.diag_indices[0], self.diag_indices[1]] = torch.tanh(log_diag)
        return upper

    def _create_R1(self):
        return self._create_R(self.upper_entries1, self.log_upper_diag1)

    def _create_R2(self):
        return self._create_R(self.upper_entries2, self.log_upper_diag2)

    def dh_dx(self, x):
        return 1 - torch.tanh(x) ** 2

    def h(self, x):
        return torch.tanh(x)

    def _mm_weights(self):
        """(W1, W2, r_diag_prod) for the matrix-core kernel, recomputed only when a parameter changed."""
        params = (self.Q_orth.q_vectors, self.upper_entries1, self.log_upper_diag1, self.upper_entries2,
                  self.log_upper_diag2)
        key = ops.cache_key(*params)
        if getattr(self, "_mm_cache", None) is None or self._mm_cache[0] != key:
            self._mm_cache = (key, ops.pack_sylvester(self.Q_orth.q_vectors, self._create_R1(), self._create_R2()))
        return self._mm_cache[1]

    _HIP_AUTOGRAD = True

    def forward(self, inputs, context=None):
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            return ops.sylvester_autograd(inputs, self.Q_orth.q_vectors, self._create_R1(), self._create_R2(), self.bias)
        with torch.no_grad():
            n = inputs.shape[0]
            if (inputs.dim() == 2 and inputs.is_cuda and ops.sylvester_mm_supported(n, self.features)
                    and options.get("sylvester_mm")):
                # batch-independent parameters: the Householder / triangular chains fold into two dense
                # [D, D] matrices and the batch goes through the matrix cores
                w1, w2, rdiag = self._mm_weights()
                body = n - n % ops.SYLVESTER_MM_ROWS
                y, lad = ops.sylvester_mm(inputs[:body], w1, w2, self.bias, rdiag)
                if body < n:
                    y2, lad2 = ops.sylvester(inputs[body:], self.Q_orth.q_vectors, self._create_R1(),
                                             self._create_R2(), self.bias)
                    y, lad = torch.cat((y, y2)), torch.cat((lad, lad2))
                return y, lad
            return ops.sylvester(inputs, self.Q_orth.q_vectors, self._create_R1(), self._create_R2(), self.bias)

    def inverse(self, inputs, context=None):
        raise ops.InverseNotAvailable()
