"""Orthogonal transforms built from Householder reflections (API of flowcon/transforms/orthogonal.py)."""

import torch
from torch import nn

from flowconductor_amd import ops, options
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.utils import typechecks as check


def _paired_unit_vectors(num_transforms, features):
    """Initial q-vectors of the reference (orthogonal.py:40-61): e_0, e_0, e_1, e_1, ... (each unit
    vector twice, so the product of reflections starts as the identity), plus e_{K//2} when K is odd."""
    half = num_transforms // 2
    qv = torch.eye(half, features).repeat_interleave(2, dim=0) if half > 0 else torch.zeros(0, features)
    if num_transforms % 2 != 0:
        last = torch.zeros(1, features)
        last[0, half] = 1
        qv = torch.cat((qv, last))
    return qv


def _apply(inputs, q_vectors, reverse):
    return ops.householder_autograd(inputs, q_vectors, reverse=reverse)


class HouseholderSequence(Transform):
    """A sequence of Householder transforms with a learnable ``q_vectors [K, D]`` parameter."""

    _HIP_AUTOGRAD = True

    def __init__(self, features, num_transforms):
        if not check.is_positive_int(features):
            raise TypeError("Number of features must be a positive integer.")
        if not check.is_positive_int(num_transforms):
            raise TypeError("Number of transforms must be a positive integer.")
        super().__init__()
        self.features = features
        self.num_transforms = num_transforms
        self.q_vectors = nn.Parameter(_paired_unit_vectors(num_transforms, features))

    def _dense(self, inputs, reverse):
        """Batch-independent reflections fold into one orthogonal [D, D] matrix (formed in float64, cached per
        parameter version); wide batches then go through the matrix cores (``fc_dense_mm``)."""
        n = inputs.shape[0]
        if not (inputs.dim() == 2 and inputs.is_cuda and ops.sylvester_mm_supported(n, self.features)
                and n >= 1024   # enough rows to pay for the fold
                and options.get("sylvester_mm")
                and not (torch.is_grad_enabled() and (inputs.requires_grad or self.q_vectors.requires_grad))):
            return None
        key = ops.cache_key(self.q_vectors)
        if getattr(self, "_dense_cache", None) is None or self._dense_cache[0] != key:
            self._dense_cache = (key, {})
        mats = self._dense_cache[1]
        if reverse not in mats:
            # householder(v, q, reverse) == v @ M  ->  as a column map the weight is M^T
            mats[reverse] = ops.householder_matrix(self.q_vectors, reverse=reverse).T.float().contiguous()
        body = n - n % ops.SYLVESTER_MM_ROWS
        out = ops.dense_mm(inputs[:body], mats[reverse])
        if body < n:
            out = torch.cat((out, ops.householder(inputs[body:], self.q_vectors, reverse=reverse)))
        return out, inputs.new_zeros(n)

    def forward(self, inputs, context=None):
        dense = self._dense(inputs, False)
        return dense if dense is not None else _apply(inputs, self.q_vectors, reverse=False)

    def inverse(self, inputs, context=None):
        # each reflection is its own inverse: apply them in reverse order
        dense = self._dense(inputs, True)
        return dense if dense is not None else _apply(inputs, self.q_vectors, reverse=True)

    def matrix(self):
        """The [D, D] orthogonal matrix of the whole sequence (inverse applied to the identity)."""
        identity = torch.eye(self.features, self.features, device=self.q_vectors.device)
        outputs, _ = self.inverse(identity)
        return outputs


class ParametrizedHouseHolder(Transform):
    """Householder sequence with externally supplied q-vectors: ``[K, D]`` or per-sample ``[N, K, D]``."""

    def __init__(self, q_vectors):
        super().__init__()
        self.features = q_vectors.shape[-1]
        self.num_transforms = q_vectors.shape[-2]
        self.q_vectors = q_vectors
        self.reverse_idx = torch.arange(self.num_transforms - 1, -1, -1).to(q_vectors.device)

    def forward(self, inputs, context=None):
        return _apply(inputs, self.q_vectors, reverse=False)

    def inverse(self, inputs, context=None):
        return _apply(inputs, self.q_vectors, reverse=True)

    def matrix(self):
        """[D, D] (or per-sample [N, D, D]) matrix: row i is the inverse applied to e_i."""
        identity = torch.eye(self.features, self.features).to(self.q_vectors.device)
        if len(self.q_vectors.shape) > 2:
            identity = torch.repeat_interleave(identity[None, ...], self.q_vectors.shape[0], 0)
        rows = []
        for i in range(self.features):
            out, _ = self.inverse(identity[..., i, :].reshape(-1, self.features))
            rows.append(out.reshape(identity[..., i, :].shape).unsqueeze(-2))
        return torch.cat(rows, -2)
