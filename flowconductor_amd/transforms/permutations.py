"""Feature permutations (same classes, constructor arguments and ``_permutation`` buffer as
flowcon/transforms/permutations.py:10-64).  The gather itself is the ``fc_permute`` HIP kernel: bit-exact, zero
log-determinant."""
import torch

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.utils import typechecks as check


def _need_feature_count(features):
    if not check.is_positive_int(features):
        raise ValueError("Number of features must be a positive integer.")
    return features


class Permutation(Transform):
    """Reorders ``inputs`` along ``dim`` by a fixed index vector; the inverse applies the argsort of that vector."""

    def __init__(self, permutation, dim=1):
        index = torch.as_tensor(permutation)
        if index.dim() != 1:
            raise ValueError("Permutation must be a 1D tensor.")
        if not check.is_positive_int(dim):
            raise ValueError("dim must be a positive integer.")
        super().__init__()
        self._dim = dim
        self.register_buffer("_permutation", index)

    @property
    def _inverse_permutation(self):
        # recomputed per call like the reference (:23-25): the buffer may have been replaced by load_state_dict
        return torch.argsort(self._permutation)

    def _gather(self, inputs, index):
        dim = self._dim
        if inputs.dim() <= dim:
            raise ValueError("No dimension {} in inputs.".format(dim))
        if inputs.shape[dim] != index.numel():
            raise ValueError("Dimension {} in inputs must be of size {}.".format(dim, index.numel()))
        return ops.permute(inputs, index, dim), inputs.new_zeros(inputs.shape[0])

    def _index32(self, device):
        """The permutation as the int32 device vector the kernel takes, converted once per buffer version."""
        perm = self._permutation
        key = ops.cache_key(perm, extra=(device,))
        memo = self.__dict__.get("_cols_cache")
        if memo is None or memo[0] != key:
            memo = self.__dict__["_cols_cache"] = (key, perm.to(device=device, dtype=torch.int32).contiguous())
        return memo[1]

    def _apply_accumulate(self, inputs, context, inverse, total):
        """CompositeTransform fast path: logabsdet is identically zero -- nothing to allocate or add (two tiny launches per
        permutation layer of a small-batch flow)."""
        if inverse or self._dim != 1 or inputs.dim() != 2 or not inputs.is_cuda:
            outputs, _ = self.inverse(inputs, context) if inverse else self.forward(inputs, context)
            return outputs
        if inputs.shape[1] != self._permutation.numel():
            raise ValueError("Dimension {} in inputs must be of size {}.".format(1, self._permutation.numel()))
        return ops.permute(inputs, self._index32(inputs.device), 1)

    def forward(self, inputs, context=None):
        return self._gather(inputs, self._permutation)

    def inverse(self, inputs, context=None):
        return self._gather(inputs, self._inverse_permutation)


class RandomPermutation(Permutation):
    """A permutation drawn once at construction (from torch's global generator, as the reference does)."""

    def __init__(self, features, dim=1):
        super().__init__(torch.randperm(_need_feature_count(features)), dim)


class ReversePermutation(Permutation):
    """Feature order reversed."""

    def __init__(self, features, dim=1):
        count = _need_feature_count(features)
        super().__init__(torch.arange(count - 1, -1, -1), dim)
