"""Transform protocol and composition (API of flowcon/transforms/base.py:10-60, 215-231).

``forward(inputs, context=None) -> (outputs, logabsdet[N])``; ``inverse`` likewise.
``CompositeTransform`` defers the device error-word read to once per cascade instead of
the reference's host sync inside every spline call.
"""
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.ops import InputOutsideDomain, InverseNotAvailable  # noqa: F401 (re-export)


class Transform(nn.Module):
    """Base class for all transform objects."""

    def forward(self, inputs, context=None):
        raise NotImplementedError()

    def inverse(self, inputs, context=None):
        raise InverseNotAvailable()


class CompositeTransform(Transform):
    """Composes several transforms into one, in the order they are given."""

    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)

    @staticmethod
    def _cascade(inputs, transforms, context, inverse):
        """base.py:44-60 of the reference.  A transform may offer ``_apply_accumulate(inputs, context, inverse,
        total)`` -> outputs, which adds its logabsdet onto the running total inside its own kernel (one
        element-wise pass over [N] saved per layer); everything else goes through forward / inverse."""
        outputs = inputs
        total_logabsdet = inputs.new_zeros(inputs.shape[0])
        with ops.deferred_errors():
            for t in transforms:
                fused = getattr(t, "_apply_accumulate", None)
                hooked = t._forward_hooks or t._forward_pre_hooks   # the fast path does not go through __call__
                if fused is not None and not hooked and total_logabsdet.dtype == torch.float32:
                    outputs = fused(outputs, context, inverse, total_logabsdet)
                else:
                    outputs, logabsdet = (t.inverse if inverse else t)(outputs, context)
                    total_logabsdet += logabsdet
        return outputs, total_logabsdet

    def forward(self, inputs, context=None):
        return self._cascade(inputs, self._transforms, context, False)

    def inverse(self, inputs, context=None):
        return self._cascade(inputs, self._transforms[::-1], context, True)


class InverseTransform(Transform):
    """Creates a transform that is the inverse of a given transform."""

    def __init__(self, transform):
        super().__init__()
        self._transform = transform

    def forward(self, inputs, context=None):
        return self._transform.inverse(inputs, context)

    def inverse(self, inputs, context=None):
        return self._transform(inputs, context)
