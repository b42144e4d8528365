"""Identity and fixed point-wise affine transforms (API of flowcon/transforms/standard.py:12-90)."""
import torch

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import Transform


class IdentityTransform(Transform):
    """Transform that leaves input unchanged."""

    def forward(self, inputs, context=None):
        return inputs, inputs.new_zeros(inputs.size(0))

    def inverse(self, inputs, context=None):
        return self(inputs, context)


class PointwiseAffineTransform(Transform):
    """Forward transform X = X * scale + shift (buffers ``_shift``, ``_scale``)."""

    def __init__(self, shift=0.0, scale=1.0):
        super().__init__()
        shift, scale = map(torch.as_tensor, (shift, scale))
        if (scale == 0.0).any():
            raise ValueError("Scale must be non-zero.")
        self.register_buffer("_shift", shift)
        self.register_buffer("_scale", scale)

    @property
    def _log_abs_scale(self):
        return torch.log(torch.abs(self._scale))

    def _batch_logabsdet(self, batch_shape):
        """log|det| of one batch item: sum of log|scale| broadcast over the item's shape."""
        las = self._log_abs_scale
        if las.numel() > 1:
            return las.expand(batch_shape).sum()
        return las * torch.Size(batch_shape).numel()

    def _map(self, inputs, inverse):
        batch_size, *batch_shape = inputs.size()
        outputs = ops.pointwise_affine_autograd(inputs, self._scale, self._shift, inverse=inverse)
        logabsdet = self._batch_logabsdet(batch_shape).to(torch.float32).expand(batch_size)
        return outputs, (-logabsdet if inverse else logabsdet)

    def forward(self, inputs, context=None):
        return self._map(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._map(inputs, inverse=True)


class AffineTransform(PointwiseAffineTransform):
    def __init__(self, shift=0.0, scale=1.0):
        if shift is None:
            shift = 0.0
        if scale is None:
            scale = 1.0
        super().__init__(shift, scale)


# Alias for backward compatibility.
AffineScalarTransform = AffineTransform
