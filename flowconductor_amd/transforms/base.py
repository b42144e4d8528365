"""Transform protocol and composition (API of flowcon/transforms/base.py:10-60, 215-231).

``forward(inputs, context=None) -> (outputs, logabsdet[N])``; ``inverse`` likewise.
``CompositeTransform`` defers the device error-word read to once per cascade instead of
the reference's host sync inside every spline call.
"""
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.ops import InputOutsideDomain, InverseNotAvailable  # noqa: F401 (re-export)


def _guard_missing_backward(fn):
    """forward / inverse of a transform whose kernels have no backward: called with autograd on and trainable own
    parameters, its outputs would carry no gradient and those parameters would silently never train -- fail loudly."""
    import functools

    @functools.wraps(fn)
    def guarded(self, *args, **kwargs):
        self._check_autograd()
        return fn(self, *args, **kwargs)

    guarded._guarded = True
    return guarded


def own_autograd_check(fn):
    """Marks a forward / inverse that calls ``self._check_autograd()`` itself (after checks of its own)."""
    fn._guarded = True
    return fn


class Transform(ops.RuntimeCaches, nn.Module):
    """Base class for all transform objects."""

    # True for transforms whose HIP kernels sit behind torch.autograd (or that only delegate to children); every
    # other subclass with trainable parameters of its own refuses to run with autograd on (see the guard above)
    _HIP_AUTOGRAD = False

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if not cls.__module__.startswith("flowconductor_amd."):
            return      # a user's own Transform (plain torch ops, differentiable as it is) is none of our business
        for name in ("forward", "inverse"):
            fn = cls.__dict__.get(name)
            if fn is not None and not getattr(fn, "_guarded", False):
                setattr(cls, name, _guard_missing_backward(fn))

    def _check_autograd(self):
        if (not self._HIP_AUTOGRAD and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.parameters(recurse=False))):
            raise RuntimeError(
                "flowconductor_amd: %s has no backward kernel; with autograd on its parameters would silently not "
                "train.  Wrap the call in torch.no_grad() (inference) or freeze the layer with requires_grad_(False)."
                % type(self).__name__)

    def forward(self, inputs, context=None):
        raise NotImplementedError()

    def inverse(self, inputs, context=None):
        raise InverseNotAvailable()

    def train(self, mode=True):
        """Packed-weight caches (kernel-layout copies of parameters) never survive a switch of the training mode:
        see ``ops.invalidate_hip_caches``."""
        ops.invalidate_hip_caches()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.float()`` / ``.cuda()`` replace parameter storage: the memoised fast-path predicates and packed
        images of every module go with them."""
        ops.invalidate_hip_caches()
        return super()._apply(fn, *args, **kwargs)


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
