"""Element-wise bijectors and batch-shared spline CDFs (API of flowcon/transforms/nonlinearities.py)."""
import numpy as np
import torch
from torch import nn

from flowconductor_amd import ops
from flowconductor_amd.transforms.base import CompositeTransform, InverseTransform, Transform


class Exp(Transform):
    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_EXP)

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_EXP, inverse=True, may_raise=True)


class Tanh(Transform):
    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_TANH)

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_TANH, inverse=True, may_raise=True)


class LogTanh(Transform):
    """Tanh with unbounded output: beyond +-cut_point it continues as +-alpha*log(beta*|x|), with
    alpha, beta matching value and slope of tanh at the cut point (nonlinearities.py:51-112)."""

    def __init__(self, cut_point=1):
        if cut_point <= 0:
            raise ValueError("Cut point must be positive.")
        super().__init__()
        self.cut_point = cut_point
        self.inv_cut_point = np.tanh(cut_point)
        self.alpha = (1 - np.tanh(np.tanh(cut_point))) / cut_point
        self.beta = np.exp((np.tanh(cut_point) - self.alpha * np.log(cut_point)) / self.alpha)

    def _p(self):
        return (self.cut_point, self.alpha, self.beta, self.inv_cut_point)

    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_LOGTANH, p=self._p())

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_LOGTANH, inverse=True, p=self._p())


class LeakyReLU(Transform):
    def __init__(self, negative_slope=1e-2):
        if negative_slope <= 0:
            raise ValueError("Slope must be positive.")
        super().__init__()
        self.negative_slope = negative_slope
        self.log_negative_slope = torch.nn.Parameter(torch.log(torch.as_tensor(self.negative_slope)))

    def _p(self):
        return (self.negative_slope, 1 / self.negative_slope)

    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_LEAKY_RELU, aux=self.log_negative_slope.reshape(1), p=self._p())

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_LEAKY_RELU, inverse=True, aux=self.log_negative_slope.reshape(1),
                               p=self._p())


class Sigmoid(Transform):
    def __init__(self, temperature=1, eps=1e-6, learn_temperature=False):
        super().__init__()
        self.eps = eps
        if learn_temperature:
            self.temperature = nn.Parameter(torch.Tensor([temperature]))
        else:
            self.register_buffer("temperature", torch.Tensor([temperature]))

    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_SIGMOID, aux=self.temperature, p=(self.eps,))

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_SIGMOID, inverse=True, aux=self.temperature, p=(self.eps,),
                               may_raise=True)


class Softplus(Transform):
    """softplus(x) + eps; logabsdet = sum logsigmoid(x) (nonlinearities.py:172-189)."""

    def __init__(self, threshold=20, eps=0.):
        super().__init__()
        self.eps = eps
        self.softplus = torch.nn.Softplus(beta=1, threshold=threshold)
        self.log_sigmoid = torch.nn.LogSigmoid()

    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_SOFTPLUS, p=(self.softplus.threshold, self.eps))

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_SOFTPLUS, inverse=True, p=(self.softplus.threshold, self.eps))


class Logit(InverseTransform):
    def __init__(self, temperature=1, eps=1e-6):
        super().__init__(Sigmoid(temperature=temperature, eps=eps))


class GatedLinearUnit(Transform):
    """y = x * sigmoid(context); logabsdet = log sigmoid(context) flattened (1-feature inputs)."""

    def __init__(self):
        super().__init__()

    def forward(self, inputs, context=None):
        y, lad = ops.elementwise(inputs, ops.EW_GLU, aux=context.expand_as(inputs).contiguous(), row_sum=False,
                                 elem_lad=True)
        return y, lad.reshape(-1)

    def inverse(self, inputs, context=None):
        y, lad = ops.elementwise(inputs, ops.EW_GLU, inverse=True, aux=context.expand_as(inputs).contiguous(),
                                 row_sum=False, elem_lad=True)
        return y, lad.reshape(-1)


class CauchyCDF(Transform):
    def __init__(self, location=None, scale=None, features=None):
        super().__init__()

    def forward(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_CAUCHY_CDF)

    def inverse(self, inputs, context=None):
        return ops.elementwise(inputs, ops.EW_CAUCHY_CDF, inverse=True, may_raise=True)


class CauchyCDFInverse(InverseTransform):
    def __init__(self, location=None, scale=None, features=None):
        super().__init__(CauchyCDF(location=location, scale=scale, features=features))


class CompositeCDFTransform(CompositeTransform):
    """squash -> cdf -> unsquash (nonlinearities.py:239-243)."""

    def __init__(self, squashing_transform, cdf_transform):
        super().__init__([squashing_transform, cdf_transform, InverseTransform(squashing_transform)])


def _flatten_items(inputs):
    """[N, *shape] -> [N, prod(shape)] view for the [N, D] kernels."""
    return inputs.flatten(1) if inputs.dim() > 1 else inputs.reshape(-1, 1)


class PiecewiseRationalQuadraticCDF(Transform):
    """RQ spline with learnable parameters shared across the batch (nonlinearities.py:429-487).

    Parameters ``unnormalized_widths/heights/derivatives`` have shape ``[*shape, K]`` /
    ``[*shape, K -/+ 1]``; the kernel reads one concatenated ``[prod(shape), 3K -/+ 1]`` row set
    staged once per workgroup in LDS instead of an ``expand``-ed ``[N, ...]`` view.
    """

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0, identity_init=False,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        if isinstance(shape, int):
            shape = (shape,)
        num_derivatives = (num_bins - 1) if self.tails == "linear" else (num_bins + 1)
        if identity_init:
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            constant = np.log(np.exp(1 - min_derivative) - 1)
            self.unnormalized_derivatives = nn.Parameter(constant * torch.ones(*shape, num_derivatives))
        else:
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, num_derivatives))

    _HIP_AUTOGRAD = True

    def _spline(self, inputs, inverse=False):
        flat = _flatten_items(inputs)
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training: the shared row expanded to per-sample rows (the backward kernel's layout); autograd sums the
            # row gradients back through the expand
            rows = torch.cat((self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives),
                             dim=-1).reshape(1, -1).expand(flat.shape[0], -1).contiguous()
            outputs, logabsdet = ops.rq_spline_autograd(
                flat, rows, None, num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                min_derivative=self.min_derivative, inverse=inverse)
            return outputs.reshape(inputs.shape), logabsdet
        rows = torch.cat((self.unnormalized_widths.detach(), self.unnormalized_heights.detach(),
                          self.unnormalized_derivatives.detach()), dim=-1).reshape(-1)
        outputs, logabsdet = ops.rq_spline(
            flat, rows, None, num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
            min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative, inverse=inverse, shared_params=True)
        return outputs.reshape(inputs.shape), logabsdet

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


class _SharedSplineCDF(Transform):
    """Common part of the batch-shared spline CDFs: concatenate the learnable per-dim parameters into
    one ``[prod(shape), multiplier]`` row set and run the spline kernel with ``shared_params``."""

    _kind = None
    _HIP_AUTOGRAD = True

    def _rows(self, detach=True):
        raise NotImplementedError()

    def _spline(self, inputs, inverse=False):
        flat = _flatten_items(inputs)
        kw = dict(kind=self._kind, num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                  min_bin_width=getattr(self, "min_bin_width", ops.DEFAULT_MIN_BIN_WIDTH),
                  min_bin_height=getattr(self, "min_bin_height", ops.DEFAULT_MIN_BIN_HEIGHT))
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training: the shared row set expanded to per-sample rows; autograd sums their gradients back
            rows = self._rows(detach=False).reshape(1, -1).expand(flat.shape[0], -1).contiguous()
            outputs, logabsdet = ops.piecewise_spline_autograd(flat, rows, None, inverse=inverse, **kw)
            return outputs.reshape(inputs.shape), logabsdet
        outputs, logabsdet = ops.piecewise_spline_autograd(
            flat, self._rows(), None, kind=self._kind, num_bins=self.num_bins, tails=self.tails,
            tail_bound=self.tail_bound, min_bin_width=getattr(self, "min_bin_width", ops.DEFAULT_MIN_BIN_WIDTH),
            min_bin_height=getattr(self, "min_bin_height", ops.DEFAULT_MIN_BIN_HEIGHT), inverse=inverse,
            shared_params=True)
        return outputs.reshape(inputs.shape), logabsdet

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


class PiecewiseLinearCDF(_SharedSplineCDF):
    """nonlinearities.py:250-284."""

    _kind = ops.SPLINE_LINEAR

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0):
        super().__init__()
        self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        self.unnormalized_pdf = nn.Parameter(torch.randn(*shape, num_bins))

    def _rows(self, detach=True):
        return (self.unnormalized_pdf.detach() if detach else self.unnormalized_pdf).reshape(-1)


class PiecewiseQuadraticCDF(_SharedSplineCDF):
    """nonlinearities.py:287-339."""

    _kind = ops.SPLINE_QUADRATIC

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, num_bins))
        n_heights = num_bins + 1 if tails is None else num_bins - 1
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, n_heights))

    def _rows(self, detach=True):
        rows = torch.cat((self.unnormalized_widths, self.unnormalized_heights), dim=-1).reshape(-1)
        return rows.detach() if detach else rows


class PiecewiseCubicCDF(_SharedSplineCDF):
    """nonlinearities.py:342-426."""

    _kind = ops.SPLINE_CUBIC

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnorm_derivatives_left = nn.Parameter(torch.randn(*shape, 1))
        self.unnorm_derivatives_right = nn.Parameter(torch.randn(*shape, 1))

    def _rows(self, detach=True):
        rows = torch.cat((self.unnormalized_widths, self.unnormalized_heights, self.unnorm_derivatives_left,
                          self.unnorm_derivatives_right), dim=-1).reshape(-1)
        return rows.detach() if detach else rows


class ExtendedSoftplus(torch.nn.Module):
    """softplus(x - s) - softplus(-(x + s)) with s = softplus(shift) + 0.1: linear far from the
    origin, flat around it.  ``forward`` returns (outputs, element-wise log-derivative)
    (nonlinearities.py:490-552).  Helper of SumOfSigmoids, which fuses it into its own kernel."""

    def __init__(self, features, shift=None):
        self.features = features
        super().__init__()
        if shift is None:
            self.shift = torch.nn.Parameter(torch.ones(1, features) * 3, requires_grad=True)
        elif torch.is_tensor(shift):
            self.shift = shift.reshape(-1, features)
        else:
            self.shift = torch.nn.Parameter(torch.tensor(shift), requires_grad=True)
        self._softplus = torch.nn.Softplus()

    def get_shift(self):
        return self._softplus(self.shift) + 1e-1

    def forward(self, inputs):
        shift = self.shift.detach()
        if shift.shape[0] != 1:
            raise NotImplementedError("per-sample ExtendedSoftplus runs inside the SumOfSigmoids kernel")
        return ops.elementwise(inputs, ops.EW_EXTENDED_SOFTPLUS, aux=shift.reshape(-1), row_sum=False,
                               elem_lad=True)
