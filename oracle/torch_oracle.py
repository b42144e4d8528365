"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by flowconductor_amd.

A torch-CPU restatement of the reference's algorithm for the bijector hot path: the same ATen
op sequences, in the same order, that FlowConductor's Python executes (the reference's
arithmetic *is* stock ATen), written from the survey's description with the
reference file:line each function follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.

Pinned: every function here is checked against golden vectors produced by importing the
reference itself in the build container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``), and against the reference's own
known-answer tests restated in ``tests/test_oracle_known_answers.py``.

Works in float32 (parity / CPU baseline) and float64 (error-vs-truth measurements).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


class OracleInputOutsideDomain(Exception):
    """Stands in for flowcon.transforms.base.InputOutsideDomain (transforms/base.py:16)."""


# ---- utils/torchutils.py ------------------------------------------------------------------------

def sum_except_batch(x, num_batch_dims=1):
    """utils/torchutils.py:25-30."""
    return torch.sum(x, dim=list(range(num_batch_dims, x.dim())))


def searchsorted(bin_locations, inputs, eps=1e-6):
    """utils/torchutils.py:147-149 -- compare-count; mutates the last edge in place like the reference."""
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


# ---- splines/rational_quadratic.py ----------------------------------------------------------------

def _knots(unnormalized, lo, hi, min_size, num_bins):
    """softmax -> floor + rescale -> cumsum -> pad 0 -> affine -> pinned ends -> sizes
    (rational_quadratic.py:91-98 and :106-113)."""
    sizes = F.softmax(unnormalized, dim=-1)
    sizes = min_size + (1 - min_size * num_bins) * sizes
    knots = torch.cumsum(sizes, dim=-1)
    knots = F.pad(knots, pad=(1, 0), mode="constant", value=0.0)
    knots = (hi - lo) * knots + lo
    knots[..., 0] = lo
    knots[..., -1] = hi
    return knots, knots[..., 1:] - knots[..., :-1]


def rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights,
                              unnormalized_derivatives, inverse=False, left=0.0, right=1.0,
                              bottom=0.0, top=1.0, min_bin_width=1e-3, min_bin_height=1e-3,
                              min_derivative=1e-3, enable_identity_init=False):
    """splines/rational_quadratic.py:66-181."""
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise OracleInputOutsideDomain()
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")

    cumwidths, widths = _knots(unnormalized_widths, left, right, min_bin_width, num_bins)
    beta = np.log(2) / (1 - min_derivative) if enable_identity_init else 1
    derivatives = min_derivative + F.softplus(unnormalized_derivatives, beta=beta)
    cumheights, heights = _knots(unnormalized_heights, bottom, top, min_bin_height, num_bins)

    k = searchsorted(cumheights if inverse else cumwidths, inputs)[..., None]

    def at(t):
        return t.gather(-1, k)[..., 0]

    x_k, w_k = at(cumwidths), at(widths)
    y_k, h_k = at(cumheights), at(heights)
    s_k = at(heights / widths)
    d_k, d_k1 = at(derivatives), at(derivatives[..., 1:])
    curv = d_k + d_k1 - 2 * s_k

    if inverse:
        r = inputs - y_k
        a = r * curv + h_k * (s_k - d_k)
        b = h_k * d_k - r * curv
        c = -s_k * r
        disc = b.pow(2) - 4 * a * c
        assert (disc >= 0).all()
        theta = (2 * c) / (-b - torch.sqrt(disc))
        outputs = theta * w_k + x_k
    else:
        theta = (inputs - x_k) / w_k

    tt = theta * (1 - theta)
    denominator = s_k + curv * tt
    dnum = s_k.pow(2) * (d_k1 * theta.pow(2) + 2 * s_k * tt + d_k * (1 - theta).pow(2))
    logabsdet = torch.log(dnum) - 2 * torch.log(denominator)
    if inverse:
        return outputs, -logabsdet
    numerator = h_k * (s_k * theta.pow(2) + d_k * tt)
    return y_k + numerator / denominator, logabsdet


def unconstrained_rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights,
                                            unnormalized_derivatives, inverse=False, tails="linear",
                                            tail_bound=1.0, min_bin_width=1e-3, min_bin_height=1e-3,
                                            min_derivative=1e-3, enable_identity_init=False):
    """splines/rational_quadratic.py:13-63: identity outside [-B, B], spline on the masked subset."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outside = ~inside
    outputs = torch.zeros_like(inputs)
    logabsdet = torch.zeros_like(inputs)
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    unnormalized_derivatives = F.pad(unnormalized_derivatives, pad=(1, 1))
    constant = np.log(np.exp(1 - min_derivative) - 1)
    unnormalized_derivatives[..., 0] = constant
    unnormalized_derivatives[..., -1] = constant
    outputs[outside] = inputs[outside]
    logabsdet[outside] = 0
    if torch.any(inside):
        outputs[inside], logabsdet[inside] = rational_quadratic_spline(
            inputs=inputs[inside],
            unnormalized_widths=unnormalized_widths[inside, :],
            unnormalized_heights=unnormalized_heights[inside, :],
            unnormalized_derivatives=unnormalized_derivatives[inside, :],
            inverse=inverse, left=-tail_bound, right=tail_bound, bottom=-tail_bound, top=tail_bound,
            min_bin_width=min_bin_width, min_bin_height=min_bin_height,
            min_derivative=min_derivative, enable_identity_init=enable_identity_init)
    return outputs, logabsdet


def rq_from_rows(inputs, rows, num_bins, tails, tail_bound, inverse, wh_divisor=None,
                 box=(0.0, 1.0, 0.0, 1.0), enable_identity_init=False, mins=(1e-3, 1e-3, 1e-3)):
    """[N, d] inputs + [N, d, 3K-/+1] parameter rows -> element-wise (outputs, logabsdet).

    Slicing / in-place scaling as coupling.py:549-563 and autoregressive.py:583-591."""
    uw = rows[..., :num_bins]
    uh = rows[..., num_bins:2 * num_bins]
    ud = rows[..., 2 * num_bins:]
    if wh_divisor is not None:
        uw /= wh_divisor
        uh /= wh_divisor
    kw = dict(inverse=inverse, min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2],
              enable_identity_init=enable_identity_init)
    if tails is None:
        return rational_quadratic_spline(inputs, uw, uh, ud, left=box[0], right=box[1], bottom=box[2],
                                         top=box[3], **kw)
    return unconstrained_rational_quadratic_spline(inputs, uw, uh, ud, tails=tails,
                                                   tail_bound=tail_bound, **kw)


# ---- affine bijectors -------------------------------------------------------------------------------

def affine_coupling_scale_shift(params, d_t, kind="sigmoid"):
    """coupling.py:234-238 with the two predefined activations (:224-225)."""
    shift = params[:, :d_t, ...]
    u = params[:, d_t:, ...]
    if kind == "sigmoid":
        scale = torch.sigmoid(u + 2) + 1e-3
    elif kind == "softplus_clamp":
        scale = (F.softplus(u) + 1e-3).clamp(0, 3)
    else:
        raise ValueError(kind)
    return scale, shift


def affine_elementwise(inputs, scale, shift, inverse):
    """coupling.py:240-252 / autoregressive.py:97-118."""
    log_scale = torch.log(scale)
    if inverse:
        return (inputs - shift) / scale, -sum_except_batch(log_scale)
    return inputs * scale + shift, sum_except_batch(log_scale)


def maf_scale_shift(params, features):
    """autoregressive.py:120-129 + :102: view(-1, D, 2); scale = softplus(p0) + 1e-3."""
    p = params.view(-1, features, 2)
    return F.softplus(p[..., 0]) + 1e-3, p[..., 1]


# ---- distributions/normal.py -------------------------------------------------------------------------

def standard_normal_log_prob(inputs):
    """distributions/normal.py:23-33 (log_z is a float64 0-dim tensor there; result keeps inputs' dtype)."""
    d = int(np.prod(inputs.shape[1:]))
    log_z = torch.tensor(0.5 * d * np.log(2 * np.pi), dtype=torch.float64)
    return -0.5 * sum_except_batch(inputs ** 2) - log_z.to(inputs.dtype)


# ---- module walker: evaluates a Transform tree with the functions above --------------------------------
#
# Dispatch is by class NAME and public attribute names, which the reference and
# flowconductor_amd share (API parity), so the same walker evaluates either tree on the CPU.
# Conditioner networks (nn.Modules, "not replaced") are simply called.

def _coupling_split(t, inputs):
    return inputs[:, t.identity_features, ...], inputs[:, t.transform_features, ...]


def _coupling(t, inputs, context, inverse, elementwise):
    """coupling.py:73-130."""
    if inputs.dim() not in (2, 4):
        raise ValueError("Inputs must be a 2D or a 4D tensor.")
    if inputs.shape[1] != t.features:
        raise ValueError("Expected features = {}, got {}.".format(t.features, inputs.shape[1]))
    identity_split, transform_split = _coupling_split(t, inputs)
    unc = getattr(t, "unconditional_transform", None)
    logabsdet = 0.0
    if inverse and unc is not None:
        identity_split, logabsdet = transform_apply(unc, identity_split, context, inverse=True)
    params = t.transform_net(identity_split, context)
    transform_split, lad = elementwise(t, transform_split, params, inverse)
    logabsdet = lad + logabsdet
    if not inverse and unc is not None:
        identity_split, lad_id = transform_apply(unc, identity_split, context, inverse=False)
        logabsdet = logabsdet + lad_id
    outputs = torch.empty_like(inputs)
    outputs[:, t.identity_features, ...] = identity_split
    outputs[:, t.transform_features, ...] = transform_split
    return outputs, logabsdet


def _affine_kind(t):
    act = getattr(t, "scale_activation", None)
    cls = type(t)
    if act is getattr(cls, "DEFAULT_SCALE_ACTIVATION", object()):
        return "sigmoid"
    if act is getattr(cls, "GENERAL_SCALE_ACTIVATION", object()):
        return "softplus_clamp"
    return None


def _ew_affine_coupling(t, x, params, inverse):
    kind = _affine_kind(t)
    d_t = len(t.transform_features)
    if kind is None:
        scale, shift = t.scale_activation(params[:, d_t:, ...]), params[:, :d_t, ...]
    else:
        scale, shift = affine_coupling_scale_shift(params, d_t, kind)
    return affine_elementwise(x, scale, shift, inverse)


def _ew_additive_coupling(t, x, params, inverse):
    """coupling.py:255-269."""
    return affine_elementwise(x, torch.ones_like(params), params, inverse)


def _piecewise_rows(x, params):
    """coupling.py:279-289."""
    if x.dim() == 4:
        b, c, h, w = x.shape
        return params.reshape(b, c, -1, h, w).permute(0, 1, 3, 4, 2)
    b, d = x.shape
    return params.reshape(b, d, -1)


def _net_divisor(net):
    if hasattr(net, "hidden_features"):
        return np.sqrt(net.hidden_features)
    if hasattr(net, "hidden_channels"):
        return np.sqrt(net.hidden_channels)
    return None


def _ew_rq_coupling(t, x, params, inverse):
    """coupling.py:549-582."""
    rows = _piecewise_rows(x, params)
    y, lad = rq_from_rows(x, rows, t.num_bins, t.tails, t.tail_bound, inverse,
                          wh_divisor=_net_divisor(t.transform_net),
                          mins=(t.min_bin_width, t.min_bin_height, t.min_derivative))
    return y, sum_except_batch(lad)


def _autoregressive(t, inputs, context, inverse, elementwise):
    """autoregressive.py:39-53."""
    if not inverse:
        return elementwise(t, inputs, t.autoregressive_net(inputs, context), False)
    outputs = torch.zeros_like(inputs)
    logabsdet = None
    for _ in range(int(np.prod(inputs.shape[1:]))):
        params = t.autoregressive_net(outputs, context)
        outputs, logabsdet = elementwise(t, inputs, params, True)
    return outputs, logabsdet


def _ew_maf(t, x, params, inverse):
    scale, shift = maf_scale_shift(params, t.features)
    return affine_elementwise(x, scale, shift, inverse)


def _ew_maf_shift(t, x, params, inverse):
    """autoregressive.py:164-196 (forward adds 2*tanh(shift); inverse subtracts raw shift)."""
    shift = params.view(-1, t.features) * t.shift_scale
    zeros = torch.zeros(x.shape[0], dtype=x.dtype)
    if inverse:
        return x - shift, zeros
    return x + torch.tanh(shift) * 2, zeros


def _ew_rq_ar(t, x, params, inverse):
    """autoregressive.py:583-621."""
    b, d = x.shape[0], x.shape[1]
    mult = params.shape[1] // d
    rows = params.view(b, d, mult)
    y, lad = rq_from_rows(x, rows, t.num_bins, t.tails, t.tail_bound, inverse,
                          wh_divisor=_net_divisor(t.autoregressive_net), box=(-1.2, 1.2, -1.2, 1.2),
                          enable_identity_init=True,
                          mins=(t.min_bin_width, t.min_bin_height, t.min_derivative))
    return y, sum_except_batch(lad)


def _rq_cdf(t, inputs, context, inverse):
    """nonlinearities.py:429-487 (parameters expanded across the batch)."""
    n = inputs.shape[0]

    def share(p):
        return p.detach()[None, ...].expand(n, *p.shape)

    rows = torch.cat((share(t.unnormalized_widths), share(t.unnormalized_heights),
                      share(t.unnormalized_derivatives)), dim=-1).clone()
    k = t.unnormalized_widths.shape[-1]
    y, lad = rq_from_rows(inputs, rows, k, t.tails, t.tail_bound, inverse,
                          mins=(t.min_bin_width, t.min_bin_height, t.min_derivative))
    return y, sum_except_batch(lad)


def _permutation(t, inputs, context, inverse):
    """permutations.py:23-46."""
    perm = t._permutation
    if inverse:
        perm = torch.argsort(perm)
    dim = t._dim
    if dim >= inputs.dim():
        raise ValueError("No dimension {} in inputs.".format(dim))
    if inputs.shape[dim] != len(perm):
        raise ValueError("Dimension {} in inputs must be of size {}.".format(dim, len(perm)))
    return torch.index_select(inputs, dim, perm), inputs.new_zeros(inputs.shape[0])


def _pointwise_affine(t, inputs, context, inverse):
    """standard.py:24-68."""
    batch_size, *batch_shape = inputs.size()
    las = torch.log(torch.abs(t._scale))
    if las.numel() > 1:
        lad = las.expand(batch_shape).sum()
    else:
        lad = las * torch.Size(batch_shape).numel()
    if inverse:
        return (inputs - t._shift) / t._scale, -lad.expand(batch_size)
    return inputs * t._scale + t._shift, lad.expand(batch_size)


def _identity(t, inputs, context, inverse):
    return inputs, inputs.new_zeros(inputs.size(0))


def _composite(t, inputs, context, inverse):
    """base.py:44-60."""
    layers = list(t._transforms)
    if inverse:
        layers = layers[::-1]
    outputs = inputs
    total = inputs.new_zeros(inputs.shape[0])
    for layer in layers:
        outputs, lad = transform_apply(layer, outputs, context, inverse)
        total += lad
    return outputs, total


def _inverse_transform(t, inputs, context, inverse):
    """base.py:215-231."""
    return transform_apply(t._transform, inputs, context, not inverse)


_DISPATCH = {
    "CompositeTransform": _composite,
    "CompositeCDFTransform": _composite,
    "InverseTransform": _inverse_transform,
    "IdentityTransform": _identity,
    "PointwiseAffineTransform": _pointwise_affine,
    "AffineTransform": _pointwise_affine,
    "Permutation": _permutation,
    "RandomPermutation": _permutation,
    "ReversePermutation": _permutation,
    "AffineCouplingTransform": lambda t, x, c, inv: _coupling(t, x, c, inv, _ew_affine_coupling),
    "AdditiveCouplingTransform": lambda t, x, c, inv: _coupling(t, x, c, inv, _ew_additive_coupling),
    "PiecewiseRationalQuadraticCouplingTransform":
        lambda t, x, c, inv: _coupling(t, x, c, inv, _ew_rq_coupling),
    "MaskedAffineAutoregressiveTransform": lambda t, x, c, inv: _autoregressive(t, x, c, inv, _ew_maf),
    "MaskedShiftAutoregressiveTransform":
        lambda t, x, c, inv: _autoregressive(t, x, c, inv, _ew_maf_shift),
    "MaskedPiecewiseRationalQuadraticAutoregressiveTransform":
        lambda t, x, c, inv: _autoregressive(t, x, c, inv, _ew_rq_ar),
    "PiecewiseRationalQuadraticCDF": _rq_cdf,
}


def register(name, fn):
    _DISPATCH[name] = fn


def transform_apply(t, inputs, context=None, inverse=False):
    """Evaluate transform module ``t`` (reference-compatible attribute names) on the CPU."""
    fn = _DISPATCH.get(type(t).__name__)
    if fn is None:
        raise NotImplementedError("oracle has no restatement for %s" % type(t).__name__)
    return fn(t, inputs, context, inverse)


def flow_log_prob(flow, inputs, context=None):
    """flows/base.py:41-48 with a StandardNormal base."""
    emb = flow._embedding_net(context)
    noise, logabsdet = transform_apply(flow._transform, inputs, emb, inverse=False)
    dist = type(flow._distribution).__name__
    if dist != "StandardNormal":
        raise NotImplementedError(dist)
    return standard_normal_log_prob(noise) + logabsdet
