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
import contextlib
import math

import numpy as np
import torch
import torch.nn.functional as F


_DIFFERENTIABLE_PARAMETERS = False


def _p(parameter):
    """A module parameter as the oracle uses it: detached (the oracle is an inference restatement) unless
    ``differentiable_parameters()`` is active (gradient checks against torch.autograd walking the oracle)."""
    return parameter if _DIFFERENTIABLE_PARAMETERS else parameter.detach()


@contextlib.contextmanager
def differentiable_parameters():
    global _DIFFERENTIABLE_PARAMETERS
    previous, _DIFFERENTIABLE_PARAMETERS = _DIFFERENTIABLE_PARAMETERS, True
    try:
        yield
    finally:
        _DIFFERENTIABLE_PARAMETERS = previous


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
        return _p(p)[None, ...].expand(n, *p.shape)

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



# ---- orthogonal.py ---------------------------------------------------------------------------------

def householder_apply(inputs, q_vectors):
    """orthogonal.py:144-171: K sequential reflections; q_vectors [K, D] or per-sample [N, K, D]."""
    squared_norms = torch.sum(q_vectors ** 2, dim=-1)
    outputs = inputs
    for i in range(q_vectors.shape[-2]):
        q = q_vectors[..., i, :]
        sq = squared_norms[..., i].unsqueeze(-1)
        ip = (outputs * q).sum(-1)
        outputs = outputs - ip.unsqueeze(-1) * ((2.0 / sq) * q)
    return outputs


def _householder(t, inputs, context, inverse):
    """orthogonal.py:63-72 / :111-117."""
    q = _p(t.q_vectors) if isinstance(t.q_vectors, torch.nn.Parameter) else t.q_vectors
    if inverse:
        q = q.flip(-2)
    return householder_apply(inputs, q), inputs.new_zeros(inputs.shape[0])


# ---- no_analytic_inv/planar.py -------------------------------------------------------------------------

def _planar(t, inputs, context, inverse):
    """planar.py:30-69 (no inverse)."""
    if inverse:
        raise NotImplementedError("PlanarTransform has no inverse")
    w, u, b = _p(t.w), _p(t.u), _p(t.b)
    wtu = torch.mm(u, w.T)
    m_wtu = -1 + F.softplus(wtu)
    u_hat = u + (m_wtu - wtu) * (w / (torch.norm(w, p=2, dim=1) ** 2))
    a = torch.mm(inputs, w.T) + b
    outputs = inputs + u_hat * torch.tanh(a)
    psi = (1 - torch.tanh(a) ** 2) * w
    abs_det = (1 + torch.mm(u_hat, psi.T)).abs()
    return outputs, torch.log(1e-7 + abs_det).squeeze()


def _upper_from(entries, diag, features):
    iu = np.triu_indices(features, k=1)
    m = entries.new_zeros(features, features)
    m[iu[0], iu[1]] = entries
    m[range(features), range(features)] = diag
    return m


def sylvester_forward(inputs, q_vectors, r1, r2, bias):
    """planar.py:144-166 with Q applied as reflections (orthogonal.py:63-72)."""
    qtz = householder_apply(inputs, q_vectors.flip(-2))
    rqtz = r1.unsqueeze(0) @ qtz.unsqueeze(-1) if r1.dim() == 2 else r1 @ qtz.unsqueeze(-1)
    preact = rqtz.squeeze(-1) + (bias.unsqueeze(0) if bias.dim() == 1 else bias)
    act = torch.tanh(preact)
    ract = r2.unsqueeze(0) @ act.unsqueeze(-1) if r2.dim() == 2 else r2 @ act.unsqueeze(-1)
    qract = householder_apply(ract.squeeze(-1), q_vectors)
    outputs = inputs + qract
    r_sq = torch.diagonal(r1, dim1=-2, dim2=-1) * torch.diagonal(r2, dim1=-2, dim2=-1)
    diag = 1 + (1 - act ** 2) * (r_sq.unsqueeze(0) if r_sq.dim() == 1 else r_sq)
    return outputs, torch.log(diag).sum(-1)


def _sylvester(t, inputs, context, inverse):
    if inverse:
        raise NotImplementedError("SylvesterTransform has no inverse")
    f = t.features
    r1 = _upper_from(_p(t.upper_entries1), torch.tanh(_p(t.log_upper_diag1)), f)
    r2 = _upper_from(_p(t.upper_entries2), torch.tanh(_p(t.log_upper_diag2)), f)
    return sylvester_forward(inputs, _p(t.Q_orth.q_vectors), r1, r2, _p(t.bias))


# ---- lu.py / linear.py ---------------------------------------------------------------------------------

def lu_matrices(t):
    """lu.py:44-54."""
    f = t.features
    il = np.tril_indices(f, k=-1)
    lower = _p(t.lower_entries).new_zeros(f, f)
    lower[il[0], il[1]] = _p(t.lower_entries)
    lower[range(f), range(f)] = 1.0
    upper_diag = F.softplus(_p(t.unconstrained_upper_diag)) + t.eps
    upper = _upper_from(_p(t.upper_entries), upper_diag, f)
    return lower, upper, upper_diag


def _lu_linear(t, inputs, context, inverse):
    """lu.py:56-91 (no-cache path) and linear.py:45-76 (eval-mode cache path)."""
    lower, upper, upper_diag = lu_matrices(t)
    logabsdet = torch.sum(torch.log(upper_diag))
    bias = _p(t.bias)
    ones = inputs.new_ones(inputs.shape[0])
    if not t.training and t.using_cache:
        if not inverse:
            return F.linear(inputs, lower @ upper, bias), logabsdet * ones
        eye = torch.eye(t.features, dtype=inputs.dtype)
        linv = torch.linalg.solve_triangular(lower, eye, upper=False, unitriangular=True)
        winv = torch.linalg.solve_triangular(upper, linv, upper=True, unitriangular=False)
        return F.linear(inputs - bias, winv), (-logabsdet) * ones
    if not inverse:
        outputs = F.linear(inputs, upper)
        return F.linear(outputs, lower, bias), logabsdet * ones
    outputs = inputs - bias
    outputs = torch.linalg.solve_triangular(lower, outputs.t(), upper=False, unitriangular=True)
    outputs = torch.linalg.solve_triangular(upper, outputs, upper=True, unitriangular=False)
    return outputs.t(), -logabsdet * ones


# ---- normalization.py ---------------------------------------------------------------------------------

def _actnorm(t, inputs, context, inverse):
    """normalization.py:171-204 (initialised / eval mode)."""
    if inputs.dim() not in (2, 4):
        raise ValueError("Expecting inputs to be a 2D or a 4D tensor.")
    log_scale, shift = _p(t.log_scale), _p(t.shift)
    scale = torch.exp(log_scale)
    if inputs.dim() == 4:
        scale, shift = scale.view(1, -1, 1, 1), shift.view(1, -1, 1, 1)
        mult = inputs.shape[2] * inputs.shape[3]
    else:
        scale, shift = scale.view(1, -1), shift.view(1, -1)
        mult = 1
    lad = mult * torch.sum(log_scale) * inputs.new_ones(inputs.shape[0])
    if inverse:
        return (inputs - shift) / scale, -lad
    return scale * inputs + shift, lad


def _batchnorm(t, inputs, context, inverse):
    """normalization.py:98-141 in eval mode."""
    if inputs.dim() != 2:
        raise ValueError("Expected 2-dim inputs, got inputs of shape: {}".format(inputs.shape))
    weight = F.softplus(_p(t.unconstrained_weight)) + t.eps
    bias, mean, var = _p(t.bias), t.running_mean, t.running_var
    ones = inputs.new_ones(inputs.shape[0])
    if inverse:
        outputs = torch.sqrt(var + t.eps) * ((inputs - bias) / weight) + mean
        return outputs, torch.sum(-torch.log(weight) + 0.5 * torch.log(var + t.eps)) * ones
    outputs = weight * ((inputs - mean) / torch.sqrt(var + t.eps)) + bias
    return outputs, torch.sum(torch.log(weight) - 0.5 * torch.log(var + t.eps)) * ones


# ---- nonlinearities.py (element-wise) -------------------------------------------------------------------

def _exp(t, x, c, inverse):
    """nonlinearities.py:18-32."""
    if inverse:
        if torch.min(x) <= 0.:
            raise OracleInputOutsideDomain()
        y = torch.log(x)
        return y, -sum_except_batch(y)
    return torch.exp(x), sum_except_batch(x)


def _tanh(t, x, c, inverse):
    """nonlinearities.py:35-48."""
    if inverse:
        if torch.min(x) <= -1 or torch.max(x) >= 1:
            raise OracleInputOutsideDomain()
        return 0.5 * torch.log((1 + x) / (1 - x)), sum_except_batch(-torch.log(1 - x ** 2))
    y = torch.tanh(x)
    return y, sum_except_batch(torch.log(1 - y ** 2))


def _logtanh(t, x, c, inverse):
    """nonlinearities.py:51-112."""
    alpha, beta = t.alpha, t.beta
    y = torch.zeros_like(x)
    lad = torch.zeros_like(x)
    if not inverse:
        right, left = x > t.cut_point, x < -t.cut_point
        mid = ~(right | left)
        y[mid] = torch.tanh(x[mid])
        y[right] = alpha * torch.log(beta * x[right])
        y[left] = alpha * -torch.log(-beta * x[left])
        lad[mid] = torch.log(1 - y[mid] ** 2)
        lad[right] = torch.log(alpha / x[right])
        lad[left] = torch.log(-alpha / x[left])
    else:
        right, left = x > t.inv_cut_point, x < -t.inv_cut_point
        mid = ~(right | left)
        y[mid] = 0.5 * torch.log((1 + x[mid]) / (1 - x[mid]))
        y[right] = torch.exp(x[right] / alpha) / beta
        y[left] = -torch.exp(-x[left] / alpha) / beta
        lad[mid] = -torch.log(1 - x[mid] ** 2)
        lad[right] = -np.log(alpha * beta) + x[right] / alpha
        lad[left] = -np.log(alpha * beta) - x[left] / alpha
    return y, sum_except_batch(lad)


def _leaky_relu(t, x, c, inverse):
    """nonlinearities.py:115-136."""
    mask = (x < 0).to(x.dtype)
    ls = _p(t.log_negative_slope)
    if inverse:
        return F.leaky_relu(x, negative_slope=(1 / t.negative_slope)), sum_except_batch(-ls * mask)
    return F.leaky_relu(x, negative_slope=t.negative_slope), sum_except_batch(ls * mask)


def _sigmoid(t, x, c, inverse):
    """nonlinearities.py:139-169."""
    temp = _p(t.temperature)
    if inverse:
        if torch.min(x) < 0 or torch.max(x) > 1:
            raise OracleInputOutsideDomain()
        x = torch.clamp(x, t.eps, 1 - t.eps)
        y = (1 / temp) * (torch.log(x) - torch.log1p(-x))
        return y, -sum_except_batch(torch.log(temp) - F.softplus(-temp * y) - F.softplus(temp * y))
    x = temp * x
    return torch.sigmoid(x), sum_except_batch(torch.log(temp) - F.softplus(-x) - F.softplus(x))


def _softplus_t(t, x, c, inverse):
    """nonlinearities.py:172-189."""
    thr = t.softplus.threshold
    if inverse:
        x = x - t.eps
        return torch.where(x > thr, x, x.expm1().log()), -torch.log(-torch.expm1(-x)).sum(-1)
    return F.softplus(x, beta=1, threshold=thr) + t.eps, F.logsigmoid(x).sum(-1)


def _cauchy(t, x, c, inverse):
    """nonlinearities.py:212-231."""
    if inverse:
        if torch.min(x) < 0 or torch.max(x) > 1:
            raise OracleInputOutsideDomain()
        y = torch.tan(np.pi * (x - 0.5))
        return y, -sum_except_batch(-np.log(np.pi) - torch.log(1 + y ** 2))
    return (1 / np.pi) * torch.atan(x) + 0.5, sum_except_batch(-np.log(np.pi) - torch.log(1 + x ** 2))


def _glu(t, x, c, inverse):
    """nonlinearities.py:197-209."""
    gate = torch.sigmoid(c)
    if inverse:
        return x / gate, -torch.log(gate).reshape(-1)
    return x * gate, torch.log(gate).reshape(-1)


# ---- adaptive_sigmoids.py + ExtendedSoftplus + MonotonicTransform ------------------------------------------

def extended_softplus(x, raw_shift):
    """nonlinearities.py:519-552."""
    shift = F.softplus(raw_shift) + 1e-1
    out = F.softplus(x - shift) + (-F.softplus(-(x + shift)))
    lj = torch.logaddexp(-torch.logaddexp(shift, x) + x, -F.softplus(shift + x))
    return out, lj


def sos_forward(x, shift_preact, log_scale_preact, raw_softmax, esp_shift, log_scale_postact=0.0, eps=1e-6):
    """adaptive_sigmoids.py:108-142.  Parameter tensors broadcast against x[..., None]."""
    soft_max = F.softmax(raw_softmax, dim=-1) + eps
    soft_max = soft_max / soft_max.sum(-1).unsqueeze(-1)
    scale_postact = math.exp(log_scale_postact) * soft_max
    scale_preact = torch.sigmoid(log_scale_preact) * (10. - .1) + .1
    shift = torch.tanh(shift_preact) * 10
    pre = scale_preact * (x.unsqueeze(-1) - shift)
    sig = scale_postact * torch.sigmoid(pre)
    log_jac = torch.log(scale_postact) + torch.log(scale_preact) + (pre - 2 * F.softplus(pre))
    y_sos = sig.sum(-1) / scale_postact.sum(-1)
    lj_sos = torch.logsumexp(log_jac, -1)
    y_esp, lj_esp = extended_softplus(x, esp_shift)
    return y_sos + y_esp, torch.logaddexp(lj_sos, lj_esp).sum(-1)


def monotonic_inverse(forward_fn, z, lim, num_iterations, ratio_multiplier=1.5, atol=1e-7):
    """no_analytic_inv/base.py:23-83: batch-global bracket expansion, bisection, 2 Newton steps whose
    derivative comes from autograd through ``forward_fn`` (x -> (f(x), logabsdet))."""
    def diffs(z_max, z_min):
        d = z - z_max
        imax = torch.argmax(d)
        e = z - z_min
        imin = torch.argmin(e)
        return imax, imin, d.flatten()[imax], e.flatten()[imin]

    x_max = torch.ones_like(z) * lim
    x_min = -torch.ones_like(z) * lim
    z_max, _ = forward_fn(x_max)
    z_min, _ = forward_fn(x_min)
    imax, imin, maxdiff, mindiff = diffs(z_max, z_min)
    while maxdiff > 0:
        ratio = (maxdiff + z_max.flatten()[imax]) / z_max.flatten()[imax]
        x_max = x_max * ratio_multiplier * ratio
        z_max, _ = forward_fn(x_max)
        imax, imin, maxdiff, mindiff = diffs(z_max, z_min)
    x_max = x_max + 1
    while mindiff < 0:
        ratio = (mindiff + z_min.flatten()[imin]) / z_min.flatten()[imin]
        x_min = x_min * ratio_multiplier * ratio
        z_min, _ = forward_fn(x_min)
        imax, imin, maxdiff, mindiff = diffs(z_max, z_min)
    x_min = x_min - 1
    i = 0
    x_middle = (x_max + x_min) / 2
    while i < num_iterations and (x_middle - z).abs().max() > atol:
        x_middle = (x_max + x_min) / 2
        z_middle, _ = forward_fn(x_middle)
        left = (z_middle > z).to(z.dtype)
        right = (z_middle < z).to(z.dtype)
        equal = 1 - (left + right)
        x_max = left * x_middle + right * x_max + equal * x_middle
        x_min = right * x_middle + left * x_min + equal * x_middle
        i += 1
    x_guess = ((x_max + x_min) / 2).detach()
    with torch.enable_grad():
        x_guess = x_guess.requires_grad_(True)
        for _ in range(2):
            f = forward_fn(x_guess)[0] - z
            df_dx = torch.autograd.grad(f.sum(), x_guess, create_graph=False)[0].view(f.shape)
            x_guess = (x_guess - f / (df_dx + 1e-7)).detach().requires_grad_(True)
    x_guess = x_guess.detach()
    return x_guess, -forward_fn(x_guess)[1].reshape(-1)


def _sos_params(t):
    def val(p):
        return _p(p) if isinstance(p, torch.nn.Parameter) else p
    esp = t.extended_softplus.shift
    return (val(t.shift_preact), val(t.log_scale_preact), val(t.raw_softmax), val(esp),
            float(val(t.log_scale_postact).reshape(-1)[0]))


def _sum_of_sigmoids(t, x, c, inverse):
    sp, ls, rs, es, lp = _sos_params(t)

    def fwd(v):
        return sos_forward(v, sp, ls, rs, es, lp)

    if inverse:
        return monotonic_inverse(fwd, x, t.lim, t.num_iterations, t.ratio_multiplier, t.atol)
    return fwd(x)


def _ew_sos_ar(t, x, params, inverse):
    """autoregressive.py:301-318."""
    s = t.n_sigmoids
    raw = params.view(x.shape[0], t.features, 3 * s + 1)
    sp, ls, rs, es = torch.split(raw, [s, s, s, 1], dim=-1)
    es = es.reshape(-1, t.features)

    def fwd(v):
        return sos_forward(v, sp, ls, rs, es)

    if inverse:
        return monotonic_inverse(fwd, x + 0.5, 120, 50)
    z, lad = fwd(x)
    return z - 0.5, lad


def _named_inverse(inner_factory):
    """Logit / CauchyCDFInverse are InverseTransform subclasses (nonlinearities.py:192-194, 234-236)."""
    def fn(t, x, c, inverse):
        return transform_apply(t._transform, x, c, not inverse)
    return fn


# ---- splines/linear.py, quadratic.py, cubic.py -------------------------------------------------------------

def _unconstrained(spline_fn, inputs, tail_bound, tails, param_list, **kw):
    """Shared shape of the unconstrained_* wrappers (linear.py:8-35, quadratic.py:11-52, cubic.py:14-60)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outside = ~inside
    outputs = torch.zeros_like(inputs)
    logabsdet = torch.zeros_like(inputs)
    if tails != "linear":
        raise RuntimeError("{} tails are not implemented.".format(tails))
    outputs[outside] = inputs[outside]
    logabsdet[outside] = 0
    if torch.any(inside):
        outputs[inside], logabsdet[inside] = spline_fn(
            inputs[inside], *[p[inside, :] for p in param_list], left=-tail_bound, right=tail_bound,
            bottom=-tail_bound, top=tail_bound, **kw)
    return outputs, logabsdet


def linear_spline(inputs, unnormalized_pdf, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0):
    """splines/linear.py:38-105."""
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise OracleInputOutsideDomain()
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)
    num_bins = unnormalized_pdf.size(-1)
    pdf = F.softmax(unnormalized_pdf, dim=-1)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf[..., -1] = 1.0
    cdf = F.pad(cdf, pad=(1, 0), mode="constant", value=0.0)
    if inverse:
        k = searchsorted(cdf, inputs).unsqueeze(-1)
        edges = torch.linspace(0, 1, num_bins + 1).view([1] * inputs.dim() + [-1]).expand(*inputs.shape, -1)
        slopes = (cdf[..., 1:] - cdf[..., :-1]) / (edges[..., 1:] - edges[..., :-1])
        offsets = cdf[..., 1:] - slopes * edges[..., 1:]
        slope_k = slopes.gather(-1, k)[..., 0]
        outputs = torch.clamp((inputs - offsets.gather(-1, k)[..., 0]) / slope_k, 0, 1)
        logabsdet = -torch.log(slope_k)
        return outputs * (right - left) + left, logabsdet
    bin_pos = inputs * num_bins
    k = torch.floor(bin_pos).long()
    k[k >= num_bins] = num_bins - 1
    alpha = bin_pos - k.float()
    pdf_k = pdf.gather(-1, k[..., None])[..., 0]
    outputs = cdf.gather(-1, k[..., None])[..., 0]
    outputs += alpha * pdf_k
    outputs = torch.clamp(outputs, 0, 1)
    logabsdet = torch.log(pdf_k) - np.log(1.0 / num_bins)
    return outputs * (top - bottom) + bottom, logabsdet


def quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, inverse=False, left=0.0, right=1.0,
                     bottom=0.0, top=1.0, min_bin_width=1e-3, min_bin_height=1e-3):
    """splines/quadratic.py:55-159."""
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise OracleInputOutsideDomain()
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    widths = F.softmax(unnormalized_widths, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * num_bins) * widths
    uh = F.softplus(unnormalized_heights) + 1e-3
    if uh.shape[-1] == num_bins - 1:
        first_w = 0.5 * widths[..., 0]
        last_w = 0.5 * widths[..., -1]
        numerator = (0.5 * first_w * uh[..., 0] + 0.5 * last_w * uh[..., -1]
                     + torch.sum(((uh[..., :-1] + uh[..., 1:]) / 2) * widths[..., 1:-1], dim=-1))
        constant = (numerator / (1 - 0.5 * first_w - 0.5 * last_w))[..., None]
        uh = torch.cat([constant, uh, constant], dim=-1)
    area = torch.sum(((uh[..., :-1] + uh[..., 1:]) / 2) * widths, dim=-1)[..., None]
    heights = uh / area
    heights = min_bin_height + (1 - min_bin_height) * heights
    left_cdf = torch.cumsum(((heights[..., :-1] + heights[..., 1:]) / 2) * widths, dim=-1)
    left_cdf[..., -1] = 1.0
    left_cdf = F.pad(left_cdf, pad=(1, 0), mode="constant", value=0.0)
    locations = torch.cumsum(widths, dim=-1)
    locations[..., -1] = 1.0
    locations = F.pad(locations, pad=(1, 0), mode="constant", value=0.0)
    k = searchsorted(left_cdf if inverse else locations, inputs)[..., None]
    loc_k = locations.gather(-1, k)[..., 0]
    w_k = widths.gather(-1, k)[..., 0]
    cdf_k = left_cdf.gather(-1, k)[..., 0]
    hl = heights.gather(-1, k)[..., 0]
    hr = heights.gather(-1, k + 1)[..., 0]
    a = 0.5 * (hr - hl) * w_k
    b = hl * w_k
    c = cdf_k
    if inverse:
        c_ = c - inputs
        alpha = (-b + torch.sqrt(b.pow(2) - 4 * a * c_)) / (2 * a)
        outputs = torch.clamp(alpha * w_k + loc_k, 0, 1)
        logabsdet = -torch.log(alpha * (hr - hl) + hl)
        return outputs * (right - left) + left, logabsdet
    alpha = (inputs - loc_k) / w_k
    outputs = torch.clamp(a * alpha.pow(2) + b * alpha + c, 0, 1)
    logabsdet = torch.log(alpha * (hr - hl) + hl)
    return outputs * (top - bottom) + bottom, logabsdet


def cubic_spline(inputs, unnormalized_widths, unnormalized_heights, unnorm_derivatives_left,
                 unnorm_derivatives_right, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0,
                 min_bin_width=1e-3, min_bin_height=1e-3, eps=1e-5, quadratic_threshold=1e-3):
    """splines/cubic.py:63-267 (Blinn's closed-form inverse)."""
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise OracleInputOutsideDomain()
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)

    def normalise(u, floor):
        v = F.softmax(u, dim=-1)
        v = floor + (1 - floor * num_bins) * v
        cum = torch.cumsum(v, dim=-1)
        cum[..., -1] = 1
        return v, F.pad(cum, pad=(1, 0), mode="constant", value=0.0)

    widths, cumwidths = normalise(unnormalized_widths, min_bin_width)
    heights, cumheights = normalise(unnormalized_heights, min_bin_height)
    slopes = heights / widths
    m1 = torch.min(torch.abs(slopes[..., :-1]), torch.abs(slopes[..., 1:]))
    m2 = 0.5 * (widths[..., 1:] * slopes[..., :-1] + widths[..., :-1] * slopes[..., 1:]) / (
        widths[..., :-1] + widths[..., 1:])
    d_left = torch.sigmoid(unnorm_derivatives_left) * 3 * slopes[..., 0][..., None]
    d_right = torch.sigmoid(unnorm_derivatives_right) * 3 * slopes[..., -1][..., None]
    derivs = torch.min(m1, m2) * (torch.sign(slopes[..., :-1]) + torch.sign(slopes[..., 1:]))
    derivs = torch.cat([d_left, derivs, d_right], dim=-1)
    a = (derivs[..., :-1] + derivs[..., 1:] - 2 * slopes) / widths.pow(2)
    b = (3 * slopes - 2 * derivs[..., :-1] - derivs[..., 1:]) / widths
    c = derivs[..., :-1]
    d = cumheights[..., :-1]
    k = searchsorted(cumheights if inverse else cumwidths, inputs)[..., None]
    ia, ib, ic, id_ = (t.gather(-1, k)[..., 0] for t in (a, b, c, d))
    lw = cumwidths.gather(-1, k)[..., 0]
    rw = cumwidths.gather(-1, k + 1)[..., 0]
    if not inverse:
        s = inputs - lw
        outputs = ia * s.pow(3) + ib * s.pow(2) + ic * s + id_
        logabsdet = torch.log(3 * ia * s.pow(2) + 2 * ib * s + ic)
        return outputs * (top - bottom) + bottom, logabsdet
    b_ = (ib / ia) / 3.0
    c_ = (ic / ia) / 3.0
    d_ = (id_ - inputs) / ia
    delta_1 = -b_.pow(2) + c_
    delta_2 = -c_ * b_ + d_
    delta_3 = b_ * d_ - c_.pow(2)
    disc = 4.0 * delta_1 * delta_3 - delta_2.pow(2)
    dep1 = -2.0 * b_ * delta_1 + delta_2
    dep2 = delta_1
    three = disc >= 0
    one = disc < 0
    outputs = torch.zeros_like(inputs)

    def cbrt(v):
        return torch.sign(v) * torch.exp(torch.log(torch.abs(v)) / 3.0)

    p = cbrt((-dep1[one] + torch.sqrt(-disc[one])) / 2.0)
    q = cbrt((-dep1[one] - torch.sqrt(-disc[one])) / 2.0)
    outputs[one] = (p + q) - b_[one] + lw[one]
    theta = torch.atan2(torch.sqrt(disc[three]), -dep1[three])
    theta /= 3.0
    cr1, cr2 = torch.cos(theta), torch.sin(theta)
    r1 = cr1
    r2 = -0.5 * cr1 - 0.5 * math.sqrt(3) * cr2
    r3 = -0.5 * cr1 + 0.5 * math.sqrt(3) * cr2
    scale = 2 * torch.sqrt(-dep2[three])
    shift = -b_[three] + lw[three]
    r1, r2, r3 = r1 * scale + shift, r2 * scale + shift, r3 * scale + shift

    def in_bin(r):
        m = ((lw[three] - eps) < r).float()
        m *= (r < (rw[three] + eps)).float()
        return m

    roots = torch.stack([r1, r2, r3], dim=-1)
    masks = torch.stack([in_bin(r1), in_bin(r2), in_bin(r3)], dim=-1)
    pick = torch.argsort(masks, dim=-1, descending=True)[..., 0][..., None]
    outputs[three] = torch.gather(roots, dim=-1, index=pick).view(-1)
    quad = ia.abs() < quadratic_threshold
    qa, qb, qc = ib[quad], ic[quad], id_[quad] - inputs[quad]
    alpha = (-qb + torch.sqrt(qb.pow(2) - 4 * qa * qc)) / (2 * qa)
    outputs[quad] = alpha + lw[quad]
    s = outputs - lw
    logabsdet = -torch.log(3 * ia * s.pow(2) + 2 * ib * s + ic)
    return outputs * (right - left) + left, logabsdet


def _spline_rows(t, x, params, net, kind):
    """Slice the [.., multiplier] rows per spline kind, scaling as the coupling / AR classes do."""
    k = t.num_bins
    is_ar = hasattr(t, "autoregressive_net")
    has_h = hasattr(net, "hidden_features") if net is not None else False
    div = np.sqrt(net.hidden_features) if has_h else None
    if kind == "linear":
        return [params]
    if kind == "quadratic":
        uw, uh = params[..., :k], params[..., k:]
        if div is not None:
            uw /= div
            if not is_ar:  # autoregressive.py:430-432 scales the widths only
                uh /= div
        return [uw, uh]
    uw, uh = params[..., :k], params[..., k:2 * k]
    dl, dr = params[..., 2 * k][..., None], params[..., 2 * k + 1][..., None]
    if div is not None:
        uw /= div
        uh /= div
    return [uw, uh, dl, dr]


def _apply_sibling(t, x, plist, kind, inverse):
    fn = {"linear": linear_spline, "quadratic": quadratic_spline, "cubic": cubic_spline}[kind]
    kw = {}
    if kind != "linear":
        kw = dict(min_bin_width=getattr(t, "min_bin_width", 1e-3), min_bin_height=getattr(t, "min_bin_height", 1e-3))
    tails = getattr(t, "tails", None)
    if tails is None:
        return fn(x, *plist, inverse=inverse, **kw)
    return _unconstrained(fn, x, t.tail_bound, tails, plist, inverse=inverse, **kw)


def _make_sibling_coupling(kind):
    def ew(t, x, params, inverse):
        rows = _piecewise_rows(x, params)
        y, lad = _apply_sibling(t, x, _spline_rows(t, x, rows, t.transform_net, kind), kind, inverse)
        return y, sum_except_batch(lad)
    return lambda t, x, c, inv: _coupling(t, x, c, inv, ew)


def _make_sibling_ar(kind):
    def ew(t, x, params, inverse):
        rows = params.view(x.shape[0], t.features, -1)
        y, lad = _apply_sibling(t, x, _spline_rows(t, x, rows, t.autoregressive_net, kind), kind, inverse)
        return y, sum_except_batch(lad)
    return lambda t, x, c, inv: _autoregressive(t, x, c, inv, ew)


def _make_sibling_cdf(kind):
    def fn(t, x, c, inverse):
        n = x.shape[0]

        def share(p):
            return _p(p)[None, ...].expand(n, *p.shape).clone()

        if kind == "linear":
            plist = [share(t.unnormalized_pdf)]
            t_num_bins = t.unnormalized_pdf.shape[-1]
        elif kind == "quadratic":
            plist = [share(t.unnormalized_widths), share(t.unnormalized_heights)]
        else:
            plist = [share(t.unnormalized_widths), share(t.unnormalized_heights),
                     share(t.unnorm_derivatives_left), share(t.unnorm_derivatives_right)]
        y, lad = _apply_sibling(t, x, plist, kind, inverse)
        return y, sum_except_batch(lad)
    return fn


# ---- conditional.py (hyper-network transforms) -------------------------------------------------------------

def _conditional(fwd_inv):
    """conditional.py:73-85: params = conditional_net(context); dispatch to the given-params functions."""
    def fn(t, x, c, inverse):
        if c is None:
            raise TypeError("Conditional transforms require a context.")
        return fwd_inv(t, x, t.conditional_net(c), inverse)
    return fn


def _cond_shift(t, x, p, inverse):
    """conditional.py:172-190."""
    shift = p.view(-1, _int(t.features)).view(x.shape)
    return (x - shift if inverse else x + shift), x.new_zeros(x.shape[0])


def _cond_affine(t, x, p, inverse):
    """conditional.py:121-152 (``_epsilon`` is read but never set by the reference: supplied by the caller, see
    tests/golden/cases.py ``cond_affine_d5``)."""
    p = p.view(-1, _int(t.features), 2)
    scale = F.softplus(p[..., 0]) + t._epsilon
    shift = p[..., 1]
    lad = torch.log(scale).sum(-1)
    if inverse:
        return (x - shift) / scale, -lad
    return scale * x + shift, lad


def _cond_scale(t, x, p, inverse):
    """conditional.py:229-260."""
    scale = F.softplus(p.view(-1, _int(t.features))) + t.eps
    lad = torch.log(scale).sum(-1).view(x.shape[0])
    if inverse:
        return x / scale.view(x.shape), -lad
    return x * scale.view(x.shape), lad


def _int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def _cond_lu(t, x, p, inverse):
    """conditional.py:300-346."""
    f = _int(t.features)
    m = p.view(-1, f, f)
    sp = F.softplus(_p(t.scale_non_diag))
    eye = torch.eye(f, dtype=x.dtype)
    lower = sp * torch.tril(m, diagonal=-1) + eye
    upper = sp * torch.triu(m, diagonal=1) + torch.diag_embed(F.softplus(m.diagonal(0, -1, -2)) + t.eps)
    lad = upper.diagonal(0, -1, -2).log().sum(-1)
    if not inverse:
        return (lower @ (upper @ x.unsqueeze(-1))).view(x.shape), lad
    pivots = torch.arange(1, f + 1, dtype=torch.int32).unsqueeze(0)
    out = torch.linalg.lu_solve(torch.tril(lower, -1) + upper, torch.broadcast_to(pivots, x.shape), x.unsqueeze(-1))
    return out.view(x.shape), -lad


def _cond_rotation(t, x, p, inverse):
    """conditional.py:374-401."""
    c, s = torch.cos(p), torch.sin(p)
    mat = torch.cat([c, -s, s, c], -1).view(-1, 2, 2)
    if inverse:
        mat = mat.transpose(-2, -1)
    return (mat @ x.unsqueeze(-1)).squeeze(-1), x.new_zeros(x.shape[0])


def _cond_orthogonal(t, x, p, inverse):
    """conditional.py:424-452."""
    f = _int(t.features)
    q = p.view(-1, f, f)
    if inverse:
        q = q.flip(-2)
    return householder_apply(x, q), x.new_zeros(x.shape[0])


def _cond_svd(t, x, p, inverse):
    """conditional.py:480-543."""
    f = _int(t.features)
    sizes = [f * f, f * f, f] + ([f] if t.use_bias else [])
    parts = torch.split(p, sizes, -1)
    q_u, q_v, s_raw = parts[0].view(-1, f, f), parts[1].view(-1, f, f), parts[2]
    bias = parts[3] if t.use_bias else None
    if t.lipschitz_constant is not None:
        s = torch.sigmoid(s_raw) * (t.lipschitz_constant - t.eps) + t.eps
    else:
        s = torch.exp(s_raw) + t.eps
    if not inverse:
        out = householder_apply(householder_apply(x, q_v) * s, q_u)
        return (out + bias if t.use_bias else out), s.log().sum(-1)
    y = x - bias if t.use_bias else x
    out = householder_apply(householder_apply(y, q_u.flip(-2)) / s, q_v.flip(-2))
    return out, -s.log().sum(-1)


def _cond_linear_spline(t, x, p, inverse):
    """conditional.py:628-653."""
    rows = p.view(x.shape[0], _int(t.features), t.num_bins)
    y, lad = linear_spline(x, rows, inverse=inverse, left=-4.0, right=4.0, bottom=-4.0, top=4.0)
    return y, sum_except_batch(lad)


def _cond_rq(t, x, p, inverse):
    """conditional.py:693-743."""
    b, d = x.shape
    rows = p.view(b, d, p.shape[1] // d)
    y, lad = rq_from_rows(x, rows, t.num_bins, t.tails, t.tail_bound, inverse,
                          wh_divisor=_net_divisor(t.conditional_net) if hasattr(t.conditional_net, "hidden_features") else None,
                          box=(-1.2, 1.2, -1.2, 1.2), enable_identity_init=True,
                          mins=(t.min_bin_width, t.min_bin_height, t.min_derivative))
    return y, sum_except_batch(lad)


def _cond_sos(t, x, p, inverse):
    """conditional.py:770-787."""
    s, f = t.n_sigmoids, _int(t.features)
    raw = p.view(x.shape[0], f, 3 * s + 1)
    sp, ls, rs, es = torch.split(raw, [s, s, s, 1], dim=-1)
    es = es.reshape(-1, f)

    def fwd(v):
        return sos_forward(v, sp, ls, rs, es)

    if inverse:
        return monotonic_inverse(fwd, x, 120, 50)
    return fwd(x)


def _cond_planar(t, x, p, inverse):
    """conditional.py:824-865 (forward only)."""
    if inverse:
        raise NotImplementedError()
    f = _int(t.features)
    b_ = p[..., -1:]
    vals = p[..., :-1].view(-1, f, 2)
    u_, w_ = vals[..., 0][:, None, :], vals[..., 1][:, None, :]
    wtu = torch.bmm(u_, w_.transpose(-2, -1))
    u_ = u_ + (-1 + torch.log(1 + torch.exp(wtu)) - wtu) * w_ / torch.norm(w_, p=2, dim=-1, keepdim=True) ** 2
    pre = torch.bmm(x.view(-1, 1, f), w_.transpose(-2, -1)).squeeze(-1).squeeze(-1) + b_.squeeze(-1)
    out = x + (u_ * torch.tanh(pre).view(-1, 1, 1)).squeeze(1)
    psi = (1 - torch.tanh(pre) ** 2).view(-1, 1, 1) * w_
    abs_det = (1 + torch.bmm(u_, psi.transpose(-2, -1))).abs()
    return out, torch.log(1e-7 + abs_det).reshape(-1)


def _cond_sylvester(t, x, p, inverse):
    """conditional.py:925-989, with Q = the full per-sample Householder product (D-general; equals the
    reference's two-row construction for features == 2)."""
    if inverse:
        raise NotImplementedError()
    f = _int(t.features)
    r_full, r2_diag, q_raw, b = torch.split(p, [f * f, f, f * f, f], -1)
    r_full = r_full.reshape(-1, f, f)
    mask = torch.triu(torch.ones(f, f, dtype=x.dtype), diagonal=1).unsqueeze(0)
    r1 = r_full * mask + torch.diag_embed(torch.tanh(torch.diagonal(r_full, dim1=-2, dim2=-1)))
    r2 = r_full.transpose(-2, -1) * mask + torch.diag_embed(torch.tanh(r2_diag))
    return sylvester_forward(x, q_raw.reshape(-1, f, f), r1, r2, b.reshape(-1, f))

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
    "HouseholderSequence": _householder,
    "ParametrizedHouseHolder": _householder,
    "PlanarTransform": _planar,
    "SylvesterTransform": _sylvester,
    "LULinear": _lu_linear,
    "ActNorm": _actnorm,
    "BatchNorm": _batchnorm,
    "Exp": _exp,
    "Tanh": _tanh,
    "LogTanh": _logtanh,
    "LeakyReLU": _leaky_relu,
    "Sigmoid": _sigmoid,
    "Softplus": _softplus_t,
    "CauchyCDF": _cauchy,
    "GatedLinearUnit": _glu,
    "Logit": _named_inverse(None),
    "CauchyCDFInverse": _named_inverse(None),
    "PiecewiseLinearCouplingTransform": _make_sibling_coupling("linear"),
    "PiecewiseQuadraticCouplingTransform": _make_sibling_coupling("quadratic"),
    "PiecewiseCubicCouplingTransform": _make_sibling_coupling("cubic"),
    "MaskedPiecewiseLinearAutoregressiveTransform": _make_sibling_ar("linear"),
    "MaskedPiecewiseQuadraticAutoregressiveTransform": _make_sibling_ar("quadratic"),
    "MaskedPiecewiseCubicAutoregressiveTransform": _make_sibling_ar("cubic"),
    "PiecewiseLinearCDF": _make_sibling_cdf("linear"),
    "PiecewiseQuadraticCDF": _make_sibling_cdf("quadratic"),
    "PiecewiseCubicCDF": _make_sibling_cdf("cubic"),
    "AffineConditionalTransform": _conditional(_cond_affine),
    "ConditionalShiftTransform": _conditional(_cond_shift),
    "ConditionalScaleTransform": _conditional(_cond_scale),
    "ConditionalLUTransform": _conditional(_cond_lu),
    "ConditionalRotationTransform": _conditional(_cond_rotation),
    "ConditionalOrthogonalTransform": _conditional(_cond_orthogonal),
    "ConditionalSVDTransform": _conditional(_cond_svd),
    "PiecewiseLinearConditionalTransform": _conditional(_cond_linear_spline),
    "ConditionalPiecewiseRationalQuadraticTransform": _conditional(_cond_rq),
    "ConditionalSumOfSigmoidsTransform": _conditional(_cond_sos),
    "ConditionalPlanarTransform": _conditional(_cond_planar),
    "ConditionalSylvesterTransform": _conditional(_cond_sylvester),
    "SumOfSigmoids": _sum_of_sigmoids,
    "MaskedSumOfSigmoidsTransform": lambda t, x, c, inv: _autoregressive(t, x, c, inv, _ew_sos_ar),
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
