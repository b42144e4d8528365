"""Hyper-network ("conditional") transforms: per-sample bijector parameters come from a network on
the context instead of from the inputs (API of flowcon/transforms/conditional.py:23-989).

The conditional network is a PyTorch-ROCm module (ResidualNet / MLP); every bijector application is the
same HIP kernel as its coupling / autoregressive sibling, fed per-sample parameter rows.
"""

import numpy as np
import torch
from torch.nn import functional as F

from flowconductor_amd import ops, options
from flowconductor_amd.nn.nets import MLP, ResidualNet
from flowconductor_amd.transforms.coupling import _is_plain_resnet
from flowconductor_amd.transforms.base import Transform
from flowconductor_amd.transforms.orthogonal import ParametrizedHouseHolder


class ConditionalTransform(Transform):
    """Transforms each input variable with an invertible transformation, conditioned on some given input."""

    def __init__(self, features, hidden_features=64, context_features=1, num_blocks=2, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 conditional_net: torch.nn.Module = None):
        super().__init__()
        self.features = features
        if conditional_net is not None:
            assert isinstance(conditional_net, torch.nn.Module)
            self.conditional_net = conditional_net
        else:
            self.conditional_net = self.set_default_network(activation, context_features, dropout_probability,
                                                            hidden_features, num_blocks, use_batch_norm,
                                                            use_residual_blocks)

    def set_default_network(self, activation, context_features, dropout_probability, hidden_features, num_blocks,
                            use_batch_norm, use_residual_blocks):
        if use_residual_blocks:
            return ResidualNet(in_features=context_features, out_features=self._num_parameters(),
                               hidden_features=hidden_features, activation=activation, num_blocks=num_blocks,
                               dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        if dropout_probability > 1e-12:
            raise NotImplementedError("No dropout for MLP")
        if use_batch_norm:
            raise NotImplementedError("No batch norm for MLP")
        return MLP(in_shape=(context_features,), out_shape=(self._num_parameters(),),
                   hidden_sizes=[hidden_features] * num_blocks)

    def _num_parameters(self):
        return self.features * self._output_dim_multiplier()

    def forward(self, inputs, context=None):
        if context is None:
            raise TypeError("Conditional transforms require a context.")
        return self._forward_given_params(inputs, self._conditional_params(context))

    def inverse(self, inputs, context=None):
        if context is None:
            raise TypeError("Conditional transforms require a context.")
        return self._inverse_given_params(inputs, self._conditional_params(context))

    # ---- the hyper-network on the matrix cores (inference) ------------------------------------------------------
    def _hip_hidden_ok(self, context):
        net = self.conditional_net
        return (_is_plain_resnet(net) and context.dim() == 2 and context.is_cuda
                and context.dtype == torch.float32 and context.shape[0] >= ops.HIDDEN_ROWS
                and options.get("fused_hidden") and net.hip_hidden_supported(context.shape[1])
                and not ops.has_hooks(net)
                and not (torch.is_grad_enabled()
                         and (context.requires_grad or any(p.requires_grad for p in net.parameters()))))

    def _hidden(self, context):
        """[N, 64] hidden activation of the conditional ResidualNet by ``fc_resnet_hidden`` (its input rows are the
        context rows; zero-padded columns for a narrower net); the leftover < 16 rows on PyTorch."""
        net = self.conditional_net
        n = context.shape[0]
        ids = getattr(self, "_ctx_cols", None)
        if ids is None or ids.device != context.device or ids.numel() != context.shape[1]:
            ids = self._ctx_cols = torch.arange(context.shape[1], dtype=torch.int32, device=context.device)
        body = n - n % ops.HIDDEN_ROWS
        hidden = net.hidden_hip(context[:body], ids)
        if body < n:
            hidden = torch.cat((hidden, net.hidden_padded(context[body:])))
        return hidden

    def _conditional_params(self, context):
        """``conditional_net(context)``; a ResidualNet the hidden-layer kernel covers runs its hidden stack there."""
        if self._hip_hidden_ok(context):
            return self.conditional_net.final_from_padded(self._hidden(context))
        return self.conditional_net(context)

    def _output_dim_multiplier(self):
        raise NotImplementedError()

    def _forward_given_params(self, inputs, autoregressive_params):
        raise NotImplementedError()

    def _inverse_given_params(self, inputs, autoregressive_params):
        raise NotImplementedError()


class AffineConditionalTransform(ConditionalTransform):
    """y = (softplus(u(context)) + eps) x + shift(context), parameters interleaved [N, F, (u, shift)] (conditional.py:98-152).

    The reference's class reads ``self._epsilon`` in both directions but never sets it (its constructor, :99-119, only
    forwards to the base class): as shipped, its first forward raises AttributeError.  Here ``_epsilon`` is 1e-3, the
    value of the identical bijector under the autoregressive conditioner (autoregressive.py:89); the golden fixture
    ``cond_affine_d5`` comes from the reference's own code with that one attribute supplied (tests/golden/cases.py)."""

    _epsilon = 1e-3

    def __init__(self, features, hidden_features, context_features, **kwargs):
        self.features = features
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         **kwargs)

    def _output_dim_multiplier(self):
        return 2

    def _check_epsilon(self):
        if self._epsilon != 1e-3:       # the kernel's constant (fc_affine.hip, FC_AFFINE_MAF_SOFTPLUS)
            raise ValueError("AffineConditionalTransform: the HIP bijector is built for _epsilon = 1e-3")

    def _forward_given_params(self, inputs, conditional_params):
        self._check_epsilon()
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_MAF_SOFTPLUS)

    def _inverse_given_params(self, inputs, conditional_params):
        self._check_epsilon()
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_MAF_SOFTPLUS, inverse=True)


class ConditionalShiftTransform(ConditionalTransform):
    """y = x + shift(context); logabsdet = 0 (conditional.py:155-209)."""

    def __init__(self, features, hidden_features, context_features, **kwargs):
        self.features = features
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         **kwargs)

    def _output_dim_multiplier(self):
        return 1

    def _forward_given_params(self, inputs, conditional_params):
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_ADDITIVE)

    def _inverse_given_params(self, inputs, conditional_params):
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_ADDITIVE, inverse=True)


class ConditionalScaleTransform(ConditionalTransform):
    """y = x * (softplus(p(context)) + 1e-5) (conditional.py:212-272)."""

    def __init__(self, features, hidden_features, context_features, **kwargs):
        self.features = features
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         **kwargs)
        self.eps = 1e-5

    def _output_dim_multiplier(self):
        return 1

    def _forward_given_params(self, inputs, conditional_params):
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_SCALE_SOFTPLUS)

    def _inverse_given_params(self, inputs, conditional_params):
        return ops.affine_coupling(inputs, conditional_params, None, activation=ops.AFFINE_SCALE_SOFTPLUS,
                                   inverse=True)


class ConditionalLUTransform(ConditionalTransform):
    """Per-sample W = L U from one raw [F, F] block per sample (conditional.py:275-346):
    L = sp * tril(M, -1) + I, U = sp * triu(M, 1) + diag(softplus(diag M) + eps), sp = softplus(scale_non_diag)."""

    def __init__(self, features, hidden_features, context_features, num_blocks=2, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False, eps=1e-7):
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        self.eps = eps
        self.lower_indices = np.tril_indices(features, k=-1)
        self.upper_indices = np.triu_indices(features, k=1)
        self.diag_indices = np.diag_indices(features)
        self.softplus = torch.nn.Softplus()
        self.diag_entries = torch.nn.Parameter(torch.eye(features), requires_grad=False)
        self.default_pivot = torch.nn.Parameter(torch.arange(1, self.features + 1, dtype=torch.int32).unsqueeze(0),
                                                requires_grad=False)
        self.scale_non_diag = torch.nn.Parameter(- 2 * torch.ones(()), requires_grad=True)
        self._sp_cache = None

    def _output_dim_multiplier(self):
        return self.features

    def _offdiag_scale(self):
        # softplus of a 0-dim parameter: read back once per value (it only changes when trained)
        key = ops.cache_key(self.scale_non_diag)
        if self._sp_cache is None or self._sp_cache[0] != key:
            self._sp_cache = (key, float(F.softplus(self.scale_non_diag.detach())))
        return self._sp_cache[1]

    def _forward_given_params(self, inputs, conditional_params):
        return ops.linear_per_sample(inputs, conditional_params, mode=ops.PER_SAMPLE_LU_FORWARD,
                                     offdiag_scale=self._offdiag_scale(), eps=self.eps, want_logabsdet=True)

    def _inverse_given_params(self, inputs, conditional_params):
        return ops.linear_per_sample(inputs, conditional_params, mode=ops.PER_SAMPLE_LU_INVERSE,
                                     offdiag_scale=self._offdiag_scale(), eps=self.eps, want_logabsdet=True)

    def _unconstrained_entries(self, conditional_params):
        return conditional_params.view(-1, self.features, self._output_dim_multiplier())

    def _create_lower_upper(self, conditional_params):
        """Dense per-sample (L, U) as tensors (API parity; the kernel builds them on the fly)."""
        m = self._unconstrained_entries(conditional_params)
        sp = F.softplus(self.scale_non_diag)
        lower = sp * torch.tril(m, diagonal=-1) + self.diag_entries
        upper_diag = torch.diag_embed(self.softplus(m.diagonal(0, -1, -2)) + self.eps)
        upper = sp * torch.triu(m, diagonal=1) + upper_diag
        return lower, upper


class ConditionalRotationTransform(ConditionalTransform):
    """2-D rotation by theta(context) (conditional.py:349-401)."""

    def _output_dim_multiplier(self):
        pass

    def __init__(self, features, hidden_features, context_features, num_blocks=1, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        assert features == 2, "Only available for 2D rotations."
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)

    def _num_parameters(self):
        return 1

    def build_matrix(self, conditional_params):
        theta = conditional_params
        c, s = torch.cos(theta), torch.sin(theta)
        return torch.cat([c, -s, s, c], -1).view(-1, self.features, self.features)

    def _forward_given_params(self, inputs, conditional_params):
        outputs = ops.linear_per_sample(inputs, self.build_matrix(conditional_params), mode=ops.PER_SAMPLE_DENSE)
        return outputs, inputs.new_zeros(inputs.shape[0])

    def _inverse_given_params(self, inputs, conditional_params):
        outputs = ops.linear_per_sample(inputs, self.build_matrix(conditional_params), mode=ops.PER_SAMPLE_DENSE_T)
        return outputs, inputs.new_zeros(inputs.shape[0])


class ConditionalOrthogonalTransform(ConditionalTransform):
    """F per-sample Householder reflections (conditional.py:404-452)."""

    def __init__(self, features, hidden_features, context_features, num_blocks=2, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)

    def _output_dim_multiplier(self):
        return self.features

    def _forward_given_params(self, inputs, conditional_params):
        return self._get_matrices(conditional_params).forward(inputs)

    def _inverse_given_params(self, inputs, conditional_params):
        outputs, logabsdet = self._get_matrices(conditional_params).inverse(inputs)
        return outputs.squeeze(), logabsdet

    def _get_matrices(self, conditional_params) -> ParametrizedHouseHolder:
        return ParametrizedHouseHolder(self._unconstrained_entries(conditional_params))

    def _unconstrained_entries(self, conditional_params):
        return conditional_params.view(-1, self.features, self._output_dim_multiplier())


class ConditionalSVDTransform(ConditionalTransform):
    """y = U diag(S) V^T x + b with per-sample Householder U, V and positive S (conditional.py:455-543)."""

    def __init__(self, features, hidden_features, context_features, num_blocks=2, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False, use_bias=True, eps=1e-3,
                 lipschitz_constant_limit=None):
        self.use_bias = use_bias
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        self.eps = eps
        self.lipschitz_constant = lipschitz_constant_limit
        self._epsilon = 1e-2

    def _output_dim_multiplier(self):
        return self.features * 2 + 1 + (1 if self.use_bias else 0)

    def _forward_given_params(self, inputs, conditional_params):
        q_u, q_v, s, bias = self._parts(conditional_params)
        vtx = ops.householder(inputs, q_v)
        shift = bias if bias is not None else torch.zeros_like(s)
        svtx, _ = ops.affine_coupling(vtx, torch.cat((torch.zeros_like(s), s), dim=1), None,
                                      activation=ops.AFFINE_SCALE_GIVEN)
        usvtx = ops.householder(svtx, q_u)
        if bias is not None:
            usvtx, _ = ops.affine_coupling(usvtx, shift, None, activation=ops.AFFINE_ADDITIVE)
        return usvtx, s.log().sum(-1)

    def _inverse_given_params(self, inputs, conditional_params):
        q_u, q_v, s, bias = self._parts(conditional_params)
        y = inputs
        if bias is not None:
            y, _ = ops.affine_coupling(y, bias, None, activation=ops.AFFINE_ADDITIVE, inverse=True)
        uty = ops.householder(y, q_u, reverse=True)
        sinv, _ = ops.affine_coupling(uty, torch.cat((torch.zeros_like(s), s), dim=1), None,
                                      activation=ops.AFFINE_SCALE_GIVEN, inverse=True)
        outputs = ops.householder(sinv, q_v, reverse=True)
        return outputs.squeeze(), -s.log().sum(-1)

    def _parts(self, conditional_params):
        f = self.features
        sizes = [f * f, f * f, f] + ([f] if self.use_bias else [])
        parts = torch.split(conditional_params, sizes, -1)
        q_u, q_v, s_raw = parts[0].reshape(-1, f, f), parts[1].reshape(-1, f, f), parts[2]
        bias = parts[3].contiguous() if self.use_bias else None
        if self.lipschitz_constant is not None:
            s = torch.sigmoid(s_raw) * (self.lipschitz_constant - self.eps) + self.eps
        else:
            s = torch.exp(s_raw) + self.eps
        return q_u.contiguous(), q_v.contiguous(), s.contiguous(), bias


class PiecewiseLinearConditionalTransform(ConditionalTransform):
    """Piecewise-linear spline on the box [-4, 4]^2 (conditional.py:606-653)."""

    def __init__(self, num_bins, features, hidden_features, context_features, num_blocks=2,
                 use_residual_blocks=True, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        self.num_bins = num_bins
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)

    def _output_dim_multiplier(self):
        return self.num_bins

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        return ops.piecewise_spline_autograd(inputs, autoregressive_params, None, kind=ops.SPLINE_LINEAR,
                                    num_bins=self.num_bins, left=-4.0, right=4.0, bottom=-4.0, top=4.0,
                                    inverse=inverse)

    def _forward_given_params(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _inverse_given_params(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)


class ConditionalPiecewiseRationalQuadraticTransform(ConditionalTransform):
    """RQ spline with per-sample parameters from the context (conditional.py:656-743)."""

    def __init__(self, features, hidden_features, context_features, num_bins=10, tails=None, tail_bound=1.0,
                 num_blocks=2, use_residual_blocks=True, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT, min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)

    def _output_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        elif self.tails is None:
            return self.num_bins * 3 + 1
        else:
            raise ValueError

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        divisor = 1.0
        if hasattr(self.conditional_net, "hidden_features"):
            divisor = float(np.sqrt(self.conditional_net.hidden_features))
        return ops.rq_spline_autograd(inputs, autoregressive_params, None, num_bins=self.num_bins, tails=self.tails,
                                      tail_bound=self.tail_bound, left=-1.2, right=1.2, bottom=-1.2, top=1.2,
                                      min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                      min_derivative=self.min_derivative, enable_identity_init=True,
                                      wh_divisor=divisor, inverse=inverse)

    def _forward_given_params(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _inverse_given_params(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)

    # <= 32 features, ResidualNet(hidden <= 64): final Linear + spline in one kernel, both directions in one pass (the
    # parameters depend on the context only) -- fc_rq_spline_fused_linear at K = 8 with linear tails,
    # fc_rq_spline_fused_general for the other shapes (the constructor's defaults num_bins = 10, tails = None included)
    def _fused_mode(self, inputs, context):
        """None, "k8" or "general"."""
        if not (context is not None and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and inputs.shape[0] == context.shape[0] and options.get("fused_final_layer")
                and self._hip_hidden_ok(context) and not (torch.is_grad_enabled() and inputs.requires_grad)):
            return None
        n, d = inputs.shape
        hidden = self.conditional_net.hidden_features
        if ops.fused_linear_supported(n, d, d, hidden, self.num_bins, self.tails):
            return "k8"
        if hidden <= 64 and ops.fused_general_supported(n, d, d, 64, self.num_bins, self.tails):
            return "general"
        return None

    def _fused_ok(self, inputs, context):
        return self._fused_mode(inputs, context) is not None

    def _fused(self, inputs, context, inverse):
        mode = self._fused_mode(inputs, context)
        net = self.conditional_net
        lin = net.final_layer
        key = (ops.cache_key(lin.weight, lin.bias), mode)
        if getattr(self, "_packed", None) is None or self._packed[0] != key:
            if mode == "k8":
                packed = ops.pack_final_layer(lin.weight, lin.bias, self.num_bins)
            else:
                packed = ops.pack_final_layer_general(lin.weight, lin.bias, self.num_bins, self.tails, 64)
            cols = torch.arange(self.features, dtype=torch.int32, device=lin.weight.device)
            self._packed = (key,) + tuple(packed) + (cols,)
        hidden = self._hidden(context)
        kw = dict(num_bins=self.num_bins, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                  min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                  wh_divisor=float(np.sqrt(net.hidden_features)), enable_identity_init=True, inverse=inverse)
        n = inputs.shape[0]
        body = n - n % ops.FUSED_ROWS
        if mode == "k8":
            outputs, logabsdet = ops.rq_spline_fused_linear(inputs[:body], hidden[:body], *self._packed[1:], **kw)
        else:
            outputs, logabsdet = ops.rq_spline_fused_general(inputs[:body], hidden[:body], *self._packed[1:], tails=self.tails,
                                                             left=-1.2, right=1.2, bottom=-1.2, top=1.2, **kw)
        if body < n:
            out_b, lad_b = self._elementwise(inputs[body:].contiguous(), net.final_from_padded(hidden[body:]), inverse)
            outputs, logabsdet = torch.cat((outputs, out_b)), torch.cat((logabsdet, lad_b))
        return outputs, logabsdet

    def forward(self, inputs, context=None):
        if context is not None and self._fused_ok(inputs, context):
            return self._fused(inputs, context, False)
        return super().forward(inputs, context)

    def inverse(self, inputs, context=None):
        if context is not None and self._fused_ok(inputs, context):
            return self._fused(inputs, context, True)
        return super().inverse(inputs, context)


class ConditionalSumOfSigmoidsTransform(ConditionalTransform):
    """Sum-of-sigmoids with per-sample raw parameters (conditional.py:746-787); no +-0.5 offset here."""

    def __init__(self, features, hidden_features, context_features, n_sigmoids=10, num_blocks=2,
                 use_residual_blocks=True, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        self.n_sigmoids = n_sigmoids
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)

    def _output_dim_multiplier(self):
        return 3 * self.n_sigmoids + 1

    def _forward_given_params(self, inputs, autoregressive_params):
        return ops.sum_of_sigmoids_autograd(inputs, autoregressive_params, self.n_sigmoids, inverse=False)

    def _inverse_given_params(self, inputs, autoregressive_params):
        return ops.sum_of_sigmoids_autograd(inputs, autoregressive_params, self.n_sigmoids, inverse=True)


class ConditionalPlanarTransform(ConditionalTransform):
    """Planar flow with per-sample (u, w, b) (conditional.py:790-865); forward only."""

    def __init__(self, features, hidden_features, context_features, num_blocks=2, use_residual_blocks=True,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        self.softplus = torch.nn.Softplus()
        # (the reference allocates this buffer on "cuda" unconditionally, conditional.py:814)
        self.diag_entries = torch.nn.Parameter(torch.eye(features), requires_grad=False)
        self.default_pivot = torch.nn.Parameter(torch.arange(1, self.features + 1, dtype=torch.int32).unsqueeze(0),
                                                requires_grad=False)

    def _num_parameters(self):
        return self.features * self._output_dim_multiplier() + self._constant_dim_addition()

    def _output_dim_multiplier(self):
        return 2

    def _constant_dim_addition(self):
        return 1

    def _forward_given_params(self, inputs, conditional_params):
        u_, w_, b_ = self._create_uwb(conditional_params)
        return ops.planar(inputs, w_.reshape(-1, self.features), u_.reshape(-1, self.features), b_.reshape(-1),
                          per_sample=True)

    def _inverse_given_params(self, inputs, conditional_params):
        raise NotImplementedError()

    def get_u_hat(self, _u, _w):
        """Enforce w^T u >= -1 per sample (sufficient for invertibility with tanh)."""
        wtu = torch.bmm(_u, torch.transpose(_w, dim0=-2, dim1=-1))
        m_wtu = -1 + torch.log(1 + torch.exp(wtu))
        return _u + (m_wtu - wtu) * _w / torch.norm(_w, p=2, dim=-1, keepdim=True) ** 2

    def _create_uwb(self, conditional_params):
        _b = conditional_params[..., -1:]
        vals = conditional_params[..., :-1].view(-1, self.features, self._output_dim_multiplier())
        _u, _w = vals[..., 0][:, None, :], vals[..., 1][:, None, :]
        return self.get_u_hat(_u, _w), _w, _b


class ConditionalSylvesterTransform(ConditionalTransform):
    """Sylvester flow with per-sample (R1, R2, Q, b) (conditional.py:876-989); forward only.

    The reference builds Q from two hard-coded identity rows and therefore only works for
    ``features == 2`` (conditional.py:970-975).  This class is D-general: Q is the product of the F
    per-sample Householder reflections applied inside the fused kernel, which coincides with the
    reference for F = 2."""

    def __init__(self, features: int, hidden_features: int, context_features: int, num_blocks=2,
                 use_residual_blocks=True, activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 eps=1e-3):
        self.eps = eps
        self.features = features
        self._splits = self._output_splits()
        super().__init__(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, use_residual_blocks=use_residual_blocks, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        self.features = torch.nn.Parameter(torch.tensor(features), requires_grad=False)
        self._f = features
        self.reverse_idx = torch.nn.Parameter(torch.arange(features - 1, -1, -1), requires_grad=False)
        self.triu_mask = torch.nn.Parameter(torch.triu(torch.ones(features, features).unsqueeze(0), diagonal=1),
                                            requires_grad=False)
        self.identity = torch.nn.Parameter(torch.eye(features, features).unsqueeze(0), requires_grad=False)

    def _output_splits(self):
        f = int(self.features)
        return [f ** 2, f, f ** 2, f]  # R_full, R2 diag, q vectors, bias

    def _num_parameters(self):
        return sum(self._splits)

    def _forward_given_params(self, inputs, conditional_params):
        r1, r2, q_vectors, bias = self._create_mats(conditional_params)
        return ops.sylvester(inputs, q_vectors, r1, r2, bias)

    def _inverse_given_params(self, inputs, conditional_params):
        raise NotImplementedError()

    def _create_mats(self, conditional_params):
        f = self._f
        r_full, r2_diag, q_raw, b = torch.split(conditional_params, self._splits, -1)
        r1, r2 = self._create_upper(r_full.reshape(-1, f, f), r2_diag, self.triu_mask)
        return r1, r2, q_raw.reshape(-1, f, f).contiguous(), b.reshape(-1, f).contiguous()

    @staticmethod
    def _create_upper(full_matr_r, diag_vals, triu_mask):
        masked_1 = full_matr_r * triu_mask
        masked_2 = torch.transpose(full_matr_r, dim0=-2, dim1=-1) * triu_mask
        diag_1 = torch.diag_embed(torch.tanh(torch.diagonal(full_matr_r, dim1=-2, dim2=-1)), dim1=-2, dim2=-1)
        diag_2 = torch.diag_embed(torch.tanh(diag_vals), dim1=-2, dim2=-1)
        return masked_1 + diag_1, masked_2 + diag_2
