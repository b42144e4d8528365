"""Masked autoregressive transforms (API of flowcon/transforms/autoregressive/autoregressive.py).

``forward`` = one MADE pass + one HIP element-wise kernel; ``inverse`` = D sequential passes
(reference autoregressive.py:39-53), each of which here only computes what it fixes: the
hidden stack, the final-layer rows of dim d and the element-wise inverse of column d.
"""

import numpy as np
import torch
from torch.nn import functional as F

from flowconductor_amd import ops, options
from flowconductor_amd.transforms import made as made_module
from flowconductor_amd.transforms.base import Transform


class AutoregressiveTransform(Transform):
    """Element-wise bijector whose parameters for dim i depend on dims < i only.

    NOTE: the inverse costs D conditioner passes (D = number of input features).
    """

    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def _hidden(self, inputs, context):
        """Everything before the conditioner's final layer, or None when the net is not a MADE-like module with
        ``hidden`` / ``final_layer``.  A MADE whose hidden stack ``fc_resnet_hidden`` covers (hidden <= 64, residual
        blocks, a known activation, optionally an additive context) runs it there on pre-masked weights -- inference only; the result then has the
        kernel's 64 columns (zeros beyond a narrower MADE's width; ``_final`` accounts for them)."""
        net = self.autoregressive_net
        if not (hasattr(net, "hidden") and hasattr(net, "final_layer")):
            return None
        n = inputs.shape[0]
        if (inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and n >= ops.HIDDEN_ROWS and options.get("fused_hidden")
                and getattr(net, "hip_hidden_supported", None) is not None and net.hip_hidden_supported(context)
                and not ops.has_hooks(net) and not self._needs_grad(inputs)):
            body = n - n % ops.HIDDEN_ROWS
            hidden = net.hidden_hip(inputs[:body].contiguous(), None if context is None else context[:body])
            if body < n:
                tail = net.hidden(inputs[body:], None if context is None else context[body:])
                if tail.shape[1] != hidden.shape[1]:
                    tail = F.pad(tail, (0, hidden.shape[1] - tail.shape[1]))
                hidden = torch.cat((hidden, tail))
            return hidden
        return net.hidden(inputs, context)

    def _final(self, hidden, rows=None):
        """Final layer on ``hidden`` ([N, H], or [N, 64] zero-padded from the kernel); ``rows``: a slice of its
        outputs (the parameters of one dim)."""
        net = self.autoregressive_net
        final = net.final_layer
        if hidden.shape[1] == final.in_features and rows is None:
            return final(hidden)
        if hasattr(net, "masked_final"):
            weight, bias = net.masked_final(hidden.shape[1])
        else:
            weight, bias = final.weight, final.bias
        if rows is not None:
            weight, bias = weight[rows], None if bias is None else bias[rows]
        return F.linear(hidden, weight, bias)

    def _needs_grad(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad
                                            or any(p.requires_grad for p in self.autoregressive_net.parameters()))

    def _conditioner(self, inputs, context):
        """``autoregressive_net(inputs, context)``."""
        hidden = self._hidden(inputs, context) if inputs.dim() == 2 else None
        if hidden is None:
            return self.autoregressive_net(inputs, context)
        return self._final(hidden)

    def forward(self, inputs, context=None):
        autoregressive_params = self._conditioner(inputs, context)
        return self._elementwise_forward(inputs, autoregressive_params)

    def inverse(self, inputs, context=None):
        if self._device_loop_ok(inputs, context):
            return self._inverse_device_loop(inputs)
        if self._incremental_ok(inputs):
            return self._inverse_incremental(inputs, context)
        num_inputs = int(np.prod(inputs.shape[1:]))
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        # Pass p fixes column p (autoregressive.py:44-53); what it leaves in the later columns is the inverse under parameters
        # made from unfinished inputs -- harmless for the reference, whose masked GEMMs multiply it by exact zeros, but the
        # hidden-layer kernel scales every ROW by its maximum before the f16 split, and with an affine form those values can
        # grow by 1 / scale per pass and drown the finished columns.  They are never read again by a dim that is kept, so
        # they go back to zero (2-D inputs: the column order is the feature order).
        clear_rest = inputs.dim() == 2 and not self._needs_grad(inputs)
        with ops.deferred_errors():
            for p in range(num_inputs):
                autoregressive_params = self._conditioner(outputs, context)
                outputs, logabsdet = self._elementwise_inverse(inputs, autoregressive_params)
                if clear_rest and p + 1 < num_inputs:
                    outputs[:, p + 1:] = 0
        return outputs, logabsdet

    # ---- the D passes on the device (round 4) -----------------------------------------------------------------------
    def _device_loop_form(self):
        """``(kind, parameters per dim, spline keyword arguments)`` of the element-wise inverse ``fc_made_inverse`` knows,
        or None (the other forms keep the host loop)."""
        return None

    def _device_loop_ok(self, inputs, context):
        """One kernel for the whole inverse (``fc_made_inverse``): a residual-block MADE with hidden <= 64, <= 3 ReLU
        blocks, no context / batch norm / active dropout / hooks, D <= 64, float32 rows on the device, inference only."""
        net = self.autoregressive_net
        if not (context is None and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and 1 < inputs.shape[1] <= 64 and inputs.shape[0] >= 1 and options.get("ar_device_loop")
                and options.get("fused_hidden") and not self._needs_grad(inputs)):
            return False

        def structure_ok():
            form = self._device_loop_form()
            code = ops.activation_code(net.activation) if isinstance(net, made_module.MADE) else None
            return (form is not None and isinstance(net, made_module.MADE) and inputs.shape[1] == net.initial_layer.in_features
                    and not hasattr(net, "context_layer") and len(net.blocks) <= 3 and net.hip_hidden_supported(None)
                    and code is not None and code[0] == ops.ACT_RELU and form[1] <= 48
                    and net.final_layer.out_features == form[1] * inputs.shape[1])

        return (ops.static_memo(self, "_fc_device_loop_ok", (inputs.shape[1],) + ops.structure_key(net), structure_ok)
                and not ops.has_hooks(net))

    def _inverse_device_loop(self, inputs):
        net = self.autoregressive_net
        kind, per_dim, rq = self._device_loop_form()
        features = inputs.shape[1]
        layers = [net.initial_layer, net.final_layer] + [lin for block in net.blocks for lin in block.linear_layers]
        key = ops.cache_key(*[t for lin in layers for t in (lin.weight, lin.bias)])
        cache = self.__dict__.get("_fc_made_inverse_pack")
        if cache is None or cache[0] != key:
            cache = self.__dict__["_fc_made_inverse_pack"] = (key, ops.pack_made_inverse(net, features, per_dim))
        n = inputs.shape[0]
        rows = inputs if n % ops.HIDDEN_ROWS == 0 else torch.nn.functional.pad(inputs, (0, 0, 0, -n % ops.HIDDEN_ROWS))
        outputs, logabsdet = ops.made_inverse(rows, cache[1], len(net.blocks), per_dim, kind, rq)
        return (outputs, logabsdet) if rows is inputs else (outputs[:n], logabsdet[:n])

    def _incremental_ok(self, inputs):
        """Column-at-a-time inverse applies to a MADE: its input degrees are 1..D and its output
        degrees tile(1..D) whatever the hidden masks are (made.py:14-49), so the parameters of dim d depend on
        outputs[:, :d] alone.  Inference only (the column writes are in place).  It pays when the final layer and
        the bijector dominate a pass, i.e. for the spline / sum-of-sigmoids forms; with 1-2 parameters per dim
        (shift, affine) a pass is the hidden stack either way and the GEMM library handles a 1-2 column product
        no better than a full-width one, so those keep the full passes; so do batches small enough to be bound by
        launches (a column pass has two more).  ``FC_AR_INCREMENTAL=force`` lifts both size rules, ``0`` disables."""
        net = self.autoregressive_net
        return (isinstance(net, made_module.MADE) and inputs.dim() == 2 and inputs.shape[1] > 1
                and ((net.final_layer.out_features >= 8 * inputs.shape[1] and inputs.shape[0] >= 8192)
                     or options.get("ar_incremental") == "force")
                and options.get("ar_incremental") != "off" and not self._needs_grad(inputs)
                and not net._forward_hooks and not net._forward_pre_hooks)

    def _inverse_incremental(self, inputs, context):
        """The reference's D passes (autoregressive.py:44-53) each recompute all D x P parameters and invert all D
        dims although pass d only fixes column d.  Here pass d runs the hidden stack on the columns found so far
        (the others still zero: they only meet zeroed weights), the P rows of the final layer that belong to dim d,
        and the element-wise inverse of that one column: final-layer and bijector work drop by a factor of D; the
        per-column log-determinants add up to the last pass's row sum."""
        net = self.autoregressive_net
        final = net.final_layer
        features = inputs.shape[1]
        per_dim = final.out_features // features
        columns = inputs.t().contiguous()           # row d = the d-th column, contiguous
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        with ops.deferred_errors():
            for d in range(features):
                hidden = self._hidden(outputs, context)
                params = self._final(hidden, slice(d * per_dim, (d + 1) * per_dim))
                column, lad = self._elementwise_inverse(columns[d].unsqueeze(1), params)
                outputs[:, d] = column[:, 0]
                logabsdet = lad if logabsdet is None else logabsdet + lad
        return outputs, logabsdet

    def _output_dim_multiplier(self):
        raise NotImplementedError()

    def _elementwise_forward(self, inputs, autoregressive_params):
        raise NotImplementedError()

    def _elementwise_inverse(self, inputs, autoregressive_params):
        raise NotImplementedError()


def _made(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
          random_mask, activation, dropout_probability, use_batch_norm):
    return made_module.MADE(
        features=features, hidden_features=hidden_features, context_features=context_features,
        num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
        use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
        dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)


class MaskedAffineAutoregressiveTransform(AutoregressiveTransform):
    """MAF layer: scale = softplus(u) + 1e-3, y = scale * x + shift (autoregressive.py:65-129)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False):
        self.features = features
        made = _made(self, features, hidden_features, context_features, num_blocks,
                     use_residual_blocks, random_mask, activation, dropout_probability, use_batch_norm)
        self._epsilon = 1e-3
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 2

    # ---- density direction in ONE kernel (round 3) ------------------------------------------------------------------
    # One MADE pass yields the parameters of all dims, so the forward map is a coupling layer that reads and transforms
    # every column: hidden stack, masked final Linear and the affine bijector run in fc_affine_coupling_resnet on
    # pre-masked weights (the README flow, examples/toy_2d.py: 3 launches per layer instead of ~7).
    def _one_kernel_ok(self, inputs, context):
        net = self.autoregressive_net
        if not (context is None and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and inputs.shape[0] >= ops.HIDDEN_ROWS and inputs.shape[1] <= 32
                and options.get("fused_hidden") and options.get("fused_final_layer")):
            return False

        def structure_ok():        # what only depends on how the conditioner is built
            return (isinstance(net, made_module.MADE) and inputs.shape[1] == net.initial_layer.in_features
                    and not hasattr(net, "context_layer") and len(net.blocks) <= 3 and net.hip_hidden_supported(None)
                    and ops.activation_code(net.activation) is not None
                    and ops.activation_code(net.activation)[0] == ops.ACT_RELU
                    and ops.affine_tail_fits(inputs.shape[1], len(net.blocks), inputs.shape[1]))

        return (ops.static_memo(self, "_fc_static_ok", (inputs.shape[1],) + ops.structure_key(net), structure_ok)
                and not ops.has_hooks(net) and not self._needs_grad(inputs))

    def _one_kernel(self, inputs, total=None):
        net = self.autoregressive_net
        features = inputs.shape[1]
        where = tuple(p.data_ptr() for p in ops.param_list(net))
        plan = getattr(self, "_tail_image", None)
        if plan is None or plan[0] != where:
            pack, packed = ops.device_pack_made_affine(net, features)
            cols = torch.arange(features, dtype=torch.int32, device=inputs.device)
            plan = self._tail_image = [where, pack, packed, cols]
        plan[1].refresh()
        n = inputs.shape[0]
        body = n - n % ops.HIDDEN_ROWS
        args = (plan[3], plan[3], plan[2], features, len(net.blocks), ops.AFFINE_MAF_SOFTPLUS)
        if body == n:
            return ops.affine_coupling_resnet(inputs, *args, logabsdet_accum=total)
        out_a, lad_a = ops.affine_coupling_resnet(inputs[:body], *args,
                                                  logabsdet_accum=None if total is None else total[:body])
        rest = inputs[body:].contiguous()          # the < 16 leftover rows: conditioner on PyTorch + the stand-alone kernel
        out_b, lad_b = self._elementwise_forward(rest, net(rest, None))
        outputs = torch.cat((out_a, out_b))
        if total is None:
            return outputs, torch.cat((lad_a, lad_b))
        total[body:] += lad_b
        return outputs, total

    def forward(self, inputs, context=None):
        if self._one_kernel_ok(inputs, context):
            return self._one_kernel(inputs)
        return super().forward(inputs, context)

    def _apply_accumulate(self, inputs, context, inverse, total):
        """CompositeTransform fast path: the kernel adds this layer's logabsdet onto ``total`` itself."""
        if not inverse and self._one_kernel_ok(inputs, context):
            return self._one_kernel(inputs, total)[0]
        outputs, logabsdet = self.inverse(inputs, context) if inverse else self.forward(inputs, context)
        total += logabsdet
        return outputs

    def _elementwise_forward(self, inputs, autoregressive_params):
        return ops.affine_coupling(inputs, autoregressive_params, None,
                                   activation=ops.AFFINE_MAF_SOFTPLUS, inverse=False)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return ops.affine_coupling(inputs, autoregressive_params, None,
                                   activation=ops.AFFINE_MAF_SOFTPLUS, inverse=True)

    def _device_loop_form(self):
        return (ops.MADE_AFFINE, 2, None)


class MaskedShiftAutoregressiveTransform(AutoregressiveTransform):
    """Shift-only AR layer.  As in the reference (autoregressive.py:164-196) ``forward`` adds
    ``2*tanh(shift)`` while ``inverse`` subtracts the raw ``shift``; logabsdet is 0 both ways."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False):
        self.features = features
        made = _made(self, features, hidden_features, context_features, num_blocks,
                     use_residual_blocks, random_mask, activation, dropout_probability, use_batch_norm)
        self._epsilon = 1e-3
        self.shift_scale = 1.
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 1

    def _params(self, autoregressive_params):
        if self.shift_scale != 1.:
            return autoregressive_params * self.shift_scale
        return autoregressive_params

    def _elementwise_forward(self, inputs, autoregressive_params):
        return ops.affine_coupling(inputs, self._params(autoregressive_params), None,
                                   activation=ops.AFFINE_SHIFT_TANH2, inverse=False)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return ops.affine_coupling(inputs, self._params(autoregressive_params), None,
                                   activation=ops.AFFINE_ADDITIVE, inverse=True)


class MaskedPiecewiseRationalQuadraticAutoregressiveTransform(AutoregressiveTransform):
    """RQ-spline AR layer (autoregressive.py:529-621): identity-init softplus beta, box
    [-1.2, 1.2]^2 when ``tails`` is None, width/height scaling only if the net exposes
    ``hidden_features`` (MADE does not)."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, tails=None,
                 tail_bound=1.0, num_blocks=2, use_residual_blocks=True, random_mask=False,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        made = _made(self, features, hidden_features, context_features, num_blocks,
                     use_residual_blocks, random_mask, activation, dropout_probability, use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        elif self.tails is None:
            return self.num_bins * 3 + 1
        else:
            raise ValueError

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        if self.tails not in (None, "linear"):
            raise ValueError
        divisor = 1.0
        if hasattr(self.autoregressive_net, "hidden_features"):
            divisor = float(np.sqrt(self.autoregressive_net.hidden_features))
        return ops.rq_spline_autograd(
            inputs, autoregressive_params, None, num_bins=self.num_bins, tails=self.tails,
            tail_bound=self.tail_bound, left=-1.2, right=1.2, bottom=-1.2, top=1.2,
            min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative, enable_identity_init=True, wh_divisor=divisor,
            inverse=inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)

    def _device_loop_form(self):
        if self.tails not in (None, "linear") or not 1 <= self.num_bins <= 16:
            return None
        divisor = 1.0
        if hasattr(self.autoregressive_net, "hidden_features"):
            divisor = float(np.sqrt(self.autoregressive_net.hidden_features))
        return (ops.MADE_RQ, self._output_dim_multiplier(),
                dict(num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound, left=-1.2, right=1.2,
                     bottom=-1.2, top=1.2, min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                     min_derivative=self.min_derivative, enable_identity_init=True, wh_divisor=divisor))

    # ---- density direction on the fused kernels ------------------------------------------------------------------
    # One MADE pass yields the parameters of all D dims, so the forward is a coupling layer that transforms every
    # column: hidden stack in fc_resnet_hidden (pre-masked weights), masked final Linear + spline in
    # fc_rq_spline_fused_linear (K = 8, linear tails) or fc_rq_spline_fused_general (the reference's defaults --
    # num_bins = 10, tails = None on the [-1.2, 1.2] box, autoregressive.py:536,595 -- and every other K = 4..16) -- the
    # [N, D (3K-/+1)] parameter tensor never reaches HBM.  Shapes: D <= 32, hidden <= 64, inference only.
    def _fused_forward_mode(self, inputs, context):
        """None, "k8" or "general"."""
        net = self.autoregressive_net
        if not (inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and options.get("fused_final_layer") and options.get("fused_hidden")
                and isinstance(net, made_module.MADE) and not hasattr(net, "hidden_features")
                and net.final_layer.in_features <= 64 and net.hip_hidden_supported(context)
                and inputs.shape[0] >= ops.FUSED_ROWS and not ops.has_hooks(net) and not self._needs_grad(inputs)):
            return None
        n, d = inputs.shape
        if ops.fused_linear_supported(n, d, d, net.final_layer.in_features, self.num_bins, self.tails):
            return "k8"
        if ops.fused_general_supported(n, d, d, 64, self.num_bins, self.tails):
            return "general"
        return None

    def _fused_forward_ok(self, inputs, context):
        return self._fused_forward_mode(inputs, context) is not None

    def _packed_final_layer(self, device, mode="k8"):
        lin = self.autoregressive_net.final_layer
        key = ops.cache_key(lin.weight, lin.bias)
        if getattr(self, "_packed", None) is None or self._packed[0] != (key, mode):
            masked = (lin.weight * lin.mask).detach()
            if mode == "k8":
                packed = ops.pack_final_layer(masked, lin.bias, self.num_bins)
            else:
                packed = ops.pack_final_layer_general(masked, lin.bias, self.num_bins, self.tails, 64)
            cols = torch.arange(lin.out_features // self._output_dim_multiplier(), dtype=torch.int32, device=device)
            self._packed = ((key, mode), masked) + tuple(packed) + (cols,)
        return self._packed[1:]

    def forward(self, inputs, context=None):
        mode = self._fused_forward_mode(inputs, context)
        if mode is None:
            return super().forward(inputs, context)
        hidden = self._hidden(inputs, context)
        kw = dict(num_bins=self.num_bins, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                  min_bin_height=self.min_bin_height, min_derivative=self.min_derivative, wh_divisor=1.0,
                  enable_identity_init=True)
        n = inputs.shape[0]
        body = n - n % ops.FUSED_ROWS
        if mode == "k8":
            _, w_pad, b_pad, cols = self._packed_final_layer(inputs.device)
            outputs, logabsdet = ops.rq_spline_fused_linear(inputs[:body], hidden[:body], w_pad, b_pad, cols, **kw)
        else:
            _, w_frag, w_un, b_pad, cols = self._packed_final_layer(inputs.device, "general")
            outputs, logabsdet = ops.rq_spline_fused_general(inputs[:body], hidden[:body], w_frag, w_un, b_pad, cols,
                                                             tails=self.tails, left=-1.2, right=1.2, bottom=-1.2, top=1.2, **kw)
        if body < n:   # the < 32 leftover rows: masked final Linear + the stand-alone kernel
            out_b, lad_b = self._elementwise_forward(inputs[body:], self._final(hidden[body:]))
            outputs, logabsdet = torch.cat((outputs, out_b)), torch.cat((logabsdet, lad_b))
        return outputs, logabsdet


class MaskedSumOfSigmoidsTransform(AutoregressiveTransform):
    """Sum-of-sigmoids AR layer (autoregressive.py:266-318): MADE emits 3S+1 raw values per feature;
    forward returns z - 0.5, the inverse first adds 0.5 back."""

    def __init__(self, features, hidden_features, n_sigmoids=30, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False):
        self.features = features
        self.n_sigmoids = n_sigmoids
        made = _made(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
                     random_mask, activation, dropout_probability, use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 3 * self.n_sigmoids + 1

    def _elementwise_forward(self, inputs, autoregressive_params):
        return ops.sum_of_sigmoids_autograd(inputs, autoregressive_params, self.n_sigmoids, inverse=False, offset=0.5)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return ops.sum_of_sigmoids_autograd(inputs, autoregressive_params, self.n_sigmoids, inverse=True, offset=0.5)


def _ar_divisor(net):
    return float(np.sqrt(net.hidden_features)) if hasattr(net, "hidden_features") else 1.0


class MaskedPiecewiseLinearAutoregressiveTransform(AutoregressiveTransform):
    """Piecewise-linear AR layer on [0, 1] (autoregressive.py:321-372)."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        made = _made(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
                     random_mask, activation, dropout_probability, use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return self.num_bins

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        return ops.piecewise_spline_autograd(inputs, autoregressive_params, None, kind=ops.SPLINE_LINEAR,
                                    num_bins=self.num_bins, inverse=inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)


class MaskedPiecewiseQuadraticAutoregressiveTransform(AutoregressiveTransform):
    """Piecewise-quadratic AR layer (autoregressive.py:375-460); only the widths are scaled, and only
    when the net exposes ``hidden_features``."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False, tails=None, tail_bound=1.0, min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT, min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        self.features = features
        made = _made(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
                     random_mask, activation, dropout_probability, use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 2 - 1
        return self.num_bins * 2 + 1

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        if self.tails not in (None, "linear"):
            raise ValueError
        return ops.piecewise_spline_autograd(inputs, autoregressive_params, None, kind=ops.SPLINE_QUADRATIC,
                                    num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                                    min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                    width_divisor=_ar_divisor(self.autoregressive_net), inverse=inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)


class MaskedPiecewiseCubicAutoregressiveTransform(AutoregressiveTransform):
    """Piecewise-cubic AR layer on [0, 1] (autoregressive.py:463-526)."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.0,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        made = _made(self, features, hidden_features, context_features, num_blocks, use_residual_blocks,
                     random_mask, activation, dropout_probability, use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _elementwise(self, inputs, autoregressive_params, inverse=False):
        div = _ar_divisor(self.autoregressive_net)
        return ops.piecewise_spline_autograd(inputs, autoregressive_params, None, kind=ops.SPLINE_CUBIC,
                                    num_bins=self.num_bins, width_divisor=div, height_divisor=div,
                                    inverse=inverse)

    def _elementwise_forward(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params)

    def _elementwise_inverse(self, inputs, autoregressive_params):
        return self._elementwise(inputs, autoregressive_params, inverse=True)
