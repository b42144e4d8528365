"""Coupling layers (API of flowcon/transforms/coupling.py:20-582).

The conditioner (``transform_net``) is whatever ``nn.Module`` the user's
``transform_net_create_fn(in_features, out_features)`` returns and runs on PyTorch-ROCm.  The
element-wise bijector, the split / merge of the two halves and the per-sample logabsdet
reduction are ONE HIP kernel launch per layer: the kernel reads the full ``[N, D]`` input and
the conditioner's ``[N, d_t * multiplier]`` output once, writes the full output once
(identity columns copied through LDS) and the ``[N]`` logabsdet once.
"""
import warnings

import numpy as np
import torch

from flowconductor_amd import ops, options
from flowconductor_amd.transforms.base import Transform


def _is_plain_resnet(net):
    """This package's ResidualNet, or a subclass that keeps its ``forward`` / ``hidden`` (a subclass that overrides
    either computes something the kernels do not know about and takes the generic path)."""
    from flowconductor_amd.nn.nets.resnet import ResidualNet

    return (isinstance(net, ResidualNet) and type(net).forward is ResidualNet.forward
            and type(net).hidden is ResidualNet.hidden)


def _rows_from_nchw(t):
    """[B, C, H, W] -> ([B*H*W, C] contiguous, (B, C, H, W))."""
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c), (b, c, h, w)


def _nchw_from_rows(rows, shape):
    b, c, h, w = shape
    return rows.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()


class CouplingTransform(Transform):
    """Base class for coupling layers: 2-D ``[N, D]`` inputs, or 4-D ``[N, C, H, W]`` split on C.

    mask[i] > 0: feature i is transformed; mask[i] <= 0: feature i passes unchanged and
    feeds the conditioner (reference coupling.py:25-63).
    """

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError("Mask must be a 1-dim tensor.")
        if mask.numel() <= 0:
            raise ValueError("Mask can't be empty.")

        super().__init__()
        self.features = len(mask)
        features_vector = torch.arange(self.features)
        self.register_buffer("identity_features", features_vector.masked_select(mask <= 0))
        self.register_buffer("transform_features", features_vector.masked_select(mask > 0))
        assert self.num_identity_features + self.num_transform_features == self.features

        self.transform_net = transform_net_create_fn(
            self.num_identity_features,
            self.num_transform_features * self._transform_dim_multiplier(),
        )
        if unconditional_transform is None:
            self.unconditional_transform = None
        else:
            self.unconditional_transform = unconditional_transform(features=self.num_identity_features)
        self._cols_cache = None

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    def _cols(self, device):
        """int32 copy of ``transform_features`` on ``device`` (what the kernel indexes with)."""
        c = self._cols_cache
        if c is None or c.device != device or c.numel() != self.num_transform_features:
            c = self.transform_features.to(device=device, dtype=torch.int32).contiguous()
            self._cols_cache = c
        return c

    def _id_cols(self, device):
        """int32 copy of ``identity_features`` on ``device``."""
        c = getattr(self, "_id_cols_cache", None)
        if c is None or c.device != device or c.numel() != self.num_identity_features:
            c = self.identity_features.to(device=device, dtype=torch.int32).contiguous()
            self._id_cols_cache = c
        return c

    def _check(self, inputs):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Inputs must be a 2D or a 4D tensor.")
        if inputs.shape[1] != self.features:
            raise ValueError("Expected features = {}, got {}.".format(self.features, inputs.shape[1]))

    def _run(self, inputs, context, inverse):
        self._check(inputs)
        # the identity half as a tensor of its own (a gather) only where something consumes it: the
        # hidden-layer kernel reads the identity columns straight from the full rows
        identity_split = logabsdet_identity = None
        if self.unconditional_transform is not None:
            identity_split = inputs[:, self.identity_features, ...]
            if inverse:
                identity_split, logabsdet_identity = self.unconditional_transform.inverse(identity_split, context)

        transform_params = self._conditioner(inputs, identity_split, context)

        if inputs.dim() == 4:
            rows, shape = _rows_from_nchw(inputs)
            prows = self._param_rows_nchw(transform_params, shape)
            out_rows, lad_rows = self._coupling_kernel(rows, prows, inverse)
            outputs = _nchw_from_rows(out_rows, shape)
            logabsdet = lad_rows.reshape(shape[0], -1).sum(dim=1)
        else:
            outputs, logabsdet = self._coupling_kernel(inputs, transform_params, inverse)

        if self.unconditional_transform is not None:
            if not inverse:
                identity_split, logabsdet_identity = self.unconditional_transform(identity_split, context)
            outputs[:, self.identity_features, ...] = identity_split
            logabsdet = logabsdet + logabsdet_identity
        return outputs, logabsdet

    def _conditioner(self, inputs, identity_split, context):
        """``transform_net(identity_split, context)``.  For this package's ResidualNet with the shape
        ``fc_resnet_hidden`` covers (hidden 64, <= 4 blocks, ReLU; with a [N, <= 32] context <= 3 blocks) the hidden
        layers run in that one kernel,
        straight from the full input rows, and only the final Linear stays a library GEMM -- for every coupling
        bijector (affine, additive, all splines), inference only."""
        net = self.transform_net
        n = inputs.shape[0]
        if (_is_plain_resnet(net) and not ops.has_hooks(net) and self.unconditional_transform is None
                and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32 and n >= ops.HIDDEN_ROWS
                and options.get("fused_hidden")
                and net.hip_hidden_supported(inputs.shape[1], context)
                and not (torch.is_grad_enabled()
                         and (inputs.requires_grad or any(p.requires_grad for p in net.parameters())))):
            body = n - n % ops.HIDDEN_ROWS
            hidden = net.hidden_hip(inputs[:body], self._id_cols(inputs.device),
                                    None if context is None else context[:body])
            if body < n:
                hidden = torch.cat((hidden, net.hidden_padded(inputs[body:, self.identity_features],
                                                              None if context is None else context[body:])))
            return net.final_from_padded(hidden)
        if identity_split is None:
            identity_split = inputs[:, self.identity_features, ...]
        return net(identity_split, context)

    def forward(self, inputs, context=None):
        return self._run(inputs, context, inverse=False)

    def inverse(self, inputs, context=None):
        return self._run(inputs, context, inverse=True)

    def _transform_dim_multiplier(self):
        """Number of features to output for each transform dimension."""
        raise NotImplementedError()

    def _coupling_kernel(self, inputs, transform_params, inverse):
        """[N, D] inputs + [N, d_t * multiplier] params -> ([N, D] outputs, [N] logabsdet)."""
        raise NotImplementedError()

    def _param_rows_nchw(self, transform_params, shape):
        """Conditioner output [B, d_t*mult, H, W] -> per-pixel rows in the 2-D layout."""
        raise NotImplementedError()


class AffineCouplingTransform(CouplingTransform):
    """RealNVP affine coupling: y = x * scale + shift on the transformed half.

    ``scale_activation`` follows the reference (coupling.py:212-252): the two predefined
    class attributes select a fused activation inside the kernel; any other callable is
    evaluated with torch and handed to the kernel as a ready scale.
    """

    DEFAULT_SCALE_ACTIVATION = lambda x: torch.sigmoid(x + 2) + 1e-3  # noqa: E731
    GENERAL_SCALE_ACTIVATION = lambda x: (torch.nn.functional.softplus(x) + 1e-3).clamp(0, 3)  # noqa: E731

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None,
                 scale_activation=DEFAULT_SCALE_ACTIVATION):
        self.scale_activation = scale_activation
        super().__init__(mask, transform_net_create_fn, unconditional_transform)

    def _transform_dim_multiplier(self):
        return 2

    def _activation_code(self):
        if self.scale_activation is AffineCouplingTransform.DEFAULT_SCALE_ACTIVATION:
            return ops.AFFINE_SIGMOID_PLUS2
        if self.scale_activation is AffineCouplingTransform.GENERAL_SCALE_ACTIVATION:
            return ops.AFFINE_SOFTPLUS_CLAMP3
        return ops.AFFINE_SCALE_GIVEN

    def _coupling_kernel(self, inputs, transform_params, inverse, total=None):
        code = self._activation_code()
        if code == ops.AFFINE_SCALE_GIVEN:
            d_t = self.num_transform_features
            scale = self.scale_activation(transform_params[:, d_t:])
            transform_params = torch.cat((transform_params[:, :d_t], scale), dim=1)
        return ops.affine_coupling(inputs, transform_params, self._cols(inputs.device),
                                   activation=code, inverse=inverse, logabsdet_accum=total)

    def _one_kernel_ok(self, inputs, context):
        """The whole layer in ``fc_affine_coupling_resnet``: plain ResidualNet(hidden <= 64, <= 3 ReLU blocks, no context,
        no batch norm / active dropout, no hooks), <= 32 transformed dims, a scale activation the kernel knows, 2-D float32
        inputs on the device, inference only."""
        net = self.transform_net
        if not (context is None and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and inputs.shape[0] >= ops.HIDDEN_ROWS and options.get("fused_hidden") and options.get("fused_final_layer")):
            return False

        def structure_ok():        # what only depends on how the layer and its conditioner are built (and the input width)
            return (self.unconditional_transform is None and self.num_transform_features <= 32
                    and ops.affine_tail_activation(self._activation_code()) and _is_plain_resnet(net)
                    and ops.affine_tail_fits(net.initial_layer.in_features, len(net.blocks), inputs.shape[1])
                    and net.hip_hidden_supported(inputs.shape[1], None)
                    and all(ops.activation_code(b.activation)[0] == ops.ACT_RELU for b in net.blocks))

        return (ops.static_memo(self, "_fc_static_ok", (inputs.shape[1], id(self.unconditional_transform)) + ops.structure_key(net),
                                structure_ok)
                and not ops.has_hooks(net)
                and not (torch.is_grad_enabled()
                         and (inputs.requires_grad or any(p.requires_grad for p in net.parameters()))))

    def _one_kernel(self, inputs, inverse, total):
        net = self.transform_net
        where = net._storage_key()
        plan = getattr(self, "_tail_image", None)
        if plan is None or plan[0] != where:
            pack, packed = ops.device_pack_affine_coupling(net, self.num_transform_features,
                                                           self._activation_code() == ops.AFFINE_ADDITIVE)
            plan = self._tail_image = [where, pack, packed]
        plan[1].refresh()
        n = inputs.shape[0]
        body = n - n % ops.HIDDEN_ROWS
        dev = inputs.device
        args = (self._id_cols(dev), self._cols(dev), plan[2], net.initial_layer.in_features, len(net.blocks),
                self._activation_code())
        if body == n:
            return ops.affine_coupling_resnet(inputs, *args, inverse=inverse, logabsdet_accum=total)
        # the < 16 leftover rows: conditioner on PyTorch + the stand-alone kernel
        out_a, lad_a = ops.affine_coupling_resnet(inputs[:body], *args, inverse=inverse,
                                                  logabsdet_accum=None if total is None else total[:body])
        rest = inputs[body:].contiguous()
        params = net(rest[:, self.identity_features], None)
        out_b, lad_b = self._coupling_kernel(rest, params, inverse)
        outputs = torch.cat((out_a, out_b))
        if total is None:
            return outputs, torch.cat((lad_a, lad_b))
        total[body:] += lad_b
        return outputs, total

    def _run(self, inputs, context, inverse):
        if self._one_kernel_ok(inputs, context):
            self._check(inputs)
            return self._one_kernel(inputs, inverse, None)
        return super()._run(inputs, context, inverse)

    def _apply_accumulate(self, inputs, context, inverse, total):
        """CompositeTransform fast path: the kernel adds this layer's logabsdet onto ``total`` itself."""
        if inputs.dim() != 2 or self.unconditional_transform is not None or not inputs.is_cuda:
            outputs, logabsdet = self._run(inputs, context, inverse)
            total += logabsdet
            return outputs
        self._check(inputs)
        if self._one_kernel_ok(inputs, context):
            outputs, _ = self._one_kernel(inputs, inverse, total)
            return outputs
        transform_params = self._conditioner(inputs, None, context)
        outputs, _ = self._coupling_kernel(inputs, transform_params, inverse, total=total)
        return outputs

    def _param_rows_nchw(self, transform_params, shape):
        return _rows_from_nchw(transform_params)[0]


class AdditiveCouplingTransform(AffineCouplingTransform):
    """NICE additive coupling: y = x + shift, logabsdet = 0 (coupling.py:255-269)."""

    def _transform_dim_multiplier(self):
        return 1

    def _activation_code(self):
        return ops.AFFINE_ADDITIVE

    def _coupling_kernel(self, inputs, transform_params, inverse, total=None):
        # logabsdet == 0: nothing to add onto a running total
        return ops.affine_coupling(inputs, transform_params, self._cols(inputs.device),
                                   activation=ops.AFFINE_ADDITIVE, inverse=inverse)


class PiecewiseCouplingTransform(CouplingTransform):
    """Coupling layers whose bijector is a K-bin spline with per-(sample, dim) parameters."""

    def _param_rows_nchw(self, transform_params, shape):
        b, _, h, w = shape
        c = self.num_transform_features
        # [B, c*mult, H, W] -> [B, c, mult, H, W] -> [B, H, W, c, mult] -> rows
        p = transform_params.reshape(b, c, -1, h, w).permute(0, 3, 4, 1, 2)
        return p.reshape(b * h * w, -1)

    def _wh_divisor(self):
        raise NotImplementedError()


def _softmax_divisor(net, warn):
    """sqrt(hidden width) the reference divides unnormalised widths/heights by (coupling.py:554-563)."""
    if hasattr(net, "hidden_features"):
        return float(np.sqrt(net.hidden_features))
    if hasattr(net, "hidden_channels"):
        return float(np.sqrt(net.hidden_channels))
    if warn:
        warnings.warn("Inputs to the softmax are not scaled down: initialization might be bad.")
    return 1.0


class PiecewiseRationalQuadraticCouplingTransform(PiecewiseCouplingTransform):
    """Neural-spline-flow coupling layer (coupling.py:502-582); the north-star kernel."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None,
                 min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH, min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=ops.DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound

        if apply_unconditional_transform:
            from flowconductor_amd.transforms.nonlinearities import PiecewiseRationalQuadraticCDF

            def unconditional_transform(features):
                return PiecewiseRationalQuadraticCDF(
                    shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                    tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                    min_bin_height=min_bin_height, min_derivative=min_derivative)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        return self.num_bins * 3 + 1

    def _coupling_kernel(self, inputs, transform_params, inverse):
        # (records an autograd node when gradients are required: ops._RQSplineFunction)
        return ops.rq_spline_autograd(
            inputs, transform_params, self._cols(inputs.device), num_bins=self.num_bins,
            tails=self.tails, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
            min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
            wh_divisor=_softmax_divisor(self.transform_net, warn=True), inverse=inverse)

    # ---- fused final conditioner layer -------------------------------------------
    # When the conditioner is this package's ResidualNet its last nn.Linear is evaluated INSIDE the spline kernel on the
    # matrix cores (split-f16 products, f32-GEMM accuracy), so the [N, d_t (3K -/+ 1)] parameter tensor never touches
    # HBM.  Two kernels: "k8" -- the hand-scheduled north-star shape (K = 8, linear tails, hidden <= 64: weights
    # resident in registers, fc_rq_spline_fused_linear) -- and "general" -- K = 4..16, tails linear or None, hidden
    # <= 256, weights streamed from L2 in packed fragment order (fc_rq_spline_fused_general; the reference's default
    # layer, num_bins = 10, lands here).  Both take up to 32 transformed dims per launch: wider layers (D <= 128)
    # chain one launch per 32 dims, each passing the other columns through.  The hidden layers run in fc_resnet_hidden
    # when that kernel covers them, else on PyTorch-ROCm.  Inference only; any other conditioner / shape takes the
    # generic path; options.override(fused_final_layer=False) disables it.

    def _fused_mode(self, inputs):
        """None, "k8" or "general"."""
        net = self.transform_net
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in net.parameters())):
            return None   # training: conditioner on PyTorch autograd + the spline's own backward kernel
        if not (options.get("fused_final_layer") and _is_plain_resnet(net) and not ops.has_hooks(net)
                and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32):
            return None
        n, d = inputs.shape
        d_t = min(self.num_transform_features, ops.FUSED_DT)
        if ops.fused_linear_supported(n, d, d_t, net.hidden_features, self.num_bins, self.tails):
            return "k8"
        if ops.fused_general_supported(n, d, d_t, net.hidden_features, self.num_bins, self.tails):
            return "general"
        return None

    def _fused_ok(self, inputs):
        return self._fused_mode(inputs) is not None

    def _fused_chunks(self, device, mode):
        """Per group of <= 32 transformed dims: the final Linear's rows of those dims in the kernel's layout + their
        column indices -- (w_pad, bias_pad, cols) for "k8", (w_frag, w_unscale, bias_pad, cols) for "general"."""
        lin = self.transform_net.final_layer
        key = ops.cache_key(lin.weight, lin.bias)
        cache = getattr(self, "_packed", None)
        if cache is None or cache[0] != key:
            cache = self._packed = (key, {})
        if mode not in cache[1]:
            per_dim = self._transform_dim_multiplier()
            cols = self._cols(device)
            hidden_pad = ops.general_hidden_width(lin.in_features)
            chunks = []
            for lo in range(0, self.num_transform_features, ops.FUSED_DT):
                hi = min(lo + ops.FUSED_DT, self.num_transform_features)
                rows = slice(lo * per_dim, hi * per_dim)
                if mode == "k8":
                    packed = ops.pack_final_layer(lin.weight[rows], lin.bias[rows], self.num_bins)
                elif mode == "general":
                    packed = ops.pack_final_layer_general(lin.weight[rows], lin.bias[rows], self.num_bins, self.tails,
                                                          hidden_pad)
                else:   # "transposed": W^T fragments of the backward product gh = W^T G
                    packed = (ops.pack_final_layer_transposed(lin.weight[rows], self.num_bins, self.tails), rows)
                chunks.append(packed + (cols[lo:hi].contiguous(),))
            cache[1][mode] = chunks
        return cache[1][mode]

    def _train_chunks(self, device):
        """Training: forward + W^T fragments of the final Linear for every group of <= 32 dims, re-packed on the device
        in one launch whenever the weights changed -- [(w_frag, w_unscale, bias_pad, wt_frag, cols, row slice)]."""
        net = self.transform_net
        lin = net.final_layer
        hidden_pack = None
        if getattr(net, "hip_hidden_backward_supported", None) is not None and net.hip_hidden_backward_supported():
            hidden_pack = net.hidden_backward_plan()[1]        # rebuilt by the net when ITS storages moved
        where = (lin.weight.data_ptr(), lin.bias.data_ptr(), device)
        plan = getattr(self, "_train_pack", None)
        if plan is None or plan[0] is not lin.weight or plan[1] != where or plan[4] is not hidden_pack:
            per_dim = self._transform_dim_multiplier()
            cols = self._cols(device)
            spec = []
            for lo in range(0, self.num_transform_features, ops.FUSED_DT):
                hi = min(lo + ops.FUSED_DT, self.num_transform_features)
                spec.append((slice(lo * per_dim, hi * per_dim), cols[lo:hi].contiguous()))
            pack, chunks = ops.device_pack_final_layer(lin.weight.detach(), lin.bias.detach(), self.num_bins, self.tails,
                                                       spec)
            if hidden_pack is not None:
                pack.merge(hidden_pack)      # the hidden stack's images ride in the same launch
            plan = self._train_pack = [lin.weight, where, pack, chunks, hidden_pack]
        plan[2].refresh()
        return plan[3]

    def _apply_accumulate(self, inputs, context, inverse, total):
        """CompositeTransform fast path: the fused kernel adds this layer's logabsdet onto ``total`` itself."""
        if (self._fused_mode(inputs) is None or self.unconditional_transform is not None
                or self._fused_training_ok(inputs, context, inverse)):
            outputs, logabsdet = self._run(inputs, context, inverse)
            total += logabsdet
            return outputs
        outputs, _ = self._run(inputs, context, inverse, total=total)
        return outputs

    def _fused_kw(self, net):
        return dict(num_bins=self.num_bins, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                    min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                    wh_divisor=_softmax_divisor(net, warn=False))

    def _hidden_for_fused(self, inputs, identity_split, context, width):
        """[N, width] hidden activation of the conditioner (zero columns beyond ``hidden_features``): the hidden-layer
        kernel straight from the full input rows where it covers the net, PyTorch-ROCm otherwise / for leftover rows."""
        net = self.transform_net
        n = inputs.shape[0]

        def on_torch(rows_identity, ctx):
            h = net.hidden(rows_identity, ctx)
            return h if h.shape[1] == width else torch.nn.functional.pad(h, (0, width - h.shape[1]))

        body16 = n - n % ops.HIDDEN_ROWS
        if (identity_split is None and width == 64 and body16 > 0 and options.get("fused_hidden")
                and net.hip_hidden_supported(inputs.shape[1], context)):
            hidden = net.hidden_hip(inputs[:body16], self._id_cols(inputs.device),
                                    None if context is None else context[:body16])
            if body16 < n:
                hidden = torch.cat((hidden, on_torch(inputs[body16:, self.identity_features],
                                                     None if context is None else context[body16:])))
            return hidden
        body64 = n - n % ops.WIDE_ROWS
        if (identity_split is None and width > 64 and body64 > 0 and options.get("fused_hidden")
                and net.hip_hidden_wide_supported(inputs.shape[1], context)):
            hidden = net.hidden_hip_wide(inputs[:body64], self._id_cols(inputs.device))
            if body64 < n:
                hidden = torch.cat((hidden, on_torch(inputs[body64:, self.identity_features], None)))
            return hidden
        if identity_split is None:
            identity_split = inputs[:, self.identity_features]
        return on_torch(identity_split, context)

    def _fused_training_ok(self, inputs, context, inverse):
        """Training through the HIP kernels (SURVEY 8f #3 with #4): conditioner forward in fc_resnet_hidden + the fused
        final-layer kernel, backward in fc_rq_fused_linear_backward -- no [N, d_t P] tensor in either direction."""
        net = self.transform_net
        return (not inverse and context is None and self.unconditional_transform is None
                and options.get("fused_training") and options.get("fused_final_layer") and options.get("fused_hidden")
                and torch.is_grad_enabled() and _is_plain_resnet(net) and not ops.has_hooks(net)
                and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and (inputs.requires_grad or any(p.requires_grad for p in net.parameters()))
                and net.hip_hidden_supported(inputs.shape[1], None)
                and ops.fused_backward_supported(inputs.shape[0], inputs.shape[1],
                                                 min(self.num_transform_features, ops.FUSED_DT), net.hidden_features,
                                                 self.num_bins, self.tails))

    def _run(self, inputs, context, inverse, total=None):
        if self._fused_training_ok(inputs, context, inverse):
            self._check(inputs)
            outputs, logabsdet = _FusedRQCouplingFunction.apply(self, inputs, *self.transform_net.parameters())
            if total is not None:
                total += logabsdet
                logabsdet = total
            return outputs, logabsdet
        mode = self._fused_mode(inputs)
        if mode is None:
            return super()._run(inputs, context, inverse)
        self._check(inputs)
        net = self.transform_net
        n = inputs.shape[0]
        identity_split = logabsdet_identity = None
        if self.unconditional_transform is not None:
            identity_split = inputs[:, self.identity_features]
            if inverse:
                identity_split, logabsdet_identity = self.unconditional_transform.inverse(identity_split, context)
        width = 64 if mode == "k8" else ops.general_hidden_width(net.hidden_features)
        hidden = self._hidden_for_fused(inputs, identity_split, context, width)
        chunks = self._fused_chunks(inputs.device, mode)
        kw = dict(num_bins=self.num_bins, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                  min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                  wh_divisor=_softmax_divisor(net, warn=False), inverse=inverse)

        def fused(rows, h, accum):
            # the groups only depend on the identity columns: any order, each launch passes the rest through
            lad = accum
            for chunk in chunks:
                if mode == "k8":
                    w_pad, bias_pad, cols = chunk
                    rows, lad = ops.rq_spline_fused_linear(rows, h, w_pad, bias_pad, cols, logabsdet_accum=lad, **kw)
                else:
                    w_frag, w_un, bias_pad, cols = chunk
                    rows, lad = ops.rq_spline_fused_general(rows, h, w_frag, w_un, bias_pad, cols, tails=self.tails,
                                                            logabsdet_accum=lad, **kw)
            return rows, lad

        body = n - n % ops.FUSED_ROWS
        if body == n:
            outputs, logabsdet = fused(inputs, hidden, total)
        else:
            # the < 32 leftover rows go through the final Linear + the stand-alone kernel
            out_a, lad_a = fused(inputs[:body], hidden[:body], None if total is None else total[:body])
            tail_params = torch.nn.functional.linear(hidden[body:, :net.final_layer.in_features],
                                                     net.final_layer.weight, net.final_layer.bias)
            out_b, lad_b = self._coupling_kernel(inputs[body:].contiguous(), tail_params, inverse)
            outputs = torch.cat((out_a, out_b))
            if total is None:
                logabsdet = torch.cat((lad_a, lad_b))
            else:
                total[body:] += lad_b
                logabsdet = total
        if self.unconditional_transform is not None:
            if not inverse:
                identity_split, logabsdet_identity = self.unconditional_transform(identity_split, context)
            outputs[:, self.identity_features] = identity_split
            logabsdet = logabsdet + logabsdet_identity
        return outputs, logabsdet


class _FusedRQCouplingFunction(torch.autograd.Function):
    """One RQ coupling layer with a ResidualNet(hidden <= 64) conditioner as ONE autograd node.

    forward: hidden stack in ``fc_resnet_hidden``, final Linear + spline in ``fc_rq_spline_fused_general``; saves the
    layer input and the [N, 64] hidden activation (512 B per sample) instead of the [N, d_t P] parameter tensor.
    backward: ``fc_rq_fused_linear_backward`` (gx, gh, gW, gb of the final layer; parameters recomputed on the matrix
    cores), then the hidden stack's gradients -- its forward is recomputed from the saved input columns."""

    @staticmethod
    def forward(ctx, layer, inputs, *net_params):
        net = layer.transform_net
        n = inputs.shape[0]
        with torch.no_grad():
            x = inputs.detach()
            pad = (-n) % ops.HIDDEN_BWD_ROWS
            if pad:       # whole 128-row rounds: zero rows (inside every spline's interval) that get zero upstream gradients
                x = torch.cat((x, x.new_zeros(pad, x.shape[1])))
            hidden = layer._hidden_for_fused(x, None, None, 64)
            kw = layer._fused_kw(net)
            rows, lad = x, None
            # (raw weights: only a conditioner that IS 64 wide -- a narrower one is zero-padded by the packed fragments)
            k8 = net.hidden_features == ops.FUSED_HIDDEN and ops.fused_linear_supported(
                x.shape[0], x.shape[1], min(layer.num_transform_features, ops.FUSED_DT), net.hidden_features,
                layer.num_bins, layer.tails)
            lin = net.final_layer
            for w_frag, w_un, bias_pad, _, cols, rows_slice in layer._train_chunks(x.device):
                if k8:    # the hand-scheduled north-star kernel takes the f32 weights as they are
                    rows, lad = ops.rq_spline_fused_linear(rows, hidden, lin.weight[rows_slice], lin.bias[rows_slice], cols,
                                                           logabsdet_accum=lad, inverse=False, **kw)
                else:
                    rows, lad = ops.rq_spline_fused_general(rows, hidden, w_frag, w_un, bias_pad, cols,
                                                            tails=layer.tails, logabsdet_accum=lad, inverse=False, **kw)
        ctx.layer, ctx.n = layer, n
        # The backward re-packs and reads the LIVE weights (they are not copied): saving the parameters makes autograd's
        # version check refuse a backward after an optimizer step / in-place edit between forward and backward, exactly
        # as it would for the plain torch graph ("modified by an inplace operation").
        ctx.save_for_backward(x, hidden, *net_params)
        ctx.param_ids = [id(p) for p in net_params]
        return rows[:n], lad[:n]

    @staticmethod
    @torch.autograd.function.once_differentiable      # no double backward: create_graph=True raises instead of going silent
    def backward(ctx, grad_outputs, grad_logabsdet):
        layer, n = ctx.layer, ctx.n
        net = layer.transform_net
        x, hidden = ctx.saved_tensors[:2]
        rows_total = x.shape[0]
        gy = torch.zeros_like(x) if grad_outputs is None else grad_outputs
        gl = grad_logabsdet
        if rows_total != n:
            gy = torch.cat((gy, gy.new_zeros(rows_total - n, gy.shape[1])))
            gl = None if gl is None else torch.cat((gl, gl.new_zeros(rows_total - n)))
        kw = layer._fused_kw(net)
        lin = net.final_layer
        chunks = layer._train_chunks(x.device)
        grad_w = grad_b = gh = None
        g = gy.contiguous()
        for w_frag, w_un, bias_pad, wt_frag, cols, rows_slice in chunks:
            # each group reads x at its own columns only (the others were not touched by it): the saved input serves all
            g, gh_c, gw_c, gb_c = ops.rq_fused_linear_backward(x, hidden, g, gl, (w_frag, w_un, bias_pad), wt_frag, cols,
                                                               tails=layer.tails, **kw)
            gh = gh_c if gh is None else gh + gh_c
            if len(chunks) == 1:
                grad_w, grad_b = gw_c[:, :lin.in_features], gb_c
            else:
                if grad_w is None:
                    grad_w, grad_b = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
                grad_w[rows_slice] = gw_c[:, :lin.in_features]
                grad_b[rows_slice] = gb_c
        by_id = {id(lin.weight): grad_w, id(lin.bias): grad_b}
        if net.hip_hidden_backward_supported():
            # hidden stack in fc_resnet_hidden_backward: activations recomputed from the identity columns in registers
            # (the gradient wrt the identity columns goes straight into g: g owns its storage here -- it is the gx output
            #  of the fused backward launch above, never the caller's grad_outputs)
            _, gw0, gwb, gb = ops.resnet_hidden_backward(x, gh, layer._id_cols(x.device), net.hidden_backward_packed(),
                                                         net.initial_layer.in_features, len(net.blocks), grad_inputs_accum=g)
            hf = net.hidden_features
            by_id[id(net.initial_layer.weight)] = gw0[:hf]
            by_id[id(net.initial_layer.bias)] = gb[0, :hf]
            for bi, block in enumerate(net.blocks):
                for li, l2 in enumerate(block.linear_layers):
                    by_id[id(l2.weight)] = gwb[2 * bi + li, :hf, :hf]
                    by_id[id(l2.bias)] = gb[1 + 2 * bi + li, :hf]
            return (None, g[:n]) + tuple(by_id.get(pid) for pid in ctx.param_ids)
        # otherwise: recompute its forward from the identity columns (PyTorch-ROCm ops) and pull gh through it
        hidden_params = [p for p in net.parameters() if p is not lin.weight and p is not lin.bias]
        with torch.enable_grad():
            xid = x[:, layer.identity_features].detach().requires_grad_(True)
            h2 = net.hidden(xid)
            wanted = [xid] + [p for p in hidden_params if p.requires_grad]
            got = torch.autograd.grad(h2, wanted, gh[:, :h2.shape[1]], allow_unused=True)
        g[:, layer.identity_features] += got[0]
        for p, gp_ in zip(wanted[1:], got[1:]):
            by_id[id(p)] = gp_
        grads = tuple(by_id.get(pid) for pid in ctx.param_ids)
        return (None, g[:n]) + grads


def _divisor_if_hidden_features(net):
    """Only ``hidden_features`` triggers the scaling in the linear-tailed siblings (coupling.py:438-440, :483-485)."""
    return float(np.sqrt(net.hidden_features)) if hasattr(net, "hidden_features") else 1.0


class PiecewiseLinearCouplingTransform(PiecewiseCouplingTransform):
    """Piecewise-linear CDF coupling (Mueller et al. 2018; coupling.py:299-353)."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None):
        self.num_bins = num_bins
        self.tails = tails
        self.tail_bound = tail_bound
        if apply_unconditional_transform:
            from flowconductor_amd.transforms.nonlinearities import PiecewiseLinearCDF

            def unconditional_transform(features):
                return PiecewiseLinearCDF(shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                                          tails=tails, tail_bound=tail_bound)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins

    def _coupling_kernel(self, inputs, transform_params, inverse):
        return ops.piecewise_spline_autograd(inputs, transform_params, self._cols(inputs.device), kind=ops.SPLINE_LINEAR,
                                    num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                                    inverse=inverse)


class PiecewiseQuadraticCouplingTransform(PiecewiseCouplingTransform):
    """Piecewise-quadratic CDF coupling (coupling.py:356-447)."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None, min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT):
        self.num_bins = num_bins
        self.tails = tails
        self.tail_bound = tail_bound
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        if apply_unconditional_transform:
            from flowconductor_amd.transforms.nonlinearities import PiecewiseQuadraticCDF

            def unconditional_transform(features):
                return PiecewiseQuadraticCDF(shape=[features] + (img_shape if img_shape else []),
                                             num_bins=num_bins, tails=tails, tail_bound=tail_bound,
                                             min_bin_width=min_bin_width, min_bin_height=min_bin_height)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return ops.spline_multiplier(ops.SPLINE_QUADRATIC, self.num_bins, self.tails)

    def _coupling_kernel(self, inputs, transform_params, inverse):
        div = _divisor_if_hidden_features(self.transform_net)
        return ops.piecewise_spline_autograd(inputs, transform_params, self._cols(inputs.device),
                                    kind=ops.SPLINE_QUADRATIC, num_bins=self.num_bins, tails=self.tails,
                                    tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                                    min_bin_height=self.min_bin_height, width_divisor=div, height_divisor=div,
                                    inverse=inverse)


class PiecewiseCubicCouplingTransform(PiecewiseCouplingTransform):
    """Monotone piecewise-cubic coupling (coupling.py:450-499)."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None, min_bin_width=ops.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=ops.DEFAULT_MIN_BIN_HEIGHT):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tails = tails
        self.tail_bound = tail_bound
        if apply_unconditional_transform:
            from flowconductor_amd.transforms.nonlinearities import PiecewiseCubicCDF

            def unconditional_transform(features):
                return PiecewiseCubicCDF(shape=[features] + (img_shape if img_shape else []), num_bins=num_bins,
                                         tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                                         min_bin_height=min_bin_height)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _coupling_kernel(self, inputs, transform_params, inverse):
        div = _divisor_if_hidden_features(self.transform_net)
        return ops.piecewise_spline_autograd(inputs, transform_params, self._cols(inputs.device), kind=ops.SPLINE_CUBIC,
                                    num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                                    min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                    width_divisor=div, height_divisor=div, inverse=inverse)
