"""Residual MLP conditioner (stays on PyTorch-ROCm; boundary only).

Same constructor arguments, attribute names and ``state_dict`` keys as the reference's
flowcon/nn/nets/resnet.py:9-100 (``initial_layer``, ``blocks.N.linear_layers.M``,
``blocks.N.context_layer``, ``final_layer``, attribute ``hidden_features`` that the spline
coupling layers look for), and the same parameter construction order so that a given
``torch.manual_seed`` yields the same default initialisation.
"""
import torch
from torch import nn
from torch.nn import functional as F

from flowconductor_amd import ops


class ResidualBlock(nn.Module):
    """pre-activation block: x + W1 drop(act(bn(W0 act(bn(x))))) with an optional GLU gate on context."""

    def __init__(self, features, context_features, activation=torch.nn.ReLU(),
                 dropout_probability=0.0, use_batch_norm=False, zero_initialization=True):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList(nn.BatchNorm1d(features, eps=1e-3) for _ in range(2))
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList(nn.Linear(features, features) for _ in range(2))
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            last = self.linear_layers[-1]
            nn.init.uniform_(last.weight, -1e-3, 1e-3)
            nn.init.uniform_(last.bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        h = inputs
        for i in range(2):
            if self.use_batch_norm:
                h = self.batch_norm_layers[i](h)
            h = self.activation(h)
            if i == 1:
                h = self.dropout(h)
            h = self.linear_layers[i](h)
        if context is not None:
            h = F.glu(torch.cat((h, self.context_layer(context)), dim=1), dim=1)
        return inputs + h


class ResidualNet(ops.RuntimeCaches, nn.Module):
    """Linear -> num_blocks x ResidualBlock -> Linear, for 1-dim feature vectors."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None,
                 num_blocks=2, activation=torch.nn.ReLU(), dropout_probability=0.0,
                 use_batch_norm=False):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        first_in = in_features if context_features is None else in_features + context_features
        self.initial_layer = nn.Linear(first_in, hidden_features)
        self.blocks = nn.ModuleList(
            ResidualBlock(features=hidden_features, context_features=context_features,
                          activation=activation, dropout_probability=dropout_probability,
                          use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)
        )
        self.final_layer = nn.Linear(hidden_features, out_features)

    def hidden(self, inputs, context=None):
        """Everything up to (not including) ``final_layer``: the [N, hidden_features] activation that the
        fused final-layer + spline kernel consumes (``forward`` = ``final_layer(hidden(...))``)."""
        if context is not None:
            inputs = torch.cat((inputs, context), dim=1)
        h = self.initial_layer(inputs)
        for block in self.blocks:
            h = block(h, context=context)
        return h

    def forward(self, inputs, context=None):
        if self._hip_forward_ok(inputs, context):
            # inference on a HIP device: hidden stack in fc_resnet_hidden, the < 16 leftover rows on PyTorch
            n = inputs.shape[0]
            body = n - n % 16
            ids = getattr(self, "_all_cols", None)
            if ids is None or ids.device != inputs.device or ids.numel() != inputs.shape[1]:
                ids = self._all_cols = torch.arange(inputs.shape[1], dtype=torch.int32, device=inputs.device)
            hidden = self.hidden_hip(inputs[:body], ids, None if context is None else context[:body])
            if body < n:
                hidden = torch.cat((hidden, self.hidden_padded(inputs[body:],
                                                               None if context is None else context[body:])))
            return self.final_from_padded(hidden)
        return self.final_layer(self.hidden(inputs, context))

    def _hip_forward_ok(self, inputs, context):
        from flowconductor_amd import options

        from flowconductor_amd import ops

        if not (inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.shape[0] >= 16
                and options.get("fused_hidden") and not ops.has_hooks(self)):
            return False
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False        # training: PyTorch autograd
        return inputs.shape[1] + (0 if context is None else self.context_features or 0) == self.initial_layer.in_features \
            and self.hip_hidden_supported(inputs.shape[1], context)

    def train(self, mode=True):
        """Switching into or out of training mode drops the packed-weight caches (``ops.invalidate_hip_caches``)."""
        from flowconductor_amd import ops

        ops.invalidate_hip_caches()
        return super().train(mode)

    # ---- device fast path for the hidden layers (inference) ------------------------------------------
    def hip_hidden_supported(self, features_total, context=None):
        """True when ``fc_resnet_hidden`` covers this net: hidden <= 64 (narrower nets run zero-padded in the
        64-wide kernel), <= 4 blocks, an activation ``ops.activation_code`` knows (ReLU, tanh, SiLU, ELU, LeakyReLU,
        sigmoid; the same in every block), no batch norm, dropout
        inactive, input width (identity features + context features) <= 64; a context must be a [N, C <= 32] f32
        tensor matching ``context_features`` and allows <= 3 blocks (the gate layers' fragments share the LDS)."""
        from flowconductor_amd import ops

        def known(f):
            return ops.activation_code(f) is not None

        if self.hidden_features > 64 or len(self.blocks) > 4:
            return False
        in_f = self.initial_layer.in_features
        if (self.context_features is None) != (context is None):
            return False        # the module itself raises / ignores: leave that to PyTorch
        if context is not None:
            c = self.context_features
            if len(self.blocks) > 3:
                return False
            if (context.dim() != 2 or context.shape[1] != c or c > 32 or context.dtype != torch.float32
                    or not context.is_cuda or context.requires_grad and torch.is_grad_enabled()):
                return False
            if in_f - c <= 0 or in_f - c > features_total:
                return False
        if in_f > 64 or (context is None and in_f > features_total):
            return False
        for block in self.blocks:
            if block.use_batch_norm or not known(block.activation):
                return False
            if ops.activation_code(block.activation) != ops.activation_code(self.blocks[0].activation):
                return False
            if block.dropout.p > 0 and self.training:
                return False
        return True

    def hip_hidden_wide_supported(self, features_total, context=None):
        """True when ``fc_resnet_hidden_wide`` covers this net: 64 < hidden <= 256 (zero-padded to 128 / 256), no
        context, no batch norm, dropout inactive, a known activation, <= 64 input features, <= 16 blocks."""
        from flowconductor_amd import ops

        if not 64 < self.hidden_features <= 256 or context is not None or self.context_features is not None:
            return False
        in_f = self.initial_layer.in_features
        if in_f > 64 or in_f > features_total or len(self.blocks) > 16:
            return False
        for block in self.blocks:
            code = ops.activation_code(block.activation)
            if block.use_batch_norm or code is None or code != ops.activation_code(self.blocks[0].activation):
                return False
            if block.dropout.p > 0 and self.training:
                return False
        return True

    def hidden_hip_wide(self, rows, id_cols):
        """h [N, 128 or 256] from FULL input rows + the identity column indices (N a multiple of 64); columns
        ``hidden_features``.. are zero."""
        from flowconductor_amd import ops

        width = ops.general_hidden_width(self.hidden_features)
        key = ops.cache_key(*self._param_list(), extra=("wide", width))
        if getattr(self, "_hip_packed_wide", None) is None or self._hip_packed_wide[0] != key:
            self._hip_packed_wide = (key, ops.pack_resnet_hidden_wide(self, width))
        act = ops.activation_code(self.blocks[0].activation) if len(self.blocks) else (ops.ACT_RELU, 0.0)
        return ops.resnet_hidden_wide(rows, id_cols, self._hip_packed_wide[1], self.initial_layer.in_features,
                                      len(self.blocks), width, act)

    def hip_hidden_backward_supported(self):
        """True when ``fc_resnet_hidden_backward`` covers this net: hidden <= 64, <= 2 blocks, ReLU, no context, no batch
        norm, dropout inactive."""
        from flowconductor_amd import ops

        if self.hidden_features > 64 or len(self.blocks) > 2 or self.context_features is not None:
            return False
        if self.initial_layer.in_features > 64:
            return False
        for block in self.blocks:
            code = ops.activation_code(block.activation)
            if block.use_batch_norm or code is None or code[0] != ops.ACT_RELU or (block.dropout.p > 0 and self.training):
                return False
        return True

    def _storage_key(self):
        """Where the parameters live (a device pack plan holds raw pointers to these storages)."""
        return tuple(p.data_ptr() for p in self._param_list())

    def _param_list(self):
        """``ops.param_list``: the memoised parameter tuple, valid while every slot holds the same Parameter object."""
        return ops.param_list(self)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float(): new storages (and possibly new Parameter objects)
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_fc_param_list", None)
        return out

    def hidden_backward_packed(self):
        from flowconductor_amd import ops

        # persistent device-side pack plan (one launch per refresh); rebuilt when the parameter storages moved
        plan = self.hidden_backward_plan()
        plan[1].refresh()
        return plan[2]

    def hidden_backward_plan(self):
        """[where, DevicePack, packed] of the training-time images (not refreshed here)."""
        from flowconductor_amd import ops

        where = self._storage_key()
        plan = getattr(self, "_hip_packed_bwd", None)
        if plan is None or plan[0] != where:
            pack, packed = ops.device_pack_resnet_hidden_backward(self)
            plan = self._hip_packed_bwd = [where, pack, packed]
        return plan

    def hidden_padded(self, inputs, context=None):
        """``hidden`` on PyTorch, zero-padded to the kernel's 64 columns (leftover rows next to ``hidden_hip``)."""
        h = self.hidden(inputs, context)
        return h if h.shape[1] == 64 else F.pad(h, (0, 64 - h.shape[1]))

    def final_from_padded(self, hidden64):
        """``final_layer`` applied to a [N, 64] zero-padded hidden activation."""
        from flowconductor_amd import ops

        lin = self.final_layer
        if lin.in_features == 64:
            return lin(hidden64)
        key = ops.cache_key(lin.weight)
        if getattr(self, "_final_padded", None) is None or self._final_padded[0] != key:
            self._final_padded = (key, F.pad(lin.weight.detach(), (0, 64 - lin.in_features)))
        return F.linear(hidden64, self._final_padded[1], lin.bias)

    def hidden_hip(self, rows, id_cols, context=None):
        """h [N, 64] from FULL input rows + the identity column indices (N a multiple of 16) [+ context rows]; for a
        narrower net columns ``hidden_features``.. are zero."""
        from flowconductor_amd import ops

        in_features = self.initial_layer.in_features - (self.context_features or 0)
        act = ops.activation_code(self.blocks[0].activation) if len(self.blocks) else (ops.ACT_RELU, 0.0)
        if context is None:
            # weight image made on the device (one launch per refresh), copied into LDS by every launch.  In training
            # mode the image is the forward part of the backward kernel's images (same layout): one refresh per
            # optimizer step serves both directions.
            # (`self.training` alone decides: the fused layer's autograd node runs its forward under no_grad)
            if self.training and self.hip_hidden_backward_supported():
                w_frag, _, w_un, bias_acc, _ = self.hidden_backward_packed()
                return ops.resnet_hidden_packed(rows, id_cols, (w_frag, w_un, bias_acc), in_features, len(self.blocks), act)
            where = self._storage_key()
            plan = getattr(self, "_hip_image", None)
            if plan is None or plan[0] != where:
                pack, packed = ops.device_pack_resnet_hidden_forward(self)
                plan = self._hip_image = [where, pack, packed]
            plan[1].refresh()
            return ops.resnet_hidden_packed(rows, id_cols, plan[2], in_features, len(self.blocks), act)
        key = ops.cache_key(*self._param_list())
        if getattr(self, "_hip_packed", None) is None or self._hip_packed[0] != key:
            self._hip_packed = (key, ops.pack_resnet_hidden(self))
        return ops.resnet_hidden(rows, id_cols, self._hip_packed[1], in_features, len(self.blocks), context, act)
