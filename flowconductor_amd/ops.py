"""Functional host wrappers over the HIP kernels (tensor in -> tensor out).

The ``Transform`` modules in ``flowconductor_amd.transforms`` call these; they allocate the
outputs with torch, pass raw device pointers + the current HIP stream through the C ABI and
turn the device error word into the reference's Python exceptions.
"""
import contextlib
import ctypes
import math
import threading

import numpy as np
import torch

from flowconductor_amd import _hip, options


class InverseNotAvailable(Exception):
    """Exception to be thrown when a transform does not have an inverse."""


class InputOutsideDomain(Exception):
    """Exception to be thrown when the input to a transform is not within its domain."""


# ---- device error word -> Python exceptions ------------------------------------------------
#
# The reference raises synchronously inside each spline call (four host syncs per layer on a
# GPU).  Here kernels OR bits into one device word; a stand-alone transform
# call reads it right away, a CompositeTransform / Flow defers the read to once per cascade.

_state = threading.local()


def _flags():
    if not hasattr(_state, "flags"):
        _state.flags = {}
        _state.depth = 0
        _state.dirty = set()
    return _state.flags


def _flag_for(device):
    flags = _flags()
    key = (device.type, device.index)
    t = flags.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        flags[key] = t
    return t


def _raise_for(bits):
    if bits & _hip.ERR_OUTSIDE_DOMAIN:
        raise InputOutsideDomain()
    if bits & _hip.ERR_DISCRIMINANT:
        raise AssertionError("rational-quadratic inverse: negative discriminant")
    if bits & _hip.ERR_NONFINITE:
        raise FloatingPointError("non-finite value inside a bijector kernel")


def _check_now():
    flags = _flags()
    dirty, _state.dirty = _state.dirty, set()
    bits = 0
    for key in dirty:
        t = flags[key]
        bits |= int(t.item())
        if bits:
            t.zero_()
    if bits:
        _raise_for(bits)


@contextlib.contextmanager
def deferred_errors():
    """Read the device error word once when the outermost block exits."""
    _flags()
    _state.depth += 1
    try:
        yield
    except BaseException:
        _state.depth -= 1
        if _state.depth == 0:
            for key in _state.dirty:
                _state.flags[key].zero_()
            _state.dirty = set()
        raise
    else:
        _state.depth -= 1
        if _state.depth == 0:
            _check_now()


@contextlib.contextmanager
def capture_mode():
    """Inside a HIP-graph capture the error word cannot be read (no host sync): kernels still OR their bits into
    it, ``check_errors`` reads it after a replay."""
    _flags()
    _state.depth += 1
    try:
        yield
    finally:
        _state.depth -= 1
        _state.dirty = set()


def check_errors(device):
    """Read (and clear) the device error word of ``device`` now; raises the reference's exceptions."""
    _flags()
    _state.dirty.add((device.type, device.index))
    _flag_for(device)
    _check_now()


def _err_word(device, may_raise):
    """Device pointer for the kernel's error word (None when the op cannot raise)."""
    if not may_raise:
        return None
    _flags()
    _state.dirty.add((device.type, device.index))
    return _flag_for(device)


def _finish(may_raise):
    if may_raise and _state.depth == 0:
        _check_now()


# ---- helpers --------------------------------------------------------------------------------

def _as_cols(cols, device):
    if cols is None:
        return None
    if cols.dtype != torch.int32 or cols.device != device or not cols.is_contiguous():
        cols = cols.to(device=device, dtype=torch.int32).contiguous()
    return cols


def _prep_2d(inputs, name="inputs", align16=False):
    x = _hip.dev_f32(inputs, name)
    if x.dim() != 2:
        raise ValueError("%s must be [batch, features], got shape %s" % (name, tuple(x.shape)))
    return _aligned16(x) if align16 else x


def _aligned16(t):
    """The matrix-core kernels move rows as 16-byte pieces: a contiguous view that starts off a 16-byte boundary
    (``data[1:]`` with a feature count that is not a multiple of 4) is copied to a fresh allocation first."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


# ---- packed-weight caches ---------------------------------------------------------------------------------------
#
# Layers cache kernel-layout copies of their weights (pre-masked, zero-padded, folded matrices), keyed on the
# parameters' version counters and storage pointers.  In-place writes THROUGH ``.data`` (``p.data.copy_(ema)``,
# ``p.data.clamp_()``, some checkpoint loaders) change neither; ``invalidate_hip_caches()`` bumps an epoch that is
# part of every key.  ``nn.Module.train()`` of this package's modules calls it, so a cache never survives a switch
# into or out of training mode; after ``.data`` surgery on an eval-mode model call it yourself.

_cache_epoch = 0


def invalidate_hip_caches():
    """Drop every packed-weight cache of this package (re-packed on the next call that needs them)."""
    global _cache_epoch
    _cache_epoch += 1


def cache_epoch():
    return _cache_epoch


_paranoid_tick = 0


def cache_key(*tensors, extra=()):
    """Key of a packed copy of ``tensors``: version counter, storage pointer and device of each + the epoch.  With
    ``options.paranoid_caches`` no two keys are equal: every lookup misses and re-packs."""
    global _paranoid_tick
    if options.get("paranoid_caches"):
        _paranoid_tick += 1
        return (("paranoid", _paranoid_tick),) + (_cache_epoch,) + tuple(extra)
    return tuple((t._version, t.data_ptr(), t.device) for t in tensors) + (_cache_epoch,) + tuple(extra)


# Attributes in which this package's modules keep run-time state derived from their parameters (kernel-layout weight
# images, DevicePack plans with raw device pointers, index tensors, memo lists).  None of it is model state: it is
# rebuilt on demand and must not travel with ``copy.deepcopy`` / ``pickle`` / ``torch.save(module)``.
RUNTIME_CACHE_ATTRS = frozenset((
    "_hip_packed", "_hip_packed_wide", "_hip_packed_bwd", "_hip_image", "_final_padded", "_tail_image", "_train_pack",
    "_packed", "_masked_final", "_mm_cache", "_dense_cache", "_sp_cache", "_all_cols", "_ctx_cols", "_cols_cache",
    "_id_cols_cache", "_fc_param_list", "_fc_module_list", "_fc_static_ok"))


class RuntimeCaches:
    """Mixin (before ``nn.Module`` in the bases): ``copy.deepcopy``, ``pickle`` and ``torch.save`` of the module see the
    run-time caches of ``RUNTIME_CACHE_ATTRS`` as ``None`` -- a copy starts cold and re-packs from ITS OWN parameters; a
    checkpoint of the whole module holds parameters and buffers only (``fc_pack_job`` structs carry raw device
    pointers and cannot be pickled at all)."""

    def __getstate__(self):
        state = dict(super().__getstate__())
        for name in RUNTIME_CACHE_ATTRS:
            if state.get(name) is not None:
                state[name] = None
        return state


def param_list(module):
    """``tuple(module.parameters())`` memoised on the module (walking the module tree on every call is a third of the
    per-layer host time of a small batch) together with WHERE each parameter hangs: the memo is valid only while every slot
    still holds the same Parameter object (``lin.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)`` and late
    parametrizations replace objects without touching versions or pointers of the orphans) and the cache epoch stands;
    modules drop it in ``_apply`` (.to / .cuda / .float)."""
    memo = module.__dict__.get("_fc_param_list")
    if memo is None or memo[0] != _cache_epoch or not all(m._parameters.get(n) is p for m, n, p in memo[2]):
        slots = tuple((m, n, p) for m in module.modules() for n, p in m._parameters.items() if p is not None)
        memo = module.__dict__["_fc_param_list"] = (_cache_epoch, tuple(module.parameters()), slots)
    return memo[1]


def static_memo(module, slot, key, compute):
    """``compute()`` memoised on ``module`` under ``slot`` for as long as ``key`` and the cache epoch stand -- for the parts of
    a fast-path predicate that only depend on how the module is built (layer types, widths, activations, training flag):
    re-deriving them on every call was a fifth of the host time of a small batch."""
    memo = module.__dict__.get(slot)
    full = (_cache_epoch,) + tuple(key)
    if memo is None or memo[0] != full:
        memo = module.__dict__[slot] = (full, compute())
    return memo[1]


def structure_key(net):
    """The cheap MUTABLE inputs of a conditioner's fast-path predicate, for ``static_memo`` keys: per residual block the
    identity of its activation, its dropout probability and its own training flag (``block.train()`` / a swapped
    activation / ``dropout.p = 0.1`` after the first call must re-derive the predicate)."""
    blocks = getattr(net, "blocks", ())
    return (net.training, id(getattr(net, "activation", None))) + tuple(
        (id(getattr(b, "activation", None)), getattr(getattr(b, "dropout", None), "p", 0.0), b.training) for b in blocks)


def has_hooks(module):
    """True when ``module`` or a sub-module carries forward (pre-)hooks (old-style weight_norm refreshes ``weight``
    in one): the fast paths read the weights directly and never go through ``__call__``, so they step aside.
    (The sub-module list is kept on the module and rebuilt when the cache epoch moves or a child is added / removed:
    walking ``modules()`` on every call was the largest single item of the per-layer host time.)"""
    memo = module.__dict__.get("_fc_module_list")
    count = len(module._modules)
    if memo is None or memo[0] != _cache_epoch or memo[1] != count:
        memo = (_cache_epoch, count, tuple(module.modules()))
        module.__dict__["_fc_module_list"] = memo
    return any(m._forward_hooks or m._forward_pre_hooks for m in memo[2])


LAD_STORE, LAD_ACCUMULATE, LAD_STORE_NEG, LAD_ACCUMULATE_NEG = 0, 1, 2, 3


class KernelTimer:
    """Times every launch of one C-ABI entry point with HIP events recorded on the stream the
    kernel is launched on (the current torch stream).  Used by bench.py for ``roofline``."""

    _active = []

    def __init__(self, name):
        self.name = name
        self.pairs = []

    def __enter__(self):
        KernelTimer._active.append(self)
        return self

    def __exit__(self, *exc):
        KernelTimer._active.remove(self)
        return False

    def durations_ms(self):
        """Per-launch durations; call after the stream has been synchronised."""
        return [a.elapsed_time(b) for a, b in self.pairs]


def _call(name, fn, device, *args):
    """Launch C-ABI entry ``fn`` (asynchronous), bracketing it with events for active timers."""
    if device.index is not None and device.index != torch.cuda.current_device():
        # the launchers size grids and set kernel attributes for the CURRENT device: make it the tensors' device
        with torch.cuda.device(device):
            return _call(name, fn, device, *args)
    timers = [t for t in KernelTimer._active if t.name == name]
    if timers:
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream(device))
    code = fn(*args)
    if timers:
        end.record(torch.cuda.current_stream(device))
        for t in timers:
            t.pairs.append((start, end))
    _hip.check(code, name)


# ---- rational-quadratic spline ----------------------------------------------------------------

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def _rq_config(num_bins, tails, tail_bound, box, min_bin_width, min_bin_height, min_derivative,
               enable_identity_init, wh_divisor, inverse):
    cfg = _hip.RQConfig()
    cfg.num_bins = num_bins
    cfg.tails = 0 if tails is None else 1
    cfg.inverse = 1 if inverse else 0
    if tails == "linear":
        cfg.left, cfg.right, cfg.bottom, cfg.top = -tail_bound, tail_bound, -tail_bound, tail_bound
    else:
        cfg.left, cfg.right, cfg.bottom, cfg.top = box
    cfg.min_bin_width = min_bin_width
    cfg.min_bin_height = min_bin_height
    cfg.min_derivative = min_derivative
    cfg.wh_divisor = wh_divisor
    cfg.softplus_beta = (math.log(2) / (1 - min_derivative)) if enable_identity_init else 1.0
    cfg.tail_constant = float(np.log(np.exp(1 - min_derivative) - 1))
    return cfg


def rq_spline(inputs, params, cols=None, *, num_bins, tails=None, tail_bound=1.0,
              left=0.0, right=1.0, bottom=0.0, top=1.0,
              min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
              min_derivative=DEFAULT_MIN_DERIVATIVE, enable_identity_init=False,
              wh_divisor=1.0, inverse=False, shared_params=False, out=None):
    """RQ spline over ``inputs[:, cols]`` (all columns if ``cols`` is None).

    ``params``: ``[N, d_t * (3K -/+ 1)]`` per-sample rows, or ``[d_t * (3K -/+ 1)]`` with
    ``shared_params``.  Returns ``(outputs [N, D], logabsdet [N])``; other columns are copied.
    Semantics: reference splines/rational_quadratic.py:13-181.
    """
    lib = _hip.load()
    x = _prep_2d(inputs)
    p = _hip.dev_f32(params, "params")
    _hip.require_no_grad(inputs, params)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = d if cols is None else cols.numel()
    if tails is None:
        mult = 3 * num_bins + 1
    elif tails == "linear":
        mult = 3 * num_bins - 1
    else:
        raise RuntimeError("{} tails are not implemented.".format(tails))
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    rowlen = d_t * mult
    want = rowlen if shared_params else n * rowlen
    if p.numel() != want:
        raise ValueError("params has %d elements, expected %d" % (p.numel(), want))

    cfg = _rq_config(num_bins, tails, tail_bound, (left, right, bottom, top), min_bin_width, min_bin_height,
                     min_derivative, enable_identity_init, wh_divisor, inverse)

    if options.get("rq_force_tile"):
        cfg.flags |= 2  # FC_RQ_FORCE_TILE
    y = torch.empty_like(x) if out is None else out
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    err = _err_word(x.device, True)
    _call("fc_rq_spline", lib.fc_rq_spline, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p),
          _hip.ptr(cols), _hip.ptr(lad), _hip.ptr(err), n, d, d_t, 1 if shared_params else 0,
          LAD_STORE, cfg, _hip.stream_ptr(x.device))
    _finish(True)
    return y, lad


def rq_spline_backward(inputs, params, cols, grad_outputs, grad_logabsdet, **kw):
    """Gradients of ``rq_spline(inputs, params, cols, **kw)`` (forward direction, per-sample params):
    returns ``(grad_inputs [N, D], grad_params [N, d_t * P])``; identity columns pass ``grad_outputs`` through."""
    lib = _hip.load()
    x = _prep_2d(inputs.detach())
    p = _hip.dev_f32(params.detach(), "params")
    gy = _hip.dev_f32(grad_outputs, "grad_outputs")
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = d if cols is None else cols.numel()
    tails = kw.get("tails")
    cfg = _rq_config(kw["num_bins"], tails, kw.get("tail_bound", 1.0),
                     (kw.get("left", 0.0), kw.get("right", 1.0), kw.get("bottom", 0.0), kw.get("top", 1.0)),
                     kw.get("min_bin_width", DEFAULT_MIN_BIN_WIDTH), kw.get("min_bin_height", DEFAULT_MIN_BIN_HEIGHT),
                     kw.get("min_derivative", DEFAULT_MIN_DERIVATIVE), kw.get("enable_identity_init", False),
                     kw.get("wh_divisor", 1.0), False)
    gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
    gx = torch.empty_like(gy)             # every column is written (identity columns: grad_outputs)
    gp = torch.empty_like(p)
    _call("fc_rq_spline_backward", lib.fc_rq_spline_backward, x.device, _hip.ptr(x), _hip.ptr(p), _hip.ptr(cols),
          _hip.ptr(gy), _hip.ptr(gl), _hip.ptr(gx), _hip.ptr(gp), n, d, d_t, cfg, _hip.stream_ptr(x.device))
    return gx, gp.view_as(params)


class _RQSplineFunction(torch.autograd.Function):
    """``rq_spline`` with gradients (forward direction): the HIP forward and backward kernels behind autograd, so
    that ``-flow.log_prob(x).mean().backward()`` trains through this path (reference: examples/toy_2d.py:57-68)."""

    @staticmethod
    def forward(ctx, inputs, params, cols, kw):
        with torch.no_grad():
            outputs, logabsdet = rq_spline(inputs, params, cols, **kw)
        ctx.save_for_backward(inputs, params)
        ctx.cols, ctx.kw = cols, kw
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, params = ctx.saved_tensors
        if grad_outputs is None:
            grad_outputs = torch.zeros_like(inputs)
        gx, gp = rq_spline_backward(inputs, params, ctx.cols, grad_outputs.contiguous(), grad_logabsdet, **ctx.kw)
        return gx, gp, None, None


def _inverse_through_forward(forward_fn, inverse_nograd_fn, inputs, params):
    """Differentiable inverse of an element-wise bijector whose gradients exist for the forward direction only
    (sampling / reverse-KL training through ``Flow.sample``, ``sample_and_log_prob``): the inverse kernel finds
    ``y0 = f^-1(x; p)`` without a graph, then one Newton-shaped step ``y = y0 - (f(y0; p) - x) / f'(y0)`` through the
    differentiable forward op carries the implicit-function gradients ``dy/dx = 1 / f'``, ``dy/dp = -(df/dp) / f'``
    (its value only removes the inverse's rounding residual), and ``logabsdet = -logabsdet_f(y; p)`` is the forward op
    evaluated at that ``y``.  Three kernel passes instead of one; ``f'`` per element comes from the backward kernel."""
    with torch.no_grad():
        y0, _ = inverse_nograd_fn(inputs, params)
    y0 = y0.detach().requires_grad_(True)
    with torch.enable_grad():
        x_hat, _ = forward_fn(y0, params)
        fprime, = torch.autograd.grad(x_hat.sum(), y0, retain_graph=True)
        y = y0.detach() - (x_hat - inputs) / fprime.detach()
        _, logabsdet = forward_fn(y, params)
    return y, -logabsdet


def rq_spline_autograd(inputs, params, cols=None, **kw):
    """``rq_spline`` that records an autograd node when gradients are required (per-sample parameters; the inverse
    direction through ``_inverse_through_forward``); otherwise exactly ``rq_spline``."""
    needs = torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad)
    if not needs:
        return rq_spline(inputs, params, cols, **kw)
    if kw.get("shared_params"):
        raise RuntimeError("flowconductor_amd: gradients are implemented for the RQ spline with per-sample "
                           "parameters; wrap other calls in torch.no_grad().")
    if kw.get("inverse"):
        fwd_kw = dict(kw, inverse=False)
        return _inverse_through_forward(lambda y, p: _RQSplineFunction.apply(y, p, cols, fwd_kw),
                                        lambda x, p: rq_spline(x, p, cols, **kw), inputs, params)
    return _RQSplineFunction.apply(inputs, params, cols, kw)


FUSED_ROWS, FUSED_HIDDEN, FUSED_DT, FUSED_BINS = 32, 64, 32, 8


def fused_linear_supported(n, d, d_t, hidden, num_bins, tails):
    """Shapes the fused final-layer + RQ-spline kernel is specialised for (the north-star layer)."""
    return (1 <= hidden <= FUSED_HIDDEN and 1 <= d_t <= FUSED_DT and num_bins == FUSED_BINS and tails == "linear"
            and d <= 128 and n >= FUSED_ROWS)


HIDDEN_ROWS = 16


def _pad_to(t, shape):
    """``t`` zero-padded at the end of every dim up to ``shape`` (returns ``t`` itself when nothing is missing)."""
    if tuple(t.shape) == tuple(shape):
        return t.contiguous()
    out = t.new_zeros(shape)
    out[tuple(slice(0, k) for k in t.shape)] = t
    return out


def pack_resnet_hidden(net):
    """Weights of the hidden layers of a ResidualNet (hidden <= 64, <= 4 blocks) as ``fc_resnet_hidden`` takes
    them: the nn.Linear tensors row-major, the block layers stacked [blocks, 2, 64, 64] / [blocks, 2, 64]; with a
    context also the blocks' ``context_layer`` stacked [blocks, 64, C] / [blocks, 64].  A narrower net is embedded
    in the 64-wide kernel by zero padding: the extra hidden units have zero weights and biases, stay 0 through ReLU
    and the residual stream, and meet zero columns in every following layer."""
    hw = FUSED_HIDDEN
    w0 = _pad_to(net.initial_layer.weight.detach(), (hw, net.initial_layer.in_features))
    b0 = _pad_to(net.initial_layer.bias.detach(), (hw,))
    ws, bs, wcs, bcs = [], [], [], []
    for block in net.blocks:
        for lin in block.linear_layers:
            ws.append(_pad_to(lin.weight.detach(), (hw, hw)))
            bs.append(_pad_to(lin.bias.detach(), (hw,)))
        if getattr(block, "context_layer", None) is not None:
            cl = block.context_layer
            wcs.append(_pad_to(cl.weight.detach(), (hw, cl.in_features)))
            bcs.append(_pad_to(cl.bias.detach(), (hw,)))
    wb = torch.stack(ws).contiguous() if ws else None
    bb = torch.stack(bs).contiguous() if bs else None
    wc = torch.stack(wcs).contiguous() if wcs else None
    bc = torch.stack(bcs).contiguous() if bcs else None
    return w0, b0, wb, bb, wc, bc


ACT_RELU, ACT_TANH, ACT_SILU, ACT_ELU, ACT_LEAKY_RELU, ACT_SIGMOID = range(6)


def activation_code(fn):
    """``(FC_ACT_* code, parameter)`` of an activation callable / module the hidden-layer kernel knows, else None."""
    import torch.nn as nn
    from torch.nn import functional as F

    if isinstance(fn, nn.ReLU) or fn in (F.relu, torch.relu):
        return ACT_RELU, 0.0
    if isinstance(fn, nn.Tanh) or fn in (torch.tanh, F.tanh):
        return ACT_TANH, 0.0
    if isinstance(fn, nn.SiLU) or fn is F.silu:
        return ACT_SILU, 0.0
    if isinstance(fn, nn.ELU):
        return ACT_ELU, float(fn.alpha)
    if fn is F.elu:
        return ACT_ELU, 1.0
    if isinstance(fn, nn.LeakyReLU):
        return ACT_LEAKY_RELU, float(fn.negative_slope)
    if fn is F.leaky_relu:
        return ACT_LEAKY_RELU, 0.01
    if isinstance(fn, nn.Sigmoid) or fn in (torch.sigmoid, F.sigmoid):
        return ACT_SIGMOID, 0.0
    return None


CONTEXT_GLU, CONTEXT_ADDITIVE = 1, 2

WIDE_ROWS = 64


def _exact_pow2(shift):
    """2^shift as float32, built from the exponent bits (``torch.ldexp`` goes through ``pow`` and is not exact on every
    backend)."""
    return ((shift.to(torch.int32) + 127) << 23).view(torch.float32)


def _pow2_scale(m):
    """fc_split.h pow2_scale on a tensor of maxima: (scale, unscale), exact powers of two lifting each into [2^14, 2^15)."""
    _, exp = torch.frexp(m)                                   # m = mant * 2^exp, mant in [0.5, 1)
    ok = (m > 0) & torch.isfinite(m) & (exp >= -111)
    shift = torch.where(ok, 15 - exp, torch.zeros_like(exp))
    return _exact_pow2(shift), _exact_pow2(-shift)


def _a_fragments(w):
    """[rows (multiple of 16), K (multiple of 32)] f32, already scaled -> f16 [rows/16, K/32, 2 (hi, lo), 64, 8]: the
    matrix-core A fragments of v_mfma_f32_16x16x32_f16 (lane l holds row l & 15, k = 32 kstep + 8 (l >> 4) + j)."""
    rows, k = w.shape
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)

    def frag(piece):
        return piece.reshape(rows // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(rows // 16, k // 32, 64, 8)

    return torch.stack((frag(hi), frag(lo)), dim=2).contiguous()


def pack_resnet_hidden_wide(net, width):
    """Hidden layers of a ResidualNet with 64 < hidden_features <= 256 (no context) as ``fc_resnet_hidden_wide``
    streams them: every layer's weight zero-padded to ``width`` (128 / 256) rows and 32-multiples of columns, scaled
    by a power of two per layer, split into two f16 pieces, in fragment order; all layers in one flat f16 buffer.
    Returns (w_frag, w_unscale [1 + 2 blocks], bias [1 + 2 blocks, width])."""
    layers = [net.initial_layer] + [lin for block in net.blocks for lin in block.linear_layers]
    frags, uns, biases = [], [], []
    for i, lin in enumerate(layers):
        kin = (32 if lin.in_features <= 32 else 64) if i == 0 else width
        w = _pad_to(lin.weight.detach().float(), (width, kin))
        sc, un = _pow2_scale(w.abs().amax().reshape(1))
        frags.append(_a_fragments(w * sc).reshape(-1))
        uns.append(un)
        biases.append(_pad_to(lin.bias.detach().float(), (width,)))
    return torch.cat(frags).contiguous(), torch.cat(uns).float().contiguous(), torch.stack(biases).contiguous()


HIDDEN_BWD_ROWS = 128

PACK_FINAL, PACK_FINAL_T, PACK_HIDDEN, PACK_HIDDEN_T, PACK_HIDDEN_T0 = range(5)


class DevicePack:
    """A set of ``fc_pack_job``s with persistent output buffers: ``run()`` re-packs all of them in ONE launch (training
    re-packs every optimizer step; with tensor ops that is ~100 tiny launches per coupling layer).  The source pointers
    are the parameters' storages, which optimizers update in place."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.keep = []            # tensors the jobs point into
        self.sources = []         # the parameters the jobs read (their versions say when a refresh is due)
        self._jobs_dev = None
        self._root = None         # the pack this one was merged into
        self.children = []        # packs merged into this one (they keep their own jobs)
        self._key = None
        self.prepare = []         # callables run before every launch (staging copies the jobs read from)

    def add(self, mode, weight, bias, frag, unscale, bias_out=None, p=0, pp=0, nks=0, nt=0, group=0, track=True):
        """``track=False``: ``weight`` / ``bias`` are staging copies refreshed by a ``prepare`` callable -- the caller lists
        the tensors they are made from in ``sources`` instead."""
        if not weight.is_contiguous() or weight.dtype != torch.float32:
            raise ValueError("DevicePack sources must be contiguous float32 tensors")
        job = _hip.PackJob()
        job.w, job.b = weight.data_ptr(), (0 if bias is None else bias.data_ptr())
        job.frag, job.unscale = frag.data_ptr(), unscale.data_ptr()
        job.bias_out = 0 if bias_out is None else bias_out.data_ptr()
        job.rows, job.cols = weight.shape
        job.mode, job.p, job.pp, job.nks, job.nt, job.group = mode, p, pp, nks, nt, group
        self.jobs.append(job)
        self.keep += [weight, bias, frag, unscale, bias_out]
        if track:
            self.sources += [t for t in (weight, bias) if t is not None]

    def root(self):
        node = self
        while node._root is not None:
            node = node._root
        return node

    def _invalidate(self):
        self._jobs_dev = self._key = None

    def _walk(self):
        yield self
        for child in self.children:
            yield from child._walk()

    def merge(self, other):
        """Adopt ``other``: one launch then refreshes both (a coupling layer's final-layer and hidden-stack images change
        together, once per optimizer step).  Non-destructive: ``other`` keeps its jobs and can be re-parented later (its
        previous parent lets go of it), so a rebuilt parent never inherits jobs whose sources are gone."""
        mine = self.root()
        if other is mine or other._root is mine:
            return
        if other.device != mine.device:
            raise ValueError("DevicePack.merge: packs live on different devices")
        if any(node is other for node in mine._walk()) or any(node is mine for node in other._walk()):
            return
        if other._root is not None:
            other._root.children = [c for c in other._root.children if c is not other]
            other._root.root()._invalidate()
        other._root = mine
        mine.children.append(other)
        mine._invalidate()

    def all_jobs(self):
        return [job for node in self._walk() for job in node.jobs]

    def all_sources(self):
        return [t for node in self._walk() for t in node.sources]

    def run(self):
        pack = self.root()
        lib = _hip.load()
        if pack._jobs_dev is None:
            if lib.fc_pack_job_bytes() != ctypes.sizeof(_hip.PackJob):
                raise RuntimeError("fc_pack_job layout mismatch between the header and the ctypes mirror")
            jobs = pack.all_jobs()
            raw = b"".join(bytes(j) for j in jobs)
            pack._jobs_dev = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(pack.device), len(jobs))
        for node in pack._walk():
            for fn in node.prepare:
                fn()
        _call("fc_pack_fragments", lib.fc_pack_fragments, pack.device, _hip.ptr(pack._jobs_dev[0]), pack._jobs_dev[1],
              _hip.stream_ptr(pack.device))

    def refresh(self):
        """``run()`` if any source parameter changed since the last refresh (``cache_key``: versions + cache epoch)."""
        pack = self.root()
        seen, srcs = set(), []
        for t in pack.all_sources():
            if id(t) not in seen:
                seen.add(id(t))
                srcs.append(t)
        key = cache_key(*srcs)
        if pack._key != key:
            pack.run()
            pack._key = key


def device_pack_final_layer(weight, bias, num_bins, tails, cols_chunks):
    """Forward and W^T fragments of the final Linear (hidden <= 64) for every group of <= 32 transformed dims, packed on
    the device.  ``cols_chunks``: [(row slice of the weight, cols tensor)].  Returns ``(pack, chunks)`` with chunks =
    [(w_frag, w_unscale, bias_pad, wt_frag, cols, row slice)]; call ``pack.run()`` whenever the weights changed."""
    k = num_bins
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    pp = -(-p // 4) * 4
    t = pp // 4
    kk = -(-pp // 8)
    dev = weight.device
    pack = DevicePack(dev)
    chunks = []
    for rows, cols in cols_chunks:
        w, b = weight[rows], bias[rows]
        d_t = w.shape[0] // p
        groups = -(-d_t // 4)
        w_frag = torch.empty(groups, 2, t, 2, 64, 8, dtype=torch.float16, device=dev)
        wt_frag = torch.empty(groups, 4, kk, 2, 64, 8, dtype=torch.float16, device=dev)
        w_un = torch.empty(groups, dtype=torch.float32, device=dev)
        wt_un = torch.empty(groups, dtype=torch.float32, device=dev)       # (equal to w_un: the same group maximum)
        bias_pad = torch.empty(groups, 4, pp, dtype=torch.float32, device=dev)
        for g in range(groups):
            pack.add(PACK_FINAL, w, b, w_frag[g], w_un[g:g + 1], bias_pad[g], p=p, pp=pp, nks=2, nt=t, group=g)
            pack.add(PACK_FINAL_T, w, None, wt_frag[g], wt_un[g:g + 1], None, p=p, pp=pp, nks=kk, nt=4, group=g)
        chunks.append((w_frag, w_un, bias_pad, wt_frag, cols, rows))
    return pack, chunks


def device_pack_resnet_hidden_backward(net):
    """``pack_resnet_hidden_backward`` on the device: returns ``(pack, packed)``; ``pack.run()`` refreshes ``packed``."""
    dev = net.initial_layer.weight.device
    k0 = net.initial_layer.in_features
    k0s = 1 if k0 <= 32 else 2
    layers = [net.initial_layer] + [lin for block in net.blocks for lin in block.linear_layers]
    n_layers = len(layers)
    frag0, frag_l = k0s * 4 * 2 * 64 * 8, 2 * 4 * 2 * 64 * 8
    w_frag = torch.empty(frag0 + (n_layers - 1) * frag_l, dtype=torch.float16, device=dev)
    wt_frag = torch.empty((n_layers - 1) * frag_l + 2 * (2 * k0s) * 2 * 64 * 8, dtype=torch.float16, device=dev)
    w_un = torch.empty(n_layers, dtype=torch.float32, device=dev)
    wt_un = torch.empty(n_layers, dtype=torch.float32, device=dev)
    bias_acc = torch.empty(n_layers, 64, dtype=torch.float32, device=dev)
    pack = DevicePack(dev)
    off = 0
    for i, lin in enumerate(layers):
        size = frag0 if i == 0 else frag_l
        pack.add(PACK_HIDDEN, lin.weight, lin.bias, w_frag[off:off + size], w_un[i:i + 1], bias_acc[i],
                 nks=k0s if i == 0 else 2, nt=4)
        off += size
    for i, lin in enumerate(layers[1:]):
        pack.add(PACK_HIDDEN_T, lin.weight, None, wt_frag[i * frag_l:(i + 1) * frag_l], wt_un[i + 1:i + 2], None, nks=2, nt=4)
    pack.add(PACK_HIDDEN_T0, layers[0].weight, None, wt_frag[(n_layers - 1) * frag_l:], wt_un[0:1], None, nks=2, nt=2 * k0s)
    return pack, (w_frag, wt_frag, w_un, bias_acc, k0s)


def device_pack_resnet_hidden_forward(net):
    """The LDS weight image of ``fc_resnet_hidden_packed`` for a ResidualNet with hidden <= 64, <= 4 blocks, no context,
    made on the device: returns ``(pack, (w_frag, w_unscale [L], bias_acc [L, 64]))``; ``pack.run()`` refreshes it."""
    dev = net.initial_layer.weight.device
    k0s = 1 if net.initial_layer.in_features <= 32 else 2
    layers = [net.initial_layer] + [lin for block in net.blocks for lin in block.linear_layers]
    frag0, frag_l = k0s * 4 * 2 * 64 * 8, 2 * 4 * 2 * 64 * 8
    w_frag = torch.empty(frag0 + (len(layers) - 1) * frag_l, dtype=torch.float16, device=dev)
    w_un = torch.empty(len(layers), dtype=torch.float32, device=dev)
    bias_acc = torch.empty(len(layers), 64, dtype=torch.float32, device=dev)
    pack = DevicePack(dev)
    off = 0
    for i, lin in enumerate(layers):
        size = frag0 if i == 0 else frag_l
        pack.add(PACK_HIDDEN, lin.weight, lin.bias, w_frag[off:off + size], w_un[i:i + 1], bias_acc[i],
                 nks=k0s if i == 0 else 2, nt=4)
        off += size
    return pack, (w_frag, w_un, bias_acc)


def device_pack_affine_coupling(net, d_t, additive):
    """The LDS image of ``fc_affine_coupling_resnet``: the hidden layers of ``net`` (``device_pack_resnet_hidden_forward``)
    plus its final Linear as one more 64 x 64 layer with rows 0..31 = shift rows, rows 32..63 = scale rows.  The re-ordered
    copy of the final layer lives in a staging buffer refreshed before every pack launch.  Returns ``(pack, packed)``."""
    dev = net.initial_layer.weight.device
    k0s = 1 if net.initial_layer.in_features <= 32 else 2
    hidden_layers = [net.initial_layer] + [lin for block in net.blocks for lin in block.linear_layers]
    n_layers = len(hidden_layers) + 1
    frag0, frag_l = k0s * 4 * 2 * 64 * 8, 2 * 4 * 2 * 64 * 8
    w_frag = torch.empty(frag0 + (n_layers - 1) * frag_l, dtype=torch.float16, device=dev)
    w_un = torch.empty(n_layers, dtype=torch.float32, device=dev)
    bias_acc = torch.empty(n_layers, 64, dtype=torch.float32, device=dev)
    lin = net.final_layer
    stage_w = torch.zeros(64, 64, dtype=torch.float32, device=dev)
    stage_b = torch.zeros(64, dtype=torch.float32, device=dev)
    hf = lin.in_features

    def stage():
        with torch.no_grad():
            stage_w[:d_t, :hf].copy_(lin.weight[:d_t])
            stage_b[:d_t].copy_(lin.bias[:d_t])
            if not additive:
                stage_w[32:32 + d_t, :hf].copy_(lin.weight[d_t:2 * d_t])
                stage_b[32:32 + d_t].copy_(lin.bias[d_t:2 * d_t])

    pack = DevicePack(dev)
    off = 0
    for i, layer in enumerate(hidden_layers):
        size = frag0 if i == 0 else frag_l
        pack.add(PACK_HIDDEN, layer.weight, layer.bias, w_frag[off:off + size], w_un[i:i + 1], bias_acc[i],
                 nks=k0s if i == 0 else 2, nt=4)
        off += size
    pack.add(PACK_HIDDEN, stage_w, stage_b, w_frag[off:off + frag_l], w_un[n_layers - 1:n_layers], bias_acc[n_layers - 1],
             nks=2, nt=4, track=False)
    pack.sources += [lin.weight, lin.bias]        # (the staging buffers' versions move only when these do)
    pack.prepare.append(stage)
    return pack, (w_frag, w_un, bias_acc)


def device_pack_made_affine(made, features):
    """The LDS image of ``fc_affine_coupling_resnet`` for the DENSITY direction of a masked-autoregressive affine layer
    (autoregressive.py:97-129): every layer of the MADE with its mask multiplied in (staging copies, refreshed before each
    pack launch), the final masked Linear re-ordered from the interleaved [D, (u, shift)] rows to rows 0..31 = shift of dims
    0..31, rows 32..63 = u.  Returns ``(pack, packed)`` like ``device_pack_affine_coupling``."""
    dev = made.initial_layer.weight.device
    hidden_layers = [made.initial_layer] + [lin for block in made.blocks for lin in block.linear_layers]
    n_layers = len(hidden_layers) + 1
    k0s = 1
    frag0, frag_l = k0s * 4 * 2 * 64 * 8, 2 * 4 * 2 * 64 * 8
    w_frag = torch.empty(frag0 + (n_layers - 1) * frag_l, dtype=torch.float16, device=dev)
    w_un = torch.empty(n_layers, dtype=torch.float32, device=dev)
    bias_acc = torch.empty(n_layers, 64, dtype=torch.float32, device=dev)
    final = made.final_layer
    stage_w = [torch.zeros(64, 32 if i == 0 else 64, dtype=torch.float32, device=dev) for i in range(n_layers)]
    stage_b = [torch.zeros(64, dtype=torch.float32, device=dev) for _ in range(n_layers)]

    def stage():
        with torch.no_grad():
            for i, lin in enumerate(hidden_layers):
                stage_w[i][:lin.out_features, :lin.in_features].copy_(lin.weight * lin.mask)
                stage_b[i][:lin.out_features].copy_(lin.bias)
            w = final.weight * final.mask
            hf = final.in_features
            stage_w[-1][:features, :hf].copy_(w[1::2])            # shift rows (parameter 1 of every dim)
            stage_w[-1][32:32 + features, :hf].copy_(w[0::2])     # unconstrained-scale rows (parameter 0)
            stage_b[-1][:features].copy_(final.bias[1::2])
            stage_b[-1][32:32 + features].copy_(final.bias[0::2])

    pack = DevicePack(dev)
    off = 0
    for i in range(n_layers):
        size = frag0 if i == 0 else frag_l
        pack.add(PACK_HIDDEN, stage_w[i], stage_b[i], w_frag[off:off + size], w_un[i:i + 1], bias_acc[i],
                 nks=k0s if i == 0 else 2, nt=4, track=False)
        off += size
    for lin in hidden_layers + [final]:
        pack.sources += [lin.weight, lin.bias]
    pack.prepare.append(stage)
    return pack, (w_frag, w_un, bias_acc)


MADE_AFFINE, MADE_RQ = 0, 1


def _made_pass_prefix(made, features, per_dim, hw):
    """Which hidden units pass d of the inverse reads, from the masks alone: the units the rows of dim d reach backwards
    through the hidden layers (and the residual identities).  Returns ``(order, need)``: ``order[rank]`` = hidden unit, sorted
    by the first pass that reads it (zero-padded units last), and ``need[d]`` = how many leading units of that order pass d
    reads -- for the reference's degrees (made.py:13-24: unit j has degree j % (D - 1) + 1) the units of degree <= d."""
    hidden = [lin for block in made.blocks for lin in block.linear_layers]
    step = torch.eye(hw, dtype=torch.bool)
    for lin in hidden:
        step |= _pad_to((lin.mask != 0).cpu(), (hw, hw))                     # [unit, the units it reads]
    reach = _pad_to((made.final_layer.mask != 0).cpu().reshape(features, per_dim, -1).any(dim=1), (features, hw))
    while True:
        wider = reach | ((reach.float() @ step.float()) > 0)
        if bool((wider == reach).all()):
            break
        reach = wider
    dims = torch.arange(features).reshape(-1, 1).expand(features, hw)
    first = torch.where(reach, dims, torch.full_like(dims, features)).amin(dim=0)      # [hw]; `features` = never read
    order = torch.argsort(first, stable=True)
    need = (first.reshape(1, -1) <= torch.arange(features).reshape(-1, 1)).sum(dim=1).to(torch.int32)
    return order, need


def pack_made_inverse(made, features, per_dim):
    """Everything ``fc_made_inverse`` needs of a residual-block MADE (hidden <= 64, <= 3 blocks, <= 64 inputs): the hidden
    stack's image on MASKED weights (rows in the accumulator order of the hidden-layer kernels, one power-of-two scale per
    layer) and the final layer as per-dim row tiles (``per_dim`` parameter rows of every dim padded to whole 16-row tiles,
    one scale per dim).  The hidden units are renumbered in the order the passes first read them (``_made_pass_prefix``:
    the same renumbering in every layer, so the residual sums stay unit-for-unit), which lets pass d compute only the
    leading 16-unit tiles / 32-unit k-steps that hold its ``units_needed[d]`` units.  Returns ``(hidden_frag,
    hidden_unscale [L], hidden_bias [L, 64], final_frag, final_unscale [D], final_bias [D, 16 PT], units_needed [D])``."""
    hw = 64
    dev = made.initial_layer.weight.device
    perm = _hb_perm().to(dev)
    k0s = 1 if features <= 32 else 2
    layers = [made.initial_layer] + [lin for block in made.blocks for lin in block.linear_layers]
    wf, uns, biases = [], [], []
    with torch.no_grad():
        order, need = _made_pass_prefix(made, features, per_dim, hw)
        order = order.to(dev)
        # rank r sits where the accumulator layout keeps feature perm[r]: ranks 0..15 fill product tile 0, 16..31 tile 1
        # (both in k-step 0), 32..47 tile 2, 48..63 tile 3 (k-step 1)
        slot = perm
        for i, lin in enumerate(layers):
            w = _pad_to((lin.weight * lin.mask).detach().float(), (hw, 32 * k0s if i == 0 else hw))
            moved = torch.zeros_like(w)
            if i == 0:
                moved[slot] = w[order]
            else:
                moved[slot.reshape(-1, 1), slot.reshape(1, -1)] = w[order.reshape(-1, 1), order.reshape(1, -1)]
            sc, un = _pow2_scale(moved.abs().amax().reshape(1))
            uns.append(un)
            wf.append(_a_fragments((moved * sc)[perm]).permute(1, 0, 2, 3, 4).reshape(-1))        # [ks][t][piece][lane][8]
            b = torch.zeros(hw, device=dev)
            b[slot] = _pad_to(lin.bias.detach().float(), (hw,))[order]
            biases.append(b[perm].reshape(4, 4, 4).permute(1, 0, 2).reshape(-1))
        final = made.final_layer
        pt = -(-per_dim // 16)
        w = _pad_to((final.weight * final.mask).detach().float().reshape(features, per_dim, -1), (features, 16 * pt, hw))
        moved = torch.zeros_like(w)
        moved[:, :, slot] = w[:, :, order]
        w = moved
        sc, un = _pow2_scale(w.abs().amax(dim=(1, 2)))
        frag = _a_fragments((w * sc.reshape(-1, 1, 1)).reshape(features * 16 * pt, hw))           # [D PT, ks, piece, lane, 8]
        final_frag = frag.reshape(features, pt, 2, 2, 64, 8).permute(0, 2, 1, 3, 4, 5).contiguous()   # [D][ks][t][piece][lane][8]
        final_bias = _pad_to(final.bias.detach().float().reshape(features, per_dim), (features, 16 * pt)).contiguous()
    return (torch.cat(wf).contiguous(), torch.cat(uns).float().contiguous(), torch.stack(biases).contiguous(),
            final_frag, un.float().contiguous(), final_bias, need.to(dev).contiguous())


def made_inverse(inputs, packed, num_blocks, per_dim, kind, rq=None, logabsdet_accum=None):
    """The D passes of an autoregressive inverse in ONE kernel (``fc_made_inverse``): ``inputs`` [N, D <= 64] (rows a
    multiple of 16), ``packed`` from ``pack_made_inverse``; ``kind`` ``MADE_AFFINE`` or ``MADE_RQ`` (``rq``: keyword
    arguments of the spline as for ``rq_spline``).  Returns ``(outputs, logabsdet)``."""
    lib = _hip.load()
    z = _prep_2d(inputs)
    _hip.require_no_grad(inputs)
    n, d = z.shape
    if n % HIDDEN_ROWS != 0 or d > 64:
        raise ValueError("fc_made_inverse: rows must be a multiple of %d, D <= 64" % HIDDEN_ROWS)
    cfg = None
    if kind == MADE_RQ:
        rq = dict(rq)
        cfg = _rq_config(rq.pop("num_bins"), rq.pop("tails"), rq.pop("tail_bound", 1.0),
                         (rq.pop("left", 0.0), rq.pop("right", 1.0), rq.pop("bottom", 0.0), rq.pop("top", 1.0)),
                         rq.pop("min_bin_width", DEFAULT_MIN_BIN_WIDTH), rq.pop("min_bin_height", DEFAULT_MIN_BIN_HEIGHT),
                         rq.pop("min_derivative", DEFAULT_MIN_DERIVATIVE), rq.pop("enable_identity_init", False),
                         rq.pop("wh_divisor", 1.0), True)
        if rq:
            raise TypeError("made_inverse: unknown spline arguments %s" % sorted(rq))
    y = torch.empty_like(z)
    if logabsdet_accum is not None:
        lad = logabsdet_accum
        if lad.dtype != torch.float32 or lad.shape != (n,) or not lad.is_contiguous() or lad.device != z.device:
            raise ValueError("logabsdet_accum must be a contiguous float32 [N] tensor on the inputs' device")
        if cfg is None:
            cfg = _hip.RQConfig()          # affine form: only the flags are read
        cfg.flags = 1  # FC_RQ_ACCUMULATE_LOGABSDET
    else:
        lad = torch.empty(n, dtype=torch.float32, device=z.device)
    err = _err_word(z.device, True)
    hf, hu, hb, ff, fu, fb, need = packed
    if need.dtype != torch.int32 or need.numel() != d:
        raise ValueError("made_inverse: units_needed must hold one int32 per dim")
    _call("fc_made_inverse", lib.fc_made_inverse, z.device, _hip.ptr(z), _hip.ptr(y), _hip.ptr(lad), _hip.ptr(hf),
          _hip.ptr(hu), _hip.ptr(hb), _hip.ptr(ff), _hip.ptr(fu), _hip.ptr(fb), _hip.ptr(need), _hip.ptr(err), n, d, num_blocks, per_dim,
          kind, cfg, _hip.stream_ptr(z.device))
    _finish(True)
    return y, lad


def affine_tail_fits(in_features, num_blocks, d):
    """LDS budget of ``fc_affine_coupling_resnet``: the weight image (initial layer, 2 per block, the final Linear) + one
    [16, D | 1] float tile per wave next to it, 160 KB per CU; D <= 128, <= 3 blocks."""
    k0s = 1 if in_features <= 32 else 2
    layers = 2 + 2 * num_blocks
    image = (k0s * 8 + (2 * num_blocks + 1) * 16) * 1024 + layers * 64 * 4 + 64 + 32 * k0s * 4 + 512
    return d <= 128 and num_blocks <= 3 and image + 16 + 8 * 16 * (d | 1) * 4 <= 160 * 1024


def affine_tail_activation(code):
    """Scale activations ``fc_affine_coupling_resnet`` evaluates itself."""
    return code in (AFFINE_SIGMOID_PLUS2, AFFINE_SOFTPLUS_CLAMP3, AFFINE_ADDITIVE, AFFINE_MAF_SOFTPLUS)


def affine_coupling_resnet(inputs, id_cols, tr_cols, packed, in_features, num_blocks, activation, inverse=False,
                           logabsdet_accum=None):
    """One affine / additive coupling layer with a ResidualNet(hidden <= 64, <= 3 ReLU blocks) conditioner in ONE kernel
    (``fc_affine_coupling_resnet``); rows a multiple of 16.  Returns ``(outputs, logabsdet)``; with ``logabsdet_accum`` the
    layer's logabsdet is added onto that tensor, which is returned."""
    lib = _hip.load()
    x = _prep_2d(inputs, align16=True)        # the kernel moves whole 16-row chunks with 16-byte loads
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % HIDDEN_ROWS != 0 or not affine_tail_activation(activation):
        raise ValueError("fc_affine_coupling_resnet: unsupported rows / activation")
    w_frag, w_un, bias_acc = packed
    ids, cols = _as_cols(id_cols, x.device), _as_cols(tr_cols, x.device)
    y = torch.empty_like(x)
    if logabsdet_accum is not None:
        lad = logabsdet_accum
        if lad.dtype != torch.float32 or lad.shape != (n,) or not lad.is_contiguous() or lad.device != x.device:
            raise ValueError("logabsdet_accum must be a contiguous float32 [N] tensor on the inputs' device")
    else:
        lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_affine_coupling_resnet", lib.fc_affine_coupling_resnet, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(ids),
          _hip.ptr(cols), _hip.ptr(w_frag), _hip.ptr(w_un), _hip.ptr(bias_acc), _hip.ptr(lad), n, d, in_features,
          cols.numel(), 64, num_blocks, int(activation), 1 if inverse else 0, 0 if logabsdet_accum is None else 1,
          _hip.stream_ptr(x.device))
    return y, lad


def resnet_hidden_packed(inputs, id_cols, packed, in_features, num_blocks, activation=(ACT_RELU, 0.0)):
    """``resnet_hidden`` (no context) from a ready-made weight image (``device_pack_resnet_hidden_forward``)."""
    lib = _hip.load()
    x = _prep_2d(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % HIDDEN_ROWS != 0:
        raise ValueError("fc_resnet_hidden needs a multiple of %d rows" % HIDDEN_ROWS)
    w_frag, w_un, bias_acc = packed
    k0s = 1 if in_features <= 32 else 2
    if w_frag.numel() != (k0s + 4 * num_blocks) * 4096 or w_un.numel() != 1 + 2 * num_blocks:
        raise ValueError("weight image does not match in_features = %d, num_blocks = %d" % (in_features, num_blocks))
    ids = _as_cols(id_cols, x.device)
    h = torch.empty(n, 64, dtype=torch.float32, device=x.device)
    # (timed and reported under the name of the kernel it launches: fc_resnet_hidden with a ready-made image)
    _call("fc_resnet_hidden", lib.fc_resnet_hidden_packed, x.device, _hip.ptr(x), _hip.ptr(h), _hip.ptr(ids),
          _hip.ptr(w_frag), _hip.ptr(w_un), _hip.ptr(bias_acc), n, d, in_features, 64, num_blocks, int(activation[0]),
          float(activation[1]), _hip.stream_ptr(x.device))
    return h


def _hb_perm():
    """Feature held by accumulator tile t, row rho of the hidden-layer kernels: 32 (t >> 1) + 8 g + 4 (t & 1) + r with
    g = rho >> 2, r = rho & 3 (the order in which the C layout of one layer is the B operand of the next)."""
    return torch.tensor([32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3) for t in range(4) for rho in range(16)])


def pack_resnet_hidden_backward(net):
    """Everything ``fc_resnet_hidden_backward`` needs of a ResidualNet with hidden <= 64, <= 2 ReLU blocks, no context:
    forward fragments (rows in accumulator order), fragments of the transposed weights for the W^T products, one
    power-of-two scale per layer shared by both, biases in accumulator order.  Returns
    ``(w_frag, wt_frag, w_unscale [L], bias_acc [L, 64], k0s)``."""
    hw = 64
    perm = _hb_perm().to(net.initial_layer.weight.device)
    k0 = net.initial_layer.in_features
    k0s = 1 if k0 <= 32 else 2
    layers = [net.initial_layer] + [lin for block in net.blocks for lin in block.linear_layers]
    wf, wt, uns, biases = [], [], [], []
    scaled = []
    for i, lin in enumerate(layers):
        w = _pad_to(lin.weight.detach().float(), (hw, 32 * k0s if i == 0 else hw))
        sc, un = _pow2_scale(w.abs().amax().reshape(1))
        scaled.append(w * sc)
        uns.append(un)
        wf.append(_a_fragments(scaled[-1][perm]).permute(1, 0, 2, 3, 4).reshape(-1))         # [ks][t][piece][lane][8]
        biases.append(_pad_to(lin.bias.detach().float(), (hw,))[perm].reshape(4, 4, 4).permute(1, 0, 2).reshape(-1))
    for w in scaled[1:]:
        wt.append(_a_fragments(w.t().contiguous()[perm]).permute(1, 0, 2, 3, 4).reshape(-1))
    wt.append(_a_fragments(scaled[0].t().contiguous()).permute(1, 0, 2, 3, 4).reshape(-1))     # W0^T: rows natural
    return (torch.cat(wf).contiguous(), torch.cat(wt).contiguous(), torch.cat(uns).float().contiguous(),
            torch.stack(biases).contiguous(), k0s)


def resnet_hidden_backward(inputs, grad_hidden, id_cols, packed, in_features, num_blocks, grad_inputs_accum=None):
    """Backward of ``resnet_hidden`` (hidden 64, <= 2 ReLU blocks, no context; rows a multiple of 128): returns
    ``(grad_x_id [N, in_features], grad_w0 [64, in_features], grad_wb [2 blocks, 64, 64], grad_b [L, 64])`` with the
    activations recomputed from ``inputs``.  With ``grad_inputs_accum`` [N, D] the gradient wrt the identity columns is
    added into it in place (``fc_resnet_hidden_backward_accum``) and the first result is None."""
    lib = _hip.load()
    x = _prep_2d(inputs.detach())
    gh = _aligned16(_hip.dev_f32(grad_hidden, "grad_hidden"))
    n, d = x.shape
    if n % HIDDEN_BWD_ROWS != 0 or gh.shape != (n, 64) or not 0 <= num_blocks <= 2:
        raise ValueError("fc_resnet_hidden_backward: unsupported shapes")
    w_frag, wt_frag, w_un, bias_acc, k0s = packed
    ids = _as_cols(id_cols, x.device)
    layers = 1 + 2 * num_blocks
    if grad_inputs_accum is not None and (grad_inputs_accum.shape != x.shape or not grad_inputs_accum.is_contiguous()
                                          or grad_inputs_accum.dtype != torch.float32):
        raise ValueError("grad_inputs_accum must be a contiguous float32 [N, D] tensor")
    gxid = None if grad_inputs_accum is not None else torch.empty(n, 32 * k0s, dtype=torch.float32, device=x.device)
    nb2 = max(1, 2 * num_blocks)
    acc = torch.zeros(64 * 32 * k0s + nb2 * 4096 + layers * 64, dtype=torch.float32, device=x.device)   # one memset
    gw0 = acc[:64 * 32 * k0s].view(64, 32 * k0s)
    gwb = acc[64 * 32 * k0s:64 * 32 * k0s + nb2 * 4096].view(nb2, 64, 64)
    gb = acc[64 * 32 * k0s + nb2 * 4096:].view(layers, 64)
    fn = lib.fc_resnet_hidden_backward if gxid is not None else lib.fc_resnet_hidden_backward_accum
    _call("fc_resnet_hidden_backward", fn, x.device, _hip.ptr(x), _hip.ptr(gh), _hip.ptr(ids),
          _hip.ptr(w_frag), _hip.ptr(wt_frag), _hip.ptr(w_un), _hip.ptr(bias_acc),
          _hip.ptr(gxid if gxid is not None else grad_inputs_accum), _hip.ptr(gw0),
          _hip.ptr(gwb), _hip.ptr(gb), n, d, in_features, 64, num_blocks, ACT_RELU, _hip.stream_ptr(x.device))
    return (None if gxid is None else gxid[:, :in_features]), gw0[:, :in_features], gwb, gb


def resnet_hidden_wide(inputs, id_cols, packed, in_features, num_blocks, width, activation=(ACT_RELU, 0.0)):
    """Hidden layers of a wide conditioner on the rows of ``inputs`` (multiple of 64 rows) -> h [N, width]."""
    lib = _hip.load()
    x = _prep_2d(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % WIDE_ROWS != 0 or width not in (128, 256):
        raise ValueError("fc_resnet_hidden_wide needs a multiple of %d rows and a width of 128 or 256" % WIDE_ROWS)
    w_frag, w_un, bias = packed
    ids = _as_cols(id_cols, x.device)
    h = torch.empty(n, width, dtype=torch.float32, device=x.device)
    _call("fc_resnet_hidden_wide", lib.fc_resnet_hidden_wide, x.device, _hip.ptr(x), _hip.ptr(h), _hip.ptr(ids),
          _hip.ptr(w_frag), _hip.ptr(w_un), _hip.ptr(bias), n, d, in_features, width, num_blocks, int(activation[0]),
          float(activation[1]), _hip.stream_ptr(x.device))
    return h


def resnet_hidden(inputs, id_cols, packed, in_features, num_blocks, context=None, activation=(ACT_RELU, 0.0),
                  context_mode=CONTEXT_GLU):
    """Hidden layers of the conditioner on the rows of ``inputs`` (multiple of 16 rows) -> h [N, 64].
    ``activation``: ``activation_code`` of the blocks' activation.  ``context_mode``: ``CONTEXT_GLU`` (ResidualNet:
    concatenated into the initial layer, GLU gate per block) or ``CONTEXT_ADDITIVE`` (MADE: added after the initial
    layer through the activation and inside every block; ``packed`` then carries ``blocks + 1`` context layers).
    ``in_features`` = number of identity columns read from ``inputs``; ``context`` [N, C] (C <= 32,
    in_features + C <= 64) enters the initial layer after them and gates every block (``packed`` then carries
    the context layers)."""
    lib = _hip.load()
    x = _prep_2d(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % HIDDEN_ROWS != 0:
        raise ValueError("fc_resnet_hidden needs a multiple of %d rows" % HIDDEN_ROWS)
    w0, b0, wb, bb = packed[:4]
    ids = _as_cols(id_cols, x.device)
    h = torch.empty(n, 64, dtype=torch.float32, device=x.device)
    if context is None:
        _call("fc_resnet_hidden", lib.fc_resnet_hidden, x.device, _hip.ptr(x), _hip.ptr(h), _hip.ptr(ids),
              _hip.ptr(w0), _hip.ptr(b0), _hip.ptr(wb), _hip.ptr(bb), n, d, in_features, 64, num_blocks,
              int(activation[0]), float(activation[1]), _hip.stream_ptr(x.device))
        return h
    wc, bc = packed[4:6]
    c = _prep_2d(context)
    _hip.require_no_grad(context)
    if c.shape[0] != n or w0.shape[1] != in_features + (c.shape[1] if context_mode == CONTEXT_GLU else 0):
        raise ValueError("context rows / width do not match the inputs / the initial layer")
    _call("fc_resnet_hidden_context", lib.fc_resnet_hidden_context, x.device, _hip.ptr(x), _hip.ptr(c), _hip.ptr(h),
          _hip.ptr(ids), _hip.ptr(w0), _hip.ptr(b0), _hip.ptr(wb), _hip.ptr(bb), _hip.ptr(wc), _hip.ptr(bc), n, d,
          in_features, c.shape[1], 64, num_blocks, int(context_mode), int(activation[0]), float(activation[1]),
          _hip.stream_ptr(x.device))
    return h


def pack_final_layer(weight, bias, num_bins=FUSED_BINS):
    """[d_t*23, H <= 64] weight / [d_t*23] bias of the conditioner's final Linear -> (w_pad [dp*24, 64], bias_pad
    [dp*24]): one zero row / entry appended per dim so that a dim is 24 = 6 x 4 accumulator registers, zero
    dims appended up to dp = ceil(d_t / 4) * 4 (a wave owns 4 dims), zero columns up to the kernel's 64 hidden
    units (they meet the zero activations of a zero-padded hidden stack)."""
    p = 3 * num_bins - 1
    d_t = weight.shape[0] // p
    dp = -(-d_t // 4) * 4
    hidden = weight.shape[1]
    w = weight.detach().reshape(d_t, p, hidden)
    wpad = w.new_zeros(dp, p + 1, FUSED_HIDDEN)
    wpad[:d_t, :p, :hidden] = w
    bpad = bias.new_zeros(dp, p + 1)
    bpad[:d_t, :p] = bias.detach().reshape(d_t, p)
    return wpad.reshape(dp * (p + 1), FUSED_HIDDEN).contiguous(), bpad.reshape(-1).contiguous()


def rq_spline_fused_linear(inputs, hidden, w_pad, bias_pad, cols, *, num_bins, tail_bound,
                           min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                           min_derivative=DEFAULT_MIN_DERIVATIVE, wh_divisor=1.0, inverse=False,
                           logabsdet_accum=None, enable_identity_init=False):
    """RQ-spline coupling bijector with the conditioner's final Linear fused in (rows must be a multiple
    of 32).  ``hidden``: [N, 64] input of that Linear.  Returns ``(outputs [N, D], logabsdet [N])``; with
    ``logabsdet_accum`` (f32 [N], contiguous) the kernel adds the layer's logabsdet onto it in place and that
    tensor is returned.  ``enable_identity_init``: the autoregressive form's softplus beta
    (autoregressive.py:612)."""
    lib = _hip.load()
    x = _prep_2d(inputs, align16=True)
    h = _aligned16(_hip.dev_f32(hidden, "hidden"))
    _hip.require_no_grad(inputs, hidden)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = cols.numel()
    raw = w_pad.shape[0] == d_t * 23          # the nn.Linear tensors as they are (FC_RQ_RAW_WEIGHTS)
    if (n % FUSED_ROWS != 0 or h.shape != (n, FUSED_HIDDEN) or not 1 <= d_t <= FUSED_DT
            or (not raw and w_pad.shape[0] != -(-d_t // 4) * 4 * 24) or w_pad.shape[1] != FUSED_HIDDEN
            or bias_pad.numel() != w_pad.shape[0]):
        raise ValueError("fused RQ layer: unsupported shapes %s / %s" % (tuple(x.shape), tuple(h.shape)))
    w_pad = _aligned16(_hip.dev_f32(w_pad.detach(), "weight"))
    bias_pad = _hip.dev_f32(bias_pad.detach(), "bias")
    cfg = _hip.RQConfig()
    cfg.num_bins, cfg.tails, cfg.inverse = num_bins, 1, 1 if inverse else 0
    cfg.left, cfg.right, cfg.bottom, cfg.top = -tail_bound, tail_bound, -tail_bound, tail_bound
    cfg.min_bin_width, cfg.min_bin_height, cfg.min_derivative = min_bin_width, min_bin_height, min_derivative
    cfg.wh_divisor = wh_divisor
    cfg.softplus_beta = (math.log(2) / (1 - min_derivative)) if enable_identity_init else 1.0
    cfg.tail_constant = float(np.log(np.exp(1 - min_derivative) - 1))
    y = torch.empty_like(x)
    if logabsdet_accum is not None:
        lad = logabsdet_accum
        if lad.dtype != torch.float32 or lad.shape != (n,) or not lad.is_contiguous() or lad.device != x.device:
            raise ValueError("logabsdet_accum must be a contiguous float32 [N] tensor on the inputs' device")
        cfg.flags = 1  # FC_RQ_ACCUMULATE_LOGABSDET
    else:
        lad = torch.empty(n, dtype=torch.float32, device=x.device)
    if raw:
        cfg.flags |= 4  # FC_RQ_RAW_WEIGHTS
    err = _err_word(x.device, True)
    _call("fc_rq_spline_fused_linear", lib.fc_rq_spline_fused_linear, x.device, _hip.ptr(x), _hip.ptr(y),
          _hip.ptr(h), _hip.ptr(w_pad), _hip.ptr(bias_pad), _hip.ptr(cols), _hip.ptr(lad), _hip.ptr(err), n, d,
          d_t, FUSED_HIDDEN, cfg, _hip.stream_ptr(x.device))
    _finish(True)
    return y, lad


# ---- general fused final layer (any K = 4..16, tails or box, hidden <= 256) -------------------------------------

GENERAL_BINS = range(4, 17)
GENERAL_HIDDEN = (64, 128, 256)


def general_hidden_width(hidden):
    """Width (64 / 128 / 256) a hidden activation is zero-padded to for ``fc_rq_spline_fused_general``; None if wider."""
    for w in GENERAL_HIDDEN:
        if hidden <= w:
            return w
    return None


def fused_general_supported(n, d, d_t, hidden, num_bins, tails):
    """Shapes of the general fused final-layer + RQ-spline kernel: K = 4..16, linear tails or none, hidden <= 256,
    <= 32 transformed dims per launch, D <= 128, >= 32 rows."""
    return (general_hidden_width(hidden) is not None and 1 <= d_t <= FUSED_DT and num_bins in GENERAL_BINS
            and tails in (None, "linear") and d <= 128 and n >= FUSED_ROWS)


def pack_final_layer_general(weight, bias, num_bins, tails, hidden_pad):
    """The conditioner's final Linear ([d_t * P, H] weight, [d_t * P] bias, P = 3K -/+ 1) as
    ``fc_rq_spline_fused_general`` streams it: matrix-core A fragments of the power-of-two scaled weight split into
    two f16 pieces (fc_split.h), one scale per group of 4 dims --

        w_frag   f16 [groups, H/32, T, 2 (hi, lo), 64 lanes, 8]    T = ceil(P / 4) tiles of 16 rows: (dim g, param 4t + r)
        w_unscale f32 [groups]                                      2^-S of the group
        bias_pad f32 [groups, 4, 4 T]

    (zero rows / dims / columns pad P to 4 T, d_t to 4 * groups, H to ``hidden_pad``)."""
    k = num_bins
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    pp = -(-p // 4) * 4
    t = pp // 4
    d_t = weight.shape[0] // p
    groups = -(-d_t // 4)
    hidden = weight.shape[1]
    ks = hidden_pad // 32
    w = weight.new_zeros(groups * 4, pp, hidden_pad, dtype=torch.float32)
    w[:d_t, :p, :hidden] = weight.detach().reshape(d_t, p, hidden)
    b = bias.new_zeros(groups * 4, pp, dtype=torch.float32)
    b[:d_t, :p] = bias.detach().reshape(d_t, p)
    scale, unscale = _pow2_scale(w.reshape(groups, -1).abs().amax(dim=1))
    ws = w.reshape(groups, 4, pp, hidden_pad) * scale.reshape(groups, 1, 1, 1)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)

    def frag(piece):
        # [G, dim 4, T, r 4, KS, gk 4, j 8] -> [G, KS, T, gk, dim, r, j] -> [G, KS, T, 64 lanes, 8]
        v = piece.reshape(groups, 4, t, 4, ks, 4, 8).permute(0, 4, 2, 5, 1, 3, 6)
        return v.reshape(groups, ks, t, 64, 8)

    w_frag = torch.stack((frag(hi), frag(lo)), dim=3).contiguous()
    return w_frag, unscale.float().contiguous(), b.reshape(groups, 4, pp).contiguous()


def pack_final_layer_transposed(weight, num_bins, tails):
    """W^T fragments of the final Linear for the backward product gh = W^T G (``fc_rq_fused_linear_backward`` role 0):
    f16 [groups, 4 hidden tiles, KK, 2 (hi, lo), 64 lanes, 8], KK = ceil(4T / 8); lane l of fragment (group, ht, kk)
    holds 2^S W[dim 4 group + (l >> 4)][param 8 kk + j][hidden 16 ht + (l & 15)] -- the k order in which a lane of the
    kernel holds its own parameter gradients.  Same per-group scale as ``pack_final_layer_general``.  hidden <= 64."""
    k = num_bins
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    pp = -(-p // 4) * 4
    pp8 = -(-pp // 8) * 8
    kk = pp8 // 8
    d_t = weight.shape[0] // p
    groups = -(-d_t // 4)
    hidden = weight.shape[1]
    w = weight.new_zeros(groups * 4, pp8, 64, dtype=torch.float32)
    w[:d_t, :p, :hidden] = weight.detach().reshape(d_t, p, hidden)
    sc, _ = _pow2_scale(w.reshape(groups, -1).abs().amax(dim=1))
    ws = w.reshape(groups, 4, pp8, 64) * sc.reshape(groups, 1, 1, 1)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)

    def frag(piece):
        # [G, gk = dim 4, KK, j 8, HT 4, rho 16] -> [G, HT, KK, gk, rho, j] -> [G, HT, KK, 64 lanes, 8]
        v = piece.reshape(groups, 4, kk, 8, 4, 16).permute(0, 4, 2, 1, 5, 3)
        return v.reshape(groups, 4, kk, 64, 8)

    return torch.stack((frag(hi), frag(lo)), dim=3).contiguous()


def rq_fused_linear_backward(inputs, hidden, grad_outputs, grad_logabsdet, packed, packed_t, cols, *, num_bins, tails,
                             tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
                             min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                             min_derivative=DEFAULT_MIN_DERIVATIVE, wh_divisor=1.0, enable_identity_init=False):
    """Gradients of ``rq_spline_fused_general(inputs, hidden, *packed, cols, ...)`` (forward direction, hidden width
    64, rows a multiple of 32): returns ``(grad_inputs [N, D], grad_hidden [N, 64], grad_weight [d_t * P, 64],
    grad_bias [d_t * P])`` for the <= 32 dims of ``cols``.  One launch of ``fc_rq_fused_linear_backward`` (one wave per
    SIMD, ``csrc/fc_rq_fused_backward512.h``): gx / gh deterministic, grad_weight / grad_bias summed with float atomics."""
    lib = _hip.load()
    x = _prep_2d(inputs.detach(), align16=True)
    h = _aligned16(_hip.dev_f32(hidden.detach(), "hidden"))
    gy = _aligned16(_hip.dev_f32(grad_outputs, "grad_outputs"))
    gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = cols.numel()
    w_frag, w_un, bias_pad = packed
    if n % FUSED_ROWS != 0 or h.shape != (n, 64) or not 1 <= d_t <= FUSED_DT:
        raise ValueError("fused RQ layer backward: unsupported shapes %s / %s" % (tuple(x.shape), tuple(h.shape)))
    cfg = _rq_config(num_bins, tails, tail_bound, (left, right, bottom, top), min_bin_width, min_bin_height,
                     min_derivative, enable_identity_init, wh_divisor, False)
    p = 3 * num_bins - 1 if tails == "linear" else 3 * num_bins + 1
    pp = bias_pad.shape[-1]
    groups = bias_pad.shape[0]
    gx = torch.empty_like(x)
    gh = torch.empty(n, 64, dtype=torch.float32, device=x.device)
    acc = torch.zeros(groups * 4 * pp * 65, dtype=torch.float32, device=x.device)      # one memset for both accumulators
    gb = acc[:groups * 4 * pp].view(groups, 4, pp)
    gw = acc[groups * 4 * pp:].view(groups, 4, pp, 64)
    args = (_hip.ptr(x), _hip.ptr(h), _hip.ptr(gy), _hip.ptr(gl), _hip.ptr(w_frag), _hip.ptr(w_un),
            _hip.ptr(bias_pad), _hip.ptr(packed_t), _hip.ptr(cols), _hip.ptr(gx), _hip.ptr(gh), _hip.ptr(gb),
            _hip.ptr(gw), n, d, d_t, cfg, _hip.stream_ptr(x.device))
    _call("fc_rq_fused_linear_backward", lib.fc_rq_fused_linear_backward, x.device, 3, *args)      # FC_RQ_BACKWARD_ONE_LAUNCH
    grad_w = gw.reshape(groups * 4, pp, 64)[:d_t, :p].reshape(d_t * p, 64)
    grad_b = gb.reshape(groups * 4, pp)[:d_t, :p].reshape(d_t * p)
    return gx, gh, grad_w, grad_b


def fused_backward_supported(n, d, d_t, hidden, num_bins, tails):
    """Shapes of the fused training path: hidden <= 64, 3K -/+ 1 <= 32 parameters per dim (K <= 10 / 11), <= 32 dims per
    launch, D <= 128."""
    if tails not in (None, "linear") or num_bins not in GENERAL_BINS or hidden > 64:
        return False
    p = 3 * num_bins - 1 if tails == "linear" else 3 * num_bins + 1
    return p <= 32 and 1 <= d_t <= FUSED_DT and d <= 128 and n >= FUSED_ROWS


def rq_spline_fused_general(inputs, hidden, w_frag, w_unscale, bias_pad, cols, *, num_bins, tails, tail_bound=1.0,
                            left=0.0, right=1.0, bottom=0.0, top=1.0, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                            min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE,
                            wh_divisor=1.0, inverse=False, logabsdet_accum=None, enable_identity_init=False,
                            streamed_weights=False):
    """RQ-spline coupling bijector with the conditioner's final Linear fused in, general shapes (rows a multiple of
    32; ``hidden`` [N, 64 / 128 / 256] zero-padded; packed weights from ``pack_final_layer_general``).  Semantics and
    return values as ``rq_spline_fused_linear``; without tails inputs outside the box raise InputOutsideDomain.
    ``streamed_weights``: never the resident-weight instances (A/B measurements and tests)."""
    lib = _hip.load()
    x = _prep_2d(inputs, align16=True)
    h = _aligned16(_hip.dev_f32(hidden, "hidden"))
    _hip.require_no_grad(inputs, hidden)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = cols.numel()
    hw = h.shape[1]
    if (n % FUSED_ROWS != 0 or h.shape[0] != n or hw not in GENERAL_HIDDEN or not 1 <= d_t <= FUSED_DT
            or w_frag.shape[0] != -(-d_t // 4) or w_frag.shape[1] != hw // 32):
        raise ValueError("general fused RQ layer: unsupported shapes %s / %s" % (tuple(x.shape), tuple(h.shape)))
    cfg = _rq_config(num_bins, tails, tail_bound, (left, right, bottom, top), min_bin_width, min_bin_height,
                     min_derivative, enable_identity_init, wh_divisor, inverse)
    y = torch.empty_like(x)
    if logabsdet_accum is not None:
        lad = logabsdet_accum
        if lad.dtype != torch.float32 or lad.shape != (n,) or not lad.is_contiguous() or lad.device != x.device:
            raise ValueError("logabsdet_accum must be a contiguous float32 [N] tensor on the inputs' device")
        cfg.flags = 1  # FC_RQ_ACCUMULATE_LOGABSDET
    else:
        lad = torch.empty(n, dtype=torch.float32, device=x.device)
    if streamed_weights:
        cfg.flags |= 8  # FC_RQ_STREAMED_WEIGHTS
    err = _err_word(x.device, True)
    _call("fc_rq_spline_fused_general", lib.fc_rq_spline_fused_general, x.device, _hip.ptr(x), _hip.ptr(y),
          _hip.ptr(h), _hip.ptr(w_frag), _hip.ptr(w_unscale), _hip.ptr(bias_pad), _hip.ptr(cols), _hip.ptr(lad),
          _hip.ptr(err), n, d, d_t, hw, cfg, _hip.stream_ptr(x.device))
    _finish(True)
    return y, lad


# ---- affine / additive ------------------------------------------------------------------------

AFFINE_SIGMOID_PLUS2 = 0
AFFINE_SOFTPLUS_CLAMP3 = 1
AFFINE_SCALE_GIVEN = 2
AFFINE_ADDITIVE = 3
AFFINE_MAF_SOFTPLUS = 4
AFFINE_SHIFT_TANH2 = 5
AFFINE_SCALE_SOFTPLUS = 6


def affine_coupling(inputs, params, cols=None, *, activation=AFFINE_SIGMOID_PLUS2, inverse=False,
                    shared_params=False, logabsdet_accum=None):
    """Affine bijector on ``inputs[:, cols]`` with per-sample ``params`` rows; records an autograd node when
    gradients are required (forward direction, per-sample parameters).  See ``_affine_coupling_nograd``.
    ``logabsdet_accum``: running total the kernel adds onto (no-grad calls; otherwise added here)."""
    if torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad):
        if shared_params:
            raise RuntimeError("flowconductor_amd: gradients are implemented for the affine bijector with "
                               "per-sample parameters; wrap other calls in torch.no_grad().")
        if inverse:
            outputs, logabsdet = _inverse_through_forward(
                lambda y, p: _AffineFunction.apply(y, p, cols, activation),
                lambda x, p: _affine_coupling_nograd(x, p, cols, activation=activation, inverse=True), inputs, params)
        else:
            outputs, logabsdet = _AffineFunction.apply(inputs, params, cols, activation)
        if logabsdet_accum is not None:
            logabsdet_accum += logabsdet
            logabsdet = logabsdet_accum
        return outputs, logabsdet
    return _affine_coupling_nograd(inputs, params, cols, activation=activation, inverse=inverse,
                                   shared_params=shared_params, logabsdet_accum=logabsdet_accum)


class _AffineFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, params, cols, activation):
        with torch.no_grad():
            outputs, logabsdet = _affine_coupling_nograd(inputs, params, cols, activation=activation)
        ctx.save_for_backward(inputs, params)
        ctx.cols, ctx.activation = cols, activation
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, params = ctx.saved_tensors
        lib = _hip.load()
        x = _prep_2d(inputs.detach())
        p = _hip.dev_f32(params.detach(), "params")
        gy = _hip.dev_f32(grad_outputs if grad_outputs is not None else torch.zeros_like(x), "grad_outputs")
        gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
        n, d = x.shape
        cols = _as_cols(ctx.cols, x.device)
        d_t = d if cols is None else cols.numel()
        gx = gy.clone()                   # identity columns pass the upstream gradient through
        gp = torch.zeros_like(p)
        _call("fc_affine_backward", lib.fc_affine_backward, x.device, _hip.ptr(x), _hip.ptr(p), _hip.ptr(cols),
              _hip.ptr(gy), _hip.ptr(gl), _hip.ptr(gx), _hip.ptr(gp), n, d, d_t, ctx.activation,
              _hip.stream_ptr(x.device))
        return gx, gp.view_as(params), None, None


def _affine_coupling_nograd(inputs, params, cols=None, *, activation=AFFINE_SIGMOID_PLUS2, inverse=False,
                            shared_params=False, logabsdet_accum=None):
    """Affine bijector on ``inputs[:, cols]`` with per-sample ``params`` rows.

    Row layouts per ``activation``: see ``FC_AFFINE_*`` in include/flowcon_hip.h.
    Semantics: reference coupling.py:234-269, autoregressive/autoregressive.py:97-129.
    With ``logabsdet_accum`` (contiguous f32 [N]) the kernel adds the layer's logabsdet onto it and returns it.
    """
    lib = _hip.load()
    x = _prep_2d(inputs)
    p = _hip.dev_f32(params, "params")
    _hip.require_no_grad(inputs, params)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = d if cols is None else cols.numel()
    rowlen = d_t if activation in (AFFINE_ADDITIVE, AFFINE_SHIFT_TANH2, AFFINE_SCALE_SOFTPLUS) else 2 * d_t
    want = rowlen if shared_params else n * rowlen
    if p.numel() != want:
        raise ValueError("params has %d elements, expected %d" % (p.numel(), want))
    y = torch.empty_like(x)
    if logabsdet_accum is not None:
        lad = logabsdet_accum
        if lad.dtype != torch.float32 or lad.shape != (n,) or not lad.is_contiguous() or lad.device != x.device:
            raise ValueError("logabsdet_accum must be a contiguous float32 [N] tensor on the inputs' device")
    else:
        lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_affine", lib.fc_affine, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p), _hip.ptr(cols),
          _hip.ptr(lad), n, d, d_t, activation, 1 if inverse else 0, 1 if shared_params else 0,
          LAD_STORE if logabsdet_accum is None else LAD_ACCUMULATE, _hip.stream_ptr(x.device))
    return y, lad


# ---- base-distribution epilogue -----------------------------------------------------------------

def _standard_normal_log_prob_nograd(noise, log_z, add=None):
    """``-0.5 * sum(noise**2, 1) - log_z (+ add)`` -> [N]  (reference distributions/normal.py:23-33)."""
    lib = _hip.load()
    z = _hip.dev_f32(noise, "inputs")
    _hip.require_no_grad(noise, add)
    n = z.shape[0]
    z2 = z.flatten(1) if z.dim() > 1 else z.reshape(n, 1)
    if add is not None:
        add = _hip.dev_f32(add, "logabsdet")
        if add.numel() != n:
            raise ValueError("logabsdet must have one entry per row")
    out = torch.empty(n, dtype=torch.float32, device=z.device)
    _call("fc_standard_normal_log_prob", lib.fc_standard_normal_log_prob, z.device, _hip.ptr(z2),
          _hip.ptr(add), _hip.ptr(out), n, z2.shape[1], float(log_z), _hip.stream_ptr(z.device))
    return out


class _StdNormalLogProbFunction(torch.autograd.Function):
    """Gradient of ``-0.5 * sum(z**2) - log_z + add``: ``-z`` per element, 1 for ``add`` (plain torch ops)."""

    @staticmethod
    def forward(ctx, noise, add, log_z):
        with torch.no_grad():
            out = _standard_normal_log_prob_nograd(noise, log_z, add)
        ctx.save_for_backward(noise)
        ctx.has_add = add is not None
        return out

    @staticmethod
    def backward(ctx, grad):
        (noise,) = ctx.saved_tensors
        gz = -noise * grad.reshape((-1,) + (1,) * (noise.dim() - 1))
        return gz, (grad if ctx.has_add else None), None


def standard_normal_log_prob(noise, log_z, add=None):
    """``-0.5 * sum(noise**2, 1) - log_z (+ add)`` -> [N]  (reference distributions/normal.py:23-33); records
    an autograd node when gradients are required."""
    if torch.is_grad_enabled() and (noise.requires_grad or (add is not None and add.requires_grad)):
        return _StdNormalLogProbFunction.apply(noise, add, log_z)
    return _standard_normal_log_prob_nograd(noise, log_z, add)


# ---- permutation ----------------------------------------------------------------------------------

class _PermuteFunction(torch.autograd.Function):
    """Gradient of a gather along ``dim`` = the gather by the inverse permutation (same HIP kernel)."""

    @staticmethod
    def forward(ctx, inputs, permutation, dim):
        with torch.no_grad():
            out = _permute_nograd(inputs, permutation, dim)
        ctx.permutation, ctx.dim = permutation, dim
        return out

    @staticmethod
    def backward(ctx, grad):
        inverse = torch.argsort(torch.as_tensor(ctx.permutation).to(grad.device).long())
        return _permute_nograd(grad.contiguous(), inverse, ctx.dim), None, None


def permute(inputs, permutation, dim=1):
    """``index_select(inputs, dim, permutation)`` bit-exactly (reference permutations.py:27-46); records an
    autograd node when gradients are required."""
    if torch.is_grad_enabled() and inputs.requires_grad:
        return _PermuteFunction.apply(inputs, permutation, dim)
    return _permute_nograd(inputs, permutation, dim)


def _permute_nograd(inputs, permutation, dim=1):
    lib = _hip.load()
    x = _hip.dev_f32(inputs, "inputs")
    _hip.require_no_grad(inputs)
    perm = _as_cols(permutation, x.device)
    d = x.shape[dim]
    outer = 1
    for s in x.shape[:dim]:
        outer *= s
    inner = 1
    for s in x.shape[dim + 1:]:
        inner *= s
    y = torch.empty_like(x)
    _call("fc_permute", lib.fc_permute, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(perm), outer, d,
          inner, _hip.stream_ptr(x.device))
    return y


# ---- batch-shared point-wise affine -------------------------------------------------------------

def _item_vector(t, item_shape, device, name):
    """Broadcast a scalar / per-feature tensor to the flattened item shape (1 or m entries)."""
    t = torch.as_tensor(t, dtype=torch.float32, device=device)
    if t.numel() == 1:
        return t.reshape(1).contiguous()
    try:
        return t.expand(item_shape).reshape(-1).contiguous()
    except RuntimeError:
        raise RuntimeError("%s of shape %s is not broadcastable to inputs of shape %s"
                           % (name, tuple(t.shape), tuple(item_shape)))


def pointwise_affine(inputs, scale, shift, inverse=False):
    """``inputs * scale + shift`` or ``(inputs - shift) / scale`` with batch-shared scale/shift
    (reference standard.py:54-68, normalization.py:171-204)."""
    lib = _hip.load()
    x = _hip.dev_f32(inputs, "inputs")
    _hip.require_no_grad(inputs)
    n = x.shape[0]
    item_shape = x.shape[1:]
    m = 1
    for s in item_shape:
        m *= s
    sc = _item_vector(scale, item_shape, x.device, "scale")
    sh = _item_vector(shift, item_shape, x.device, "shift")
    y = torch.empty_like(x)
    _call("fc_pointwise_affine", lib.fc_pointwise_affine, x.device, _hip.ptr(x), _hip.ptr(y),
          _hip.ptr(sc), _hip.ptr(sh), None, None, n, m, sc.numel(), sh.numel(), 1 if inverse else 0,
          _hip.stream_ptr(x.device))
    return y


class _PointwiseAffineFunction(torch.autograd.Function):
    """``pointwise_affine`` with gradients: the HIP kernel forward, broadcasting reductions backward
    (y = x s + b: dx = gy s, ds = sum gy x, db = sum gy; inverse y = (x - b) / s: dx = gy / s, ds = -sum gy y / s,
    db = -sum gy / s) -- ActNorm / point-wise affine layers of a flow that is being trained."""

    @staticmethod
    def forward(ctx, inputs, scale, shift, inverse):
        with torch.no_grad():
            outputs = pointwise_affine(inputs, scale, shift, inverse=inverse)
        ctx.save_for_backward(outputs if inverse else inputs, scale, shift)
        ctx.inverse = inverse
        return outputs

    @staticmethod
    def backward(ctx, gy):
        saved, scale, shift = ctx.saved_tensors
        s = scale.to(gy.dtype)
        if ctx.inverse:
            gx = gy / s
            gs = -(gx * saved).sum(0)
            gb = -gx.sum(0)
        else:
            gx = gy * s
            gs = (gy * saved).sum(0)
            gb = gy.sum(0)
        return gx, gs.sum_to_size(scale.shape), gb.sum_to_size(shift.shape), None


def pointwise_affine_autograd(inputs, scale, shift, inverse=False):
    """``pointwise_affine``; records an autograd node when gradients are required.  ``scale`` / ``shift`` broadcast
    against one batch item."""
    scale, shift = torch.as_tensor(scale, device=inputs.device), torch.as_tensor(shift, device=inputs.device)
    if torch.is_grad_enabled() and (inputs.requires_grad or scale.requires_grad or shift.requires_grad):
        return _PointwiseAffineFunction.apply(inputs, scale, shift, inverse)
    return pointwise_affine(inputs, scale, shift, inverse=inverse)


class _LULinearFunction(torch.autograd.Function):
    """``y = L (U x) + b`` / ``y = U^-1 L^-1 (x - b)`` by the HIP kernel; gradients by library GEMMs and triangular
    solves on the device (lu.py:56-91 under autograd)."""

    @staticmethod
    def forward(ctx, inputs, lower, upper, bias, inverse):
        with torch.no_grad():
            outputs = linear(inputs, upper, lower, bias, mode=LINEAR_LU_INVERSE if inverse else LINEAR_LU_FORWARD)
        ctx.save_for_backward(inputs, lower, upper, outputs)
        ctx.inverse = inverse
        return outputs

    @staticmethod
    def backward(ctx, gy):
        x, lower, upper, y = ctx.saved_tensors
        if not ctx.inverse:
            ux = x @ upper.T
            g_ux = gy @ lower
            return g_ux @ upper, gy.T @ ux, g_ux.T @ x, gy.sum(0), None
        # y = W^-1 (x - b), W = L U:  gz = W^-T gy;  dW = -gz^T y;  dL = dW U^T, dU = L^T dW
        t = torch.linalg.solve_triangular(upper.T, gy.T, upper=False)
        gz = torch.linalg.solve_triangular(lower.T, t, upper=True, unitriangular=True).T
        gw = -(gz.T @ y)
        return gz, gw @ upper.T, lower.T @ gw, -gz.sum(0), None


def lu_linear_autograd(inputs, lower, upper, bias, inverse=False):
    """LU-parameterised linear map with an autograd node (training / differentiable sampling)."""
    return _LULinearFunction.apply(_prep_2d(inputs), lower, upper, bias, inverse)


def batchnorm_eval(inputs, mean, std, weight, bias, inverse=False):
    """Eval-mode BatchNorm map and its inverse (reference normalization.py:98-141)."""
    lib = _hip.load()
    x = _hip.dev_f32(inputs, "inputs")
    _hip.require_no_grad(inputs)
    n, m = x.shape[0], int(np.prod(x.shape[1:]))
    vecs = [_hip.dev_f32(v.detach().reshape(-1), "batch-norm statistic") for v in (std, bias, mean, weight)]
    for v in vecs:
        if v.numel() != m:
            raise ValueError("Expected features = {}, got {}.".format(v.numel(), m))
    y = torch.empty_like(x)
    _call("fc_pointwise_affine", lib.fc_pointwise_affine, x.device, _hip.ptr(x), _hip.ptr(y),
          _hip.ptr(vecs[0]), _hip.ptr(vecs[1]), _hip.ptr(vecs[2]), _hip.ptr(vecs[3]), n, m, m, m,
          3 if inverse else 2, _hip.stream_ptr(x.device))
    return y


# ---- row-per-wavefront bijectors with dense parameters --------------------------------------------

MAX_ROW_FEATURES = 512


def _rows(inputs, name="inputs", align16=False):
    x = _prep_2d(inputs, name, align16)
    if x.shape[1] > MAX_ROW_FEATURES:
        raise ValueError("flowconductor_amd: %d features exceed the %d supported by the row kernels"
                         % (x.shape[1], MAX_ROW_FEATURES))
    return x


def _param(t, device, name):
    """A parameter operand (module-owned or produced per sample by a hyper-network) as a device f32 tensor.  These
    kernels have no backward: a parameter that still carries a graph is refused rather than silently detached."""
    if isinstance(t, torch.Tensor):
        _hip.require_no_grad(t)
    return _hip.dev_f32(torch.as_tensor(t).detach().to(device), name)


def householder(inputs, q_vectors, reverse=False):
    """Apply K Householder reflections (reference orthogonal.py:144-194).

    ``q_vectors``: ``[K, D]`` shared across the batch or ``[N, K, D]`` per sample."""
    lib = _hip.load()
    x = _rows(inputs)
    _hip.require_no_grad(inputs, q_vectors)
    q = _param(q_vectors, x.device, "q_vectors")
    n, d = x.shape
    per_sample = q.dim() == 3
    if q.shape[-1] != d or (per_sample and q.shape[0] != n) or q.dim() not in (2, 3):
        raise ValueError("q_vectors of shape %s do not match inputs %s" % (tuple(q.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    _call("fc_householder", lib.fc_householder, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(q), n, d,
          q.shape[-2], 1 if per_sample else 0, 1 if reverse else 0, _hip.stream_ptr(x.device))
    return y


def planar(inputs, w, u_hat, b, per_sample=False):
    """Planar flow forward + logabsdet (reference no_analytic_inv/planar.py:30-49; with ``per_sample``
    the [N, D] / [N] parameters of ConditionalPlanarTransform, conditional.py:824-838)."""
    lib = _hip.load()
    x = _rows(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    wv = _param(w, x.device, "w").reshape(-1)
    uv = _param(u_hat, x.device, "u").reshape(-1)
    bv = _param(b, x.device, "b").reshape(-1)
    rows = n if per_sample else 1
    if wv.numel() != rows * d or uv.numel() != rows * d or bv.numel() != rows:
        raise ValueError("planar parameters do not match %d features" % d)
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_planar", lib.fc_planar, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(lad), _hip.ptr(wv),
          _hip.ptr(uv), _hip.ptr(bv), n, d, 1 if per_sample else 0, _hip.stream_ptr(x.device))
    return y, lad


PER_SAMPLE_DENSE, PER_SAMPLE_DENSE_T, PER_SAMPLE_LU_FORWARD, PER_SAMPLE_LU_INVERSE = 0, 1, 2, 3


def linear_per_sample(inputs, matrices, mode=PER_SAMPLE_DENSE, offdiag_scale=1.0, eps=0.0, want_logabsdet=False):
    """Per-sample ``[N, D, D]`` matrices applied to the rows of ``inputs`` (reference
    conditional.py:275-401): dense ``M x`` / ``M^T x`` or the LU forms built from raw hyper-network
    output on the fly.  Returns ``outputs`` or ``(outputs, logabsdet)``."""
    lib = _hip.load()
    x = _rows(inputs)
    m = _hip.dev_f32(matrices, "matrices")
    _hip.require_no_grad(inputs, matrices)
    n, d = x.shape
    if m.numel() != n * d * d:
        raise ValueError("matrices must be [%d, %d, %d]" % (n, d, d))
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device) if want_logabsdet else None
    _call("fc_linear_per_sample", lib.fc_linear_per_sample, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(lad),
          _hip.ptr(m), n, d, mode, float(offdiag_scale), float(eps), _hip.stream_ptr(x.device))
    return (y, lad) if want_logabsdet else y


LINEAR_DENSE, LINEAR_LU_FORWARD, LINEAR_LU_INVERSE = 0, 1, 2


def linear(inputs, a, b=None, bias=None, mode=LINEAR_DENSE):
    """Dense [D, D] maps on rows: ``A x + bias``; ``B (A x) + bias``; ``A^-1 B^-1 (x - bias)``
    (reference linear.py:45-76, lu.py:56-91).  ``a``/``b`` are given untransposed."""
    lib = _hip.load()
    x = _rows(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    at = _param(a, x.device, "weight").t().contiguous()
    bt = _param(b, x.device, "weight").t().contiguous() if b is not None else None
    if at.shape != (d, d) or (bt is not None and bt.shape != (d, d)):
        raise ValueError("weights must be [%d, %d]" % (d, d))
    bv = _param(bias, x.device, "bias").reshape(-1) if bias is not None else None
    y = torch.empty_like(x)
    _call("fc_linear", lib.fc_linear, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(at), _hip.ptr(bt),
          _hip.ptr(bv), n, d, mode, _hip.stream_ptr(x.device))
    return y


class _HouseholderFunction(torch.autograd.Function):
    """``householder`` with batch-shared q-vectors and its HIP backward kernel (``fc_householder_backward``: the
    reflections are involutions, so the saved OUTPUT is walked back to every intermediate)."""

    @staticmethod
    def forward(ctx, inputs, q_vectors, reverse):
        with torch.no_grad():
            outputs = householder(inputs, q_vectors, reverse=reverse)
        ctx.save_for_backward(outputs, q_vectors)
        ctx.reverse = reverse
        return outputs

    @staticmethod
    def backward(ctx, grad_outputs):
        outputs, q_vectors = ctx.saved_tensors
        lib = _hip.load()
        y = _hip.dev_f32(outputs, "outputs")
        q = _hip.dev_f32(q_vectors.detach(), "q_vectors")
        gy = _hip.dev_f32(grad_outputs, "grad_outputs")
        n, d = y.shape
        gx = torch.empty_like(y)
        gq = torch.zeros_like(q)
        _call("fc_householder_backward", lib.fc_householder_backward, y.device, _hip.ptr(y), _hip.ptr(gy), _hip.ptr(q),
              _hip.ptr(gx), _hip.ptr(gq), n, d, q.shape[0], 1 if ctx.reverse else 0, _hip.stream_ptr(y.device))
        return gx, gq, None


def householder_autograd(inputs, q_vectors, reverse=False):
    """``householder`` -> (outputs, zeros); under autograd the shared-q form sits behind ``_HouseholderFunction``."""
    if torch.is_grad_enabled() and (inputs.requires_grad or q_vectors.requires_grad) and q_vectors.dim() == 2:
        x = _prep_2d(inputs)
        return _HouseholderFunction.apply(x, q_vectors, bool(reverse)), x.new_zeros(x.shape[0])
    return householder(inputs, q_vectors, reverse=reverse), inputs.new_zeros(inputs.shape[0])


class _PlanarFunction(torch.autograd.Function):
    """Shared-parameter ``planar`` with its HIP backward kernel (``fc_planar_backward``)."""

    @staticmethod
    def forward(ctx, inputs, w, u_hat, b):
        with torch.no_grad():
            outputs, logabsdet = planar(inputs, w, u_hat, b)
        ctx.save_for_backward(inputs, w, u_hat, b)
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, w, u_hat, b = ctx.saved_tensors
        lib = _hip.load()
        x = _hip.dev_f32(inputs.detach(), "inputs")
        n, d = x.shape
        wv = _hip.dev_f32(w.detach().reshape(-1), "w")
        uv = _hip.dev_f32(u_hat.detach().reshape(-1), "u")
        bv = _hip.dev_f32(b.detach().reshape(-1), "b")
        gy = _hip.dev_f32(grad_outputs if grad_outputs is not None else torch.zeros_like(x), "grad_outputs")
        gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
        gx = torch.empty_like(x)
        gpar = torch.zeros(2 * d + 1, dtype=torch.float32, device=x.device)      # gw | gu | gb, one zero fill
        _call("fc_planar_backward", lib.fc_planar_backward, x.device, _hip.ptr(x), _hip.ptr(gy), _hip.ptr(gl),
              _hip.ptr(wv), _hip.ptr(uv), _hip.ptr(bv), _hip.ptr(gx), _hip.ptr(gpar[:d]), _hip.ptr(gpar[d:2 * d]),
              _hip.ptr(gpar[2 * d:]), n, d, _hip.stream_ptr(x.device))
        return gx, gpar[:d].view_as(w), gpar[d:2 * d].view_as(u_hat), gpar[2 * d:].view_as(b)


def planar_autograd(inputs, w, u_hat, b):
    """Shared-parameter planar flow (no_analytic_inv/planar.py:30-49) with an autograd node when needed."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (inputs, w, u_hat, b)):
        return _PlanarFunction.apply(_prep_2d(inputs), w, u_hat, b)
    return planar(inputs, w, u_hat, b)


def _householder_backward(outputs, grad_outputs, q, reverse):
    """(grad_inputs, grad_q) of ``householder`` with shared q from its saved OUTPUT (``fc_householder_backward``)."""
    lib = _hip.load()
    n, d = outputs.shape
    gx = torch.empty_like(outputs)
    gq = torch.zeros_like(q)
    _call("fc_householder_backward", lib.fc_householder_backward, outputs.device, _hip.ptr(outputs), _hip.ptr(grad_outputs),
          _hip.ptr(q), _hip.ptr(gx), _hip.ptr(gq), n, d, q.shape[0], 1 if reverse else 0, _hip.stream_ptr(outputs.device))
    return gx, gq


class _SylvesterFunction(torch.autograd.Function):
    """Shared-parameter ``sylvester`` (no_analytic_inv/planar.py:144-166) with an explicit backward: the two Householder
    sequences through ``fc_householder`` / ``fc_householder_backward``, the products with R1 / R2 as library GEMMs, the
    tanh / log-determinant middle in ``fc_sylvester_mid_backward``."""

    @staticmethod
    def forward(ctx, inputs, q_vectors, r1, r2, bias):
        with torch.no_grad():
            outputs, logabsdet = sylvester(inputs, q_vectors, r1, r2, bias)
        ctx.save_for_backward(inputs, q_vectors, r1, r2, bias)
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, q_vectors, r1, r2, bias = ctx.saved_tensors
        lib = _hip.load()
        with torch.no_grad():
            x = _hip.dev_f32(inputs.detach(), "inputs")
            q = _hip.dev_f32(q_vectors.detach(), "q_vectors")
            r1d, r2d, bd = r1.detach().float(), r2.detach().float(), bias.detach().float().reshape(-1)
            n, d = x.shape
            gy = _hip.dev_f32(grad_outputs if grad_outputs is not None else torch.zeros_like(x), "grad_outputs")
            gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
            qtz = householder(x, q, reverse=True)                               # Q^T z
            pre = torch.addmm(bd, qtz, r1d.t())
            act = torch.tanh(pre)
            out = householder(act @ r2d.t(), q, reverse=False)                  # Q R2 act (saved output of that sequence)
            g_mid, gq_a = _householder_backward(out, gy, q, False)
            g_r2 = g_mid.t() @ act
            g_act = (g_mid @ r2d).contiguous()
            rd = (torch.diagonal(r1d) * torch.diagonal(r2d)).contiguous()
            sums = torch.zeros(2, d, dtype=torch.float32, device=x.device)        # g_bias | g_rd, one zero fill
            _call("fc_sylvester_mid_backward", lib.fc_sylvester_mid_backward, x.device, _hip.ptr(pre), _hip.ptr(g_act),
                  _hip.ptr(gl), _hip.ptr(rd), _hip.ptr(sums[0]), _hip.ptr(sums[1]), n, d, _hip.stream_ptr(x.device))
            g_pre = g_act
            g_r1 = g_pre.t() @ qtz
            g_x2, gq_b = _householder_backward(qtz, (g_pre @ r1d).contiguous(), q, True)
            g_r1.diagonal().add_(sums[1] * torch.diagonal(r2d))
            g_r2.diagonal().add_(sums[1] * torch.diagonal(r1d))
            return gy + g_x2, gq_a + gq_b, g_r1, g_r2, sums[0].view_as(bias)


def sylvester_autograd(inputs, q_vectors, r1, r2, bias):
    """Shared-parameter Sylvester flow (no_analytic_inv/planar.py:144-166) with an autograd node when needed."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (inputs, q_vectors, r1, r2, bias)):
        return _SylvesterFunction.apply(_prep_2d(inputs), q_vectors, r1, r2, bias)
    return sylvester(inputs, q_vectors, r1, r2, bias)


def sylvester(inputs, q_vectors, r1, r2, bias):
    """Sylvester flow forward + logabsdet (reference no_analytic_inv/planar.py:144-166).

    Shared parameters: ``q [M, D]``, ``r1``/``r2`` ``[D, D]`` upper triangular, ``bias [D]``;
    per-sample: ``q [N, M, D]``, ``r1``/``r2`` ``[N, D, D]``, ``bias [N, D]``."""
    lib = _hip.load()
    x = _rows(inputs)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    q = _param(q_vectors, x.device, "q_vectors")
    r1 = _param(r1, x.device, "R1")
    r2 = _param(r2, x.device, "R2")
    bv = _param(bias, x.device, "bias")
    per_sample = r1.dim() == 3
    if per_sample != (q.dim() == 3) or per_sample != (bv.dim() == 2):
        raise ValueError("q, R1, R2 and bias must all be shared or all be per-sample")
    rdiag = (torch.diagonal(r1, dim1=-2, dim2=-1) * torch.diagonal(r2, dim1=-2, dim2=-1)).contiguous()
    if per_sample:
        # [N, D, D] row-major exactly as the hyper-network emits them: the kernel reads each row's upper part once
        # (a transposed copy would cost two more passes over 2 x N x D x D floats)
        r1t, r2t = r1.contiguous(), r2.contiguous()
    else:
        r1t = r1.transpose(-1, -2).contiguous()
        r2t = r2.transpose(-1, -2).contiguous()
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_sylvester", lib.fc_sylvester, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(lad), _hip.ptr(q),
          _hip.ptr(r1t), _hip.ptr(r2t), _hip.ptr(bv), _hip.ptr(rdiag), n, d, q.shape[-2],
          1 if per_sample else 0, _hip.stream_ptr(x.device))
    return y, lad


SYLVESTER_MM_ROWS = 16


def sylvester_mm_supported(n, d):
    """Shapes of the matrix-core Sylvester kernel (shared parameters only)."""
    return d % 32 == 0 and d <= 128 and n >= SYLVESTER_MM_ROWS


def householder_matrix(q_vectors, reverse=False):
    """[D, D] float64 matrix M with ``householder(v, q, reverse) == v @ M`` (rows as vectors): the K reflections
    of orthogonal.py:144-171 applied to the identity.  Host-side helper for batch-independent parameters."""
    q = q_vectors.detach().double()
    if reverse:
        q = q.flip(0)
    m = torch.eye(q.shape[1], dtype=torch.float64, device=q.device)
    for i in range(q.shape[0]):
        qi = q[i]
        m = m - torch.outer(m @ qi, (2.0 / (qi @ qi)) * qi)
    return m


def pack_sylvester(q_vectors, r1, r2):
    """(W1, W2, r_diag_prod) of ``fc_sylvester_mm`` from the reference's parameters: with rows as vectors
    Q^T z = z @ Mr (reflections in reverse order) and Q v = v @ Mf, so W1 = R1 Mr^T and W2 = Mf^T R2; formed in
    float64, rounded once."""
    mr = householder_matrix(q_vectors, reverse=True)
    mf = householder_matrix(q_vectors, reverse=False)
    w1 = (r1.detach().double() @ mr.T).float().contiguous()
    w2 = (mf.T @ r2.detach().double()).float().contiguous()
    rdiag = (torch.diagonal(r1.detach()) * torch.diagonal(r2.detach())).float().contiguous()
    return w1, w2, rdiag


def dense_mm(inputs, weight, bias=None):
    """``inputs @ weight.T + bias`` for a batch-independent [D, D] ``weight`` on the matrix cores (rows a multiple
    of 16, D % 32 == 0, D <= 128): f32-GEMM accuracy by split-f16 products."""
    lib = _hip.load()
    x = _rows(inputs, align16=True)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % SYLVESTER_MM_ROWS != 0 or not sylvester_mm_supported(n, d):
        raise ValueError("fc_dense_mm: unsupported shape %s" % (tuple(x.shape),))
    w = _aligned16(_param(weight, x.device, "weight"))
    if w.shape != (d, d):
        raise ValueError("weight must be [%d, %d]" % (d, d))
    bv = _param(bias, x.device, "bias").reshape(-1) if bias is not None else None
    y = torch.empty_like(x)
    _call("fc_dense_mm", lib.fc_dense_mm, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(w), _hip.ptr(bv), n, d,
          _hip.stream_ptr(x.device))
    return y


def sylvester_mm(inputs, w1, w2, bias, rdiag):
    """Sylvester flow forward + logabsdet with shared parameters as two matrix-core products (rows a multiple of 16)."""
    lib = _hip.load()
    x = _rows(inputs, align16=True)
    _hip.require_no_grad(inputs)
    n, d = x.shape
    if n % SYLVESTER_MM_ROWS != 0 or not sylvester_mm_supported(n, d):
        raise ValueError("fc_sylvester_mm: unsupported shape %s" % (tuple(x.shape),))
    bv = _param(bias, x.device, "bias")
    w1, w2 = _aligned16(_hip.dev_f32(w1, "w1")), _aligned16(_hip.dev_f32(w2, "w2"))
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_sylvester_mm", lib.fc_sylvester_mm, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(lad), _hip.ptr(w1),
          _hip.ptr(w2), _hip.ptr(bv), _hip.ptr(rdiag), n, d, _hip.stream_ptr(x.device))
    return y, lad


# ---- element-wise non-linearities -----------------------------------------------------------------

EW_EXP, EW_TANH, EW_LOGTANH, EW_LEAKY_RELU, EW_SIGMOID, EW_SOFTPLUS, EW_CAUCHY_CDF = range(7)
EW_EXTENDED_SOFTPLUS, EW_GLU = 7, 8


def elementwise(inputs, kind, inverse=False, aux=None, p=(0.0, 0.0, 0.0, 0.0), row_sum=True,
                elem_lad=False, may_raise=False):
    """Element-wise bijector ``kind`` (``FC_EW_*``) over ``[N, ...]`` inputs.

    Returns ``(outputs, logabsdet)`` with ``logabsdet`` summed over everything but the batch dim
    (``row_sum``) or left per element (``elem_lad``)."""
    lib = _hip.load()
    x = _hip.dev_f32(inputs, "inputs")
    _hip.require_no_grad(inputs)
    n = x.shape[0]
    m = 1
    for s in x.shape[1:]:
        m *= s
    if aux is not None:
        aux = _hip.dev_f32(torch.as_tensor(aux).detach().to(x.device), "parameter")
    y = torch.empty_like(x)
    lad_row = torch.empty(n, dtype=torch.float32, device=x.device) if row_sum else None
    lad_el = torch.empty_like(x) if elem_lad else None
    err = _err_word(x.device, may_raise)
    p = tuple(float(v) for v in p) + (0.0,) * (4 - len(p))
    _call("fc_elementwise", lib.fc_elementwise, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(lad_row),
          _hip.ptr(lad_el), _hip.ptr(aux), _hip.ptr(err), n, m, kind, 1 if inverse else 0, p[0], p[1], p[2],
          p[3], _hip.stream_ptr(x.device))
    _finish(may_raise)
    return y, (lad_el if elem_lad else lad_row)


# ---- sum of sigmoids ------------------------------------------------------------------------------

class _SoSFunction(torch.autograd.Function):
    """``sum_of_sigmoids`` (forward direction, per-sample raw parameters) with its HIP backward kernel
    (``fc_sum_of_sigmoids_backward``: closed-form derivatives of adaptive_sigmoids.py:108-142)."""

    @staticmethod
    def forward(ctx, inputs, raw_params, n_sigmoids, offset, log_scale_postact):
        with torch.no_grad():
            outputs, logabsdet = sum_of_sigmoids(inputs, raw_params, n_sigmoids, offset=offset,
                                                 log_scale_postact=log_scale_postact)
        ctx.save_for_backward(inputs, raw_params)
        ctx.n_sigmoids, ctx.log_post = n_sigmoids, log_scale_postact
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, raw_params = ctx.saved_tensors
        lib = _hip.load()
        x = _prep_2d(inputs.detach())
        p = _hip.dev_f32(raw_params.detach(), "raw_params")
        gy = _hip.dev_f32(grad_outputs if grad_outputs is not None else torch.zeros_like(x), "grad_outputs")
        gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
        n, d = x.shape
        gx = torch.empty_like(x)
        gp = torch.empty_like(p)
        _call("fc_sum_of_sigmoids_backward", lib.fc_sum_of_sigmoids_backward, x.device, _hip.ptr(x), _hip.ptr(p),
              _hip.ptr(gy), _hip.ptr(gl), _hip.ptr(gx), _hip.ptr(gp), n, d, ctx.n_sigmoids, float(ctx.log_post),
              _hip.stream_ptr(x.device))
        return gx, gp.view_as(raw_params), None, None, None


def sum_of_sigmoids_autograd(inputs, raw_params, n_sigmoids, inverse=False, offset=0.0, iterations=50, lim=120.0,
                             shared_params=False):
    """``sum_of_sigmoids``; with autograd on, the forward kernel sits behind ``_SoSFunction`` (HIP backward kernel; a
    batch-shared parameter row is expanded to per-sample rows, autograd sums its gradient), and the inverse goes through
    ``_inverse_through_forward``."""
    if not (torch.is_grad_enabled() and (inputs.requires_grad or raw_params.requires_grad)):
        return sum_of_sigmoids(inputs, raw_params, n_sigmoids, inverse=inverse, offset=offset, iterations=iterations,
                               lim=lim, shared_params=shared_params)
    x = _prep_2d(inputs)

    def forward_fn(v, raw):
        rows = raw.reshape(1, -1).expand(v.shape[0], -1) if shared_params else raw.reshape(v.shape[0], -1)
        return _SoSFunction.apply(v, rows.contiguous(), n_sigmoids, offset, 0.0)

    if not inverse:
        return forward_fn(x, raw_params)
    return _inverse_through_forward(
        forward_fn, lambda v, raw: sum_of_sigmoids(v, raw, n_sigmoids, inverse=True, offset=offset,
                                                   iterations=iterations, lim=lim, shared_params=shared_params),
        x, raw_params)


def sum_of_sigmoids(inputs, raw_params, n_sigmoids, inverse=False, offset=0.0, iterations=50, lim=120.0,
                    log_scale_postact=0.0, shared_params=False):
    """Sum-of-sigmoids bijector (reference adaptive_sigmoids.py:108-142; inverse base.py:23-83).

    ``raw_params``: ``[N, D, 3S+1]`` per-sample rows, or ``[D, 3S+1]`` with ``shared_params``."""
    lib = _hip.load()
    x = _prep_2d(inputs)
    p = _hip.dev_f32(raw_params, "raw_params")
    _hip.require_no_grad(inputs, raw_params)
    n, d = x.shape
    rowlen = d * (3 * n_sigmoids + 1)
    want = rowlen if shared_params else n * rowlen
    if p.numel() != want:
        raise ValueError("raw_params has %d elements, expected %d" % (p.numel(), want))
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    err = _err_word(x.device, inverse)
    _call("fc_sum_of_sigmoids", lib.fc_sum_of_sigmoids, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p), None,
          _hip.ptr(lad), _hip.ptr(err), n, d, d, n_sigmoids, 1 if inverse else 0, int(iterations), float(lim),
          float(offset), float(log_scale_postact), 1 if shared_params else 0, LAD_STORE,
          _hip.stream_ptr(x.device))
    _finish(inverse)
    return y, lad


# ---- linear / quadratic / cubic splines -----------------------------------------------------------

SPLINE_LINEAR, SPLINE_QUADRATIC, SPLINE_CUBIC = 0, 1, 2


def spline_multiplier(kind, num_bins, tails):
    if kind == SPLINE_LINEAR:
        return num_bins
    if kind == SPLINE_QUADRATIC:
        return num_bins * 2 - 1 if tails == "linear" else num_bins * 2 + 1
    return num_bins * 2 + 2


def _spline_config(kind, num_bins, tails, tail_bound, box, min_bin_width, min_bin_height, width_divisor,
                   height_divisor, inverse):
    cfg = _hip.SplineConfig()
    cfg.kind, cfg.num_bins = kind, num_bins
    cfg.tails = 0 if tails is None else 1
    cfg.inverse = 1 if inverse else 0
    if tails == "linear":
        cfg.left, cfg.right, cfg.bottom, cfg.top = -tail_bound, tail_bound, -tail_bound, tail_bound
    else:
        cfg.left, cfg.right, cfg.bottom, cfg.top = box
    cfg.min_bin_width, cfg.min_bin_height = min_bin_width, min_bin_height
    cfg.width_divisor, cfg.height_divisor = width_divisor, height_divisor
    cfg.cubic_eps, cfg.cubic_quadratic_threshold = 1e-5, 1e-3
    return cfg


class _PiecewiseSplineFunction(torch.autograd.Function):
    """``piecewise_spline`` (forward direction, per-sample rows) with its HIP backward kernel
    (``fc_piecewise_spline_backward``: forward-mode derivative of the kernel's own evaluation, one thread per
    (element, parameter))."""

    @staticmethod
    def forward(ctx, inputs, params, cols, kw):
        with torch.no_grad():
            outputs, logabsdet = piecewise_spline(inputs, params, cols, **kw)
        ctx.save_for_backward(inputs, params)
        ctx.cols, ctx.kw = cols, kw
        return outputs, logabsdet

    @staticmethod
    def backward(ctx, grad_outputs, grad_logabsdet):
        inputs, params = ctx.saved_tensors
        kw = dict(ctx.kw)
        lib = _hip.load()
        x = _prep_2d(inputs.detach())
        p = _hip.dev_f32(params.detach(), "params")
        n, d = x.shape
        cols = _as_cols(ctx.cols, x.device)
        d_t = d if cols is None else cols.numel()
        gy = _hip.dev_f32(grad_outputs if grad_outputs is not None else torch.zeros_like(x), "grad_outputs")
        gl = None if grad_logabsdet is None else _hip.dev_f32(grad_logabsdet, "grad_logabsdet")
        cfg = _spline_config(kw["kind"], kw["num_bins"], kw.get("tails"), kw.get("tail_bound", 1.0),
                             (kw.get("left", 0.0), kw.get("right", 1.0), kw.get("bottom", 0.0), kw.get("top", 1.0)),
                             kw.get("min_bin_width", DEFAULT_MIN_BIN_WIDTH), kw.get("min_bin_height", DEFAULT_MIN_BIN_HEIGHT),
                             kw.get("width_divisor", 1.0), kw.get("height_divisor", 1.0), False)
        gx = gy.clone() if d_t < d else torch.empty_like(x)       # identity columns pass the gradient through
        gp = torch.empty_like(p)
        _call("fc_piecewise_spline_backward", lib.fc_piecewise_spline_backward, x.device, _hip.ptr(x), _hip.ptr(p),
              _hip.ptr(cols), _hip.ptr(gy), _hip.ptr(gl), _hip.ptr(gx), _hip.ptr(gp), n, d, d_t, cfg,
              _hip.stream_ptr(x.device))
        return gx, gp.view_as(params), None, None


def piecewise_spline_backward_supported(kind, num_bins, tails):
    """``fc_piecewise_spline_backward`` keeps an element's parameters as dual numbers in LDS: <= 32 per element."""
    return spline_multiplier(kind, num_bins, tails) <= 32


def piecewise_spline_autograd(inputs, params, cols=None, *, inverse=False, shared_params=False, **kw):
    """``piecewise_spline``; under autograd (per-sample parameters) the kernel's forward sits behind a node whose
    gradients come from the same spline in torch ops, the inverse goes through ``_inverse_through_forward``."""
    if not (torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad)) or shared_params:
        return piecewise_spline(inputs, params, cols, inverse=inverse, shared_params=shared_params, **kw)
    x = _prep_2d(inputs)
    if not piecewise_spline_backward_supported(kw["kind"], kw["num_bins"], kw.get("tails")):
        raise NotImplementedError("autograd through piecewise splines needs <= 32 parameters per element "
                                  "(fc_piecewise_spline_backward); got num_bins = %d" % kw["num_bins"])

    def forward_fn(v, p):
        return _PiecewiseSplineFunction.apply(v, p, cols, kw)

    if not inverse:
        return forward_fn(x, params)
    return _inverse_through_forward(forward_fn, lambda a, b: piecewise_spline(a, b, cols, inverse=True, **kw), x, params)


def piecewise_spline(inputs, params, cols=None, *, kind, num_bins, tails=None, tail_bound=1.0,
                     left=0.0, right=1.0, bottom=0.0, top=1.0, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                     min_bin_height=DEFAULT_MIN_BIN_HEIGHT, width_divisor=1.0, height_divisor=1.0,
                     inverse=False, shared_params=False):
    """Linear / quadratic / cubic spline over ``inputs[:, cols]`` (reference splines/{linear,
    quadratic,cubic}.py).  Row layouts: ``FC_SPLINE_*`` in include/flowcon_hip.h."""
    lib = _hip.load()
    x = _prep_2d(inputs)
    p = _hip.dev_f32(params, "params")
    _hip.require_no_grad(inputs, params)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = d if cols is None else cols.numel()
    if tails not in (None, "linear"):
        raise RuntimeError("{} tails are not implemented.".format(tails))
    if kind != SPLINE_LINEAR:
        if min_bin_width * num_bins > 1.0:
            raise ValueError("Minimal bin width too large for the number of bins")
        if min_bin_height * num_bins > 1.0:
            raise ValueError("Minimal bin height too large for the number of bins")
    rowlen = d_t * spline_multiplier(kind, num_bins, tails)
    want = rowlen if shared_params else n * rowlen
    if p.numel() != want:
        raise ValueError("params has %d elements, expected %d" % (p.numel(), want))
    cfg = _spline_config(kind, num_bins, tails, tail_bound, (left, right, bottom, top), min_bin_width, min_bin_height,
                         width_divisor, height_divisor, inverse)
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    err = _err_word(x.device, True)
    _call("fc_piecewise_spline", lib.fc_piecewise_spline, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p),
          _hip.ptr(cols), _hip.ptr(lad), _hip.ptr(err), n, d, d_t, 1 if shared_params else 0, LAD_STORE, cfg,
          _hip.stream_ptr(x.device))
    _finish(True)
    return y, lad
