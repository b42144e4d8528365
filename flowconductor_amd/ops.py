"""Functional host wrappers over the HIP kernels (tensor in -> tensor out).

The ``Transform`` modules in ``flowconductor_amd.transforms`` call these; they allocate the
outputs with torch, pass raw device pointers + the current HIP stream through the C ABI and
turn the device error word into the reference's Python exceptions.
"""
import contextlib
import math
import threading

import numpy as np
import torch

from flowconductor_amd import _hip


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


def _prep_2d(inputs, name="inputs"):
    x = _hip.dev_f32(inputs, name)
    if x.dim() != 2:
        raise ValueError("%s must be [batch, features], got shape %s" % (name, tuple(x.shape)))
    return x


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

    cfg = _hip.RQConfig()
    cfg.num_bins = num_bins
    cfg.tails = 0 if tails is None else 1
    cfg.inverse = 1 if inverse else 0
    if tails == "linear":
        cfg.left, cfg.right, cfg.bottom, cfg.top = -tail_bound, tail_bound, -tail_bound, tail_bound
    else:
        cfg.left, cfg.right, cfg.bottom, cfg.top = left, right, bottom, top
    cfg.min_bin_width = min_bin_width
    cfg.min_bin_height = min_bin_height
    cfg.min_derivative = min_derivative
    cfg.wh_divisor = wh_divisor
    cfg.softplus_beta = (math.log(2) / (1 - min_derivative)) if enable_identity_init else 1.0
    cfg.tail_constant = float(np.log(np.exp(1 - min_derivative) - 1))

    y = torch.empty_like(x) if out is None else out
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    err = _err_word(x.device, True)
    _call("fc_rq_spline", lib.fc_rq_spline, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p),
          _hip.ptr(cols), _hip.ptr(lad), _hip.ptr(err), n, d, d_t, 1 if shared_params else 0,
          LAD_STORE, cfg, _hip.stream_ptr(x.device))
    _finish(True)
    return y, lad


# ---- affine / additive ------------------------------------------------------------------------

AFFINE_SIGMOID_PLUS2 = 0
AFFINE_SOFTPLUS_CLAMP3 = 1
AFFINE_SCALE_GIVEN = 2
AFFINE_ADDITIVE = 3
AFFINE_MAF_SOFTPLUS = 4
AFFINE_SHIFT_TANH2 = 5


def affine_coupling(inputs, params, cols=None, *, activation=AFFINE_SIGMOID_PLUS2, inverse=False,
                    shared_params=False):
    """Affine bijector on ``inputs[:, cols]`` with per-sample ``params`` rows.

    Row layouts per ``activation``: see ``FC_AFFINE_*`` in include/flowcon_hip.h.
    Semantics: reference coupling.py:234-269, autoregressive/autoregressive.py:97-129.
    """
    lib = _hip.load()
    x = _prep_2d(inputs)
    p = _hip.dev_f32(params, "params")
    _hip.require_no_grad(inputs, params)
    n, d = x.shape
    cols = _as_cols(cols, x.device)
    d_t = d if cols is None else cols.numel()
    rowlen = d_t if activation in (AFFINE_ADDITIVE, AFFINE_SHIFT_TANH2) else 2 * d_t
    want = rowlen if shared_params else n * rowlen
    if p.numel() != want:
        raise ValueError("params has %d elements, expected %d" % (p.numel(), want))
    y = torch.empty_like(x)
    lad = torch.empty(n, dtype=torch.float32, device=x.device)
    _call("fc_affine", lib.fc_affine, x.device, _hip.ptr(x), _hip.ptr(y), _hip.ptr(p), _hip.ptr(cols),
          _hip.ptr(lad), n, d, d_t, activation, 1 if inverse else 0, 1 if shared_params else 0,
          LAD_STORE, _hip.stream_ptr(x.device))
    return y, lad


# ---- base-distribution epilogue -----------------------------------------------------------------

def standard_normal_log_prob(noise, log_z, add=None):
    """``-0.5 * sum(noise**2, 1) - log_z (+ add)`` -> [N]  (reference distributions/normal.py:23-33)."""
    lib = _hip.load()
    z = _hip.dev_f32(noise, "inputs")
    _hip.require_no_grad(noise, add)
    n = z.shape[0]
    z2 = z.reshape(n, -1)
    if add is not None:
        add = _hip.dev_f32(add, "logabsdet")
        if add.numel() != n:
            raise ValueError("logabsdet must have one entry per row")
    out = torch.empty(n, dtype=torch.float32, device=z.device)
    _call("fc_standard_normal_log_prob", lib.fc_standard_normal_log_prob, z.device, _hip.ptr(z2),
          _hip.ptr(add), _hip.ptr(out), n, z2.shape[1], float(log_z), _hip.stream_ptr(z.device))
    return out


# ---- permutation ----------------------------------------------------------------------------------

def permute(inputs, permutation, dim=1):
    """``index_select(inputs, dim, permutation)`` bit-exactly (reference permutations.py:27-46)."""
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


def batchnorm_eval(inputs, mean, std, weight, bias, inverse=False):
    """Eval-mode BatchNorm map and its inverse (reference normalization.py:98-141)."""
    lib = _hip.load()
    x = _hip.dev_f32(inputs, "inputs")
    _hip.require_no_grad(inputs)
    n, m = x.shape[0], x[0].numel()
    vecs = [_hip.dev_f32(v.detach().reshape(-1), "batch-norm statistic") for v in (std, bias, mean, weight)]
    for v in vecs:
        if v.numel() != m:
            raise ValueError("Expected features = {}, got {}.".format(v.numel(), m))
    y = torch.empty_like(x)
    _call("fc_pointwise_affine", lib.fc_pointwise_affine, x.device, _hip.ptr(x), _hip.ptr(y),
          _hip.ptr(vecs[0]), _hip.ptr(vecs[1]), _hip.ptr(vecs[2]), _hip.ptr(vecs[3]), n, m, m, m,
          3 if inverse else 2, _hip.stream_ptr(x.device))
    return y
