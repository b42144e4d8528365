"""Explicit switches of the fast paths.

Every dispatch decision in this package (which kernel serves a layer) is a function of the layer, its inputs and
these options -- never of the process environment: a stray environment variable cannot change what a benchmark
times.  The defaults are the product configuration; tools and tests flip a switch for A/B measurements with

    with options.override(fused_final_layer=False):
        flow.log_prob(x)

``override`` restores the previous values on exit (also on an exception) and nests.
"""
import contextlib

__all__ = ["get", "override", "snapshot"]

_DEFAULTS = {
    # conditioner's final Linear evaluated inside the RQ-spline kernel (fc_rq_spline_fused_linear / _general)
    "fused_final_layer": True,
    # conditioner's hidden layers in fc_resnet_hidden (one kernel instead of GEMMs + element-wise passes)
    "fused_hidden": True,
    # shared-parameter Sylvester / Householder / LU maps folded into dense matrices on the matrix cores
    "sylvester_mm": True,
    # autoregressive inverse: "auto" (column-d-only passes where they pay), "force", "off"
    "ar_incremental": "auto",
    # autoregressive inverse: all D passes inside one kernel (fc_made_inverse) where the MADE fits it
    "ar_device_loop": True,
    # fc_rq_spline: pin the LDS-tile kernel instead of the register / wave kernel (FC_RQ_FORCE_TILE)
    "rq_force_tile": False,
    # training: conditioner forward / backward in the HIP kernels (fc_resnet_hidden_backward, fused final-layer
    # backward) instead of PyTorch autograd through library GEMMs
    "fused_training": True,
    # Packed-weight caches (kernel-layout copies of parameters) are keyed on the parameters' version counters and storage
    # pointers; a write THROUGH ``.data`` (``p.data.copy_(ema)``) moves neither.  True: every call re-packs (one
    # fc_pack_fragments launch per layer, ~1 % of a cfg-3 log_prob) -- for code that edits ``.data`` of an eval-mode model
    # and cannot call ``ops.invalidate_hip_caches()`` after it.
    "paranoid_caches": False,
}

_values = dict(_DEFAULTS)


def get(name):
    return _values[name]


def snapshot():
    """Current values (bench.py records them in its JSON line)."""
    return dict(_values)


@contextlib.contextmanager
def override(**kw):
    unknown = set(kw) - set(_DEFAULTS)
    if unknown:
        raise KeyError("unknown option(s): %s" % ", ".join(sorted(unknown)))
    saved = {k: _values[k] for k in kw}
    _values.update(kw)
    try:
        yield
    finally:
        _values.update(saved)
