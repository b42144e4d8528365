"""ctypes binding of ``libflowcon_hip.so`` (the C ABI declared in ``include/flowcon_hip.h``).

This is the only place the package touches native code.  There is NO CPU or eager-PyTorch
fallback: if the library is missing, or a tensor is not a contiguous float32 HIP tensor, the
call raises.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, built in-tree.  Nothing in the environment redirects it: profiling tools that want an ablation
# build of the same ABI call ``use_library(path)`` explicitly before the first kernel call (tools/ only).
LIB_PATH = os.path.join(_HERE, "csrc", "libflowcon_hip.so")

ABI_VERSION = 2

ERR_OUTSIDE_DOMAIN = 1
ERR_DISCRIMINANT = 2
ERR_NONFINITE = 4


class HipLibraryMissing(RuntimeError):
    pass


class RQConfig(ctypes.Structure):
    """Mirror of ``fc_rq_config``."""

    _fields_ = [
        ("num_bins", ctypes.c_int32),
        ("tails", ctypes.c_int32),
        ("inverse", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("left", ctypes.c_float),
        ("right", ctypes.c_float),
        ("bottom", ctypes.c_float),
        ("top", ctypes.c_float),
        ("min_bin_width", ctypes.c_double),
        ("min_bin_height", ctypes.c_double),
        ("min_derivative", ctypes.c_double),
        ("wh_divisor", ctypes.c_float),
        ("softplus_beta", ctypes.c_float),
        ("tail_constant", ctypes.c_float),
        ("reserved2", ctypes.c_float),
    ]


class SplineConfig(ctypes.Structure):
    """Mirror of ``fc_spline_config``."""

    _fields_ = [
        ("kind", ctypes.c_int32),
        ("num_bins", ctypes.c_int32),
        ("tails", ctypes.c_int32),
        ("inverse", ctypes.c_int32),
        ("left", ctypes.c_float),
        ("right", ctypes.c_float),
        ("bottom", ctypes.c_float),
        ("top", ctypes.c_float),
        ("min_bin_width", ctypes.c_double),
        ("min_bin_height", ctypes.c_double),
        ("width_divisor", ctypes.c_float),
        ("height_divisor", ctypes.c_float),
        ("cubic_eps", ctypes.c_float),
        ("cubic_quadratic_threshold", ctypes.c_float),
    ]


class PackJob(ctypes.Structure):
    """Mirror of ``fc_pack_job``."""

    _fields_ = [("w", ctypes.c_void_p), ("b", ctypes.c_void_p), ("frag", ctypes.c_void_p), ("unscale", ctypes.c_void_p),
                ("bias_out", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("p", ctypes.c_int32), ("pp", ctypes.c_int32), ("nks", ctypes.c_int32),
                ("nt", ctypes.c_int32), ("group", ctypes.c_int32)]


_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_F = ctypes.c_float

# name -> argtypes; every symbol here must be declared in include/flowcon_hip.h (tests check)
SIGNATURES = {
    "fc_abi_version": [],
    "fc_rq_spline": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32,
                     ctypes.POINTER(RQConfig), _P],
    "fc_rq_spline_backward": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, ctypes.POINTER(RQConfig), _P],
    "fc_standard_normal_log_prob": [_P, _P, _P, _I64, _I32, _F, _P],
    "fc_permute": [_P, _P, _P, _I64, _I32, _I64, _P],
    "fc_pointwise_affine": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P],
    "fc_householder": [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "fc_planar": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    "fc_linear_per_sample": [_P, _P, _P, _P, _I64, _I32, _I32, _F, _F, _P],
    "fc_linear": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    "fc_sylvester": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "fc_sum_of_sigmoids": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _F, _F, _F,
                           _I32, _I32, _P],
    "fc_sum_of_sigmoids_backward": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P],
    "fc_planar_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "fc_sylvester_mid_backward": [_P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "fc_householder_backward": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "fc_elementwise": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _F, _F, _F, _F, _P],
    "fc_piecewise_spline_backward": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, ctypes.POINTER(SplineConfig), _P],
    "fc_piecewise_spline": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32,
                            ctypes.POINTER(SplineConfig), _P],
    "fc_rq_spline_fused_linear": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32,
                                  ctypes.POINTER(RQConfig), _P],
    "fc_rq_spline_fused_general": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32,
                                   ctypes.POINTER(RQConfig), _P],
    "fc_rq_fused_linear_backward": [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32,
                                    ctypes.POINTER(RQConfig), _P],
    "fc_resnet_hidden": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, ctypes.c_float, _P],
    "fc_affine_coupling_resnet": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "fc_made_inverse": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, ctypes.POINTER(RQConfig), _P],
    "fc_resnet_hidden_packed": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, ctypes.c_float, _P],
    "fc_resnet_hidden_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "fc_resnet_hidden_backward_accum": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "fc_resnet_hidden_wide": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, ctypes.c_float, _P],
    "fc_resnet_hidden_context": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32,
                                 _I32, ctypes.c_float, _P],
    "fc_affine": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "fc_dense_mm": [_P, _P, _P, _P, _I64, _I32, _P],
    "fc_sylvester_mm": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "fc_affine_backward": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "fc_pack_fragments": [_P, _I32, _P],
    "fc_pack_job_bytes": [],
    "fc_comm_unique_id": [_P],
    "fc_comm_init_rank": [ctypes.POINTER(ctypes.c_void_p), _I32, _P, _I32],
    "fc_comm_init_rank_on_device": [ctypes.POINTER(ctypes.c_void_p), _I32, _P, _I32, _I32],
    "fc_comm_destroy": [_P],
    "fc_comm_abort": [_P],
    "fc_allreduce_loglik": [_P, _P, _P],
}

_lib = None


def use_library(path):
    """tools/ only: bind an ablation build of the same ABI instead of the product library.  Must be called before
    the first kernel call; refuses to swap a library that is already loaded."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("flowconductor_amd: %s is already loaded" % LIB_PATH)
    LIB_PATH = os.path.abspath(path)


def library_info():
    """``{"path", "sha256", "bytes"}`` of the library the kernels come from (bench.py records it)."""
    import hashlib

    with open(LIB_PATH, "rb") as f:
        blob = f.read()
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "sha256": hashlib.sha256(blob).hexdigest(),
            "bytes": len(blob)}


def load():
    """Load the shared library (once).  Raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "flowconductor_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C flowconductor_amd/csrc`. There is no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    got = lib.fc_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryMissing("libflowcon_hip.so ABI %d != expected %d: rebuild" % (got, ABI_VERSION))
    _lib = lib
    return lib


def is_built():
    return os.path.exists(LIB_PATH)


def check(code, what):
    if code != 0:
        raise RuntimeError("flowconductor_amd: %s failed with hipError %d" % (what, code))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """The current HIP stream of ``device`` as a ``void*`` (the raw-stream accessor when this torch has it: building a
    ``torch.cuda.Stream`` object per launch was 4 us of the ~13 us a launch costs on the host)."""
    if _raw_stream is not None:
        index = device.index
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if index is None else index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_f32(t, name):
    """Validate a tensor for the kernels: float32, on a HIP device, contiguous."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError(
            "flowconductor_amd: %s lives on %s; the bijector kernels run on a HIP device only "
            "(no CPU fallback)" % (name, t.device)
        )
    if t.dtype != torch.float32:
        raise TypeError("flowconductor_amd: %s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError(
            "flowconductor_amd: the HIP bijector kernels implement forward/inverse/logabsdet only; "
            "autograd through them is not available yet. Wrap the call in torch.no_grad()."
        )
