"""flowconductor_amd -- MI355X-native bijector hot path behind the FlowConductor Transform API.

Host side: Python ``nn.Module`` mirror of ``flowcon.transforms`` (same class names, constructor
arguments, ``state_dict`` keys and exceptions).  Device side: hand-written HIP kernels for
gfx950 in ``csrc/`` behind the C ABI of ``include/flowcon_hip.h``.  Conditioner networks stay
ordinary PyTorch-ROCm modules.  There is no CPU fallback.
"""
from flowconductor_amd import _hip  # noqa: F401
from flowconductor_amd.flows import Flow, MaskedAutoregressiveFlow  # noqa: F401

__all__ = ["Flow", "MaskedAutoregressiveFlow"]
__version__ = "0.1.0"
