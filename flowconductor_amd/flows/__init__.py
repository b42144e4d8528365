from flowconductor_amd.flows.autoregressive import MaskedAutoregressiveFlow  # noqa: F401
from flowconductor_amd.flows.base import Flow  # noqa: F401
from flowconductor_amd.flows.realnvp import SimpleRealNVP  # noqa: F401

__all__ = ["Flow", "MaskedAutoregressiveFlow", "SimpleRealNVP"]
