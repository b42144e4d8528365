from flowconductor_amd.transforms.no_analytic_inv.base import MonotonicTransform  # noqa: F401
from flowconductor_amd.transforms.no_analytic_inv.planar import (  # noqa: F401
    PlanarTransform,
    SylvesterTransform,
)
