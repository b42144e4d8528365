from flowconductor_amd.transforms.autoregressive import (  # noqa: F401
    AutoregressiveTransform,
    MaskedAffineAutoregressiveTransform,
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform,
    MaskedShiftAutoregressiveTransform,
)
from flowconductor_amd.transforms.base import (  # noqa: F401
    CompositeTransform,
    InputOutsideDomain,
    InverseNotAvailable,
    InverseTransform,
    Transform,
)
from flowconductor_amd.transforms.coupling import (  # noqa: F401
    AdditiveCouplingTransform,
    AffineCouplingTransform,
    CouplingTransform,
    PiecewiseRationalQuadraticCouplingTransform,
)
from flowconductor_amd.transforms.nonlinearities import (  # noqa: F401
    CompositeCDFTransform,
    PiecewiseRationalQuadraticCDF,
)
from flowconductor_amd.transforms.permutations import (  # noqa: F401
    Permutation,
    RandomPermutation,
    ReversePermutation,
)
from flowconductor_amd.transforms.standard import (  # noqa: F401
    AffineScalarTransform,
    AffineTransform,
    IdentityTransform,
    PointwiseAffineTransform,
)
