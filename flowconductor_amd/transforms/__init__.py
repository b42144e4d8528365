from flowconductor_amd.transforms.autoregressive import (  # noqa: F401
    AutoregressiveTransform,
    MaskedAffineAutoregressiveTransform,
    MaskedPiecewiseCubicAutoregressiveTransform,
    MaskedPiecewiseLinearAutoregressiveTransform,
    MaskedPiecewiseQuadraticAutoregressiveTransform,
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform,
    MaskedShiftAutoregressiveTransform,
    MaskedSumOfSigmoidsTransform,
)
from flowconductor_amd.transforms.base import (  # noqa: F401
    CompositeTransform,
    InputOutsideDomain,
    InverseNotAvailable,
    InverseTransform,
    Transform,
)
from flowconductor_amd.transforms.conditional import (  # noqa: F401
    AffineConditionalTransform,
    ConditionalLUTransform,
    ConditionalOrthogonalTransform,
    ConditionalPiecewiseRationalQuadraticTransform,
    ConditionalPlanarTransform,
    ConditionalRotationTransform,
    ConditionalScaleTransform,
    ConditionalShiftTransform,
    ConditionalSumOfSigmoidsTransform,
    ConditionalSVDTransform,
    ConditionalSylvesterTransform,
    ConditionalTransform,
    PiecewiseLinearConditionalTransform,
)
from flowconductor_amd.transforms.coupling import (  # noqa: F401
    AdditiveCouplingTransform,
    AffineCouplingTransform,
    CouplingTransform,
    PiecewiseCubicCouplingTransform,
    PiecewiseLinearCouplingTransform,
    PiecewiseQuadraticCouplingTransform,
    PiecewiseRationalQuadraticCouplingTransform,
)
from flowconductor_amd.transforms.adaptive_sigmoids import SumOfSigmoids  # noqa: F401
from flowconductor_amd.transforms.linear import Linear, ScalarScale, ScalarShift  # noqa: F401
from flowconductor_amd.transforms.lu import LULinear  # noqa: F401
from flowconductor_amd.transforms.no_analytic_inv import (  # noqa: F401
    MonotonicTransform,
    PlanarTransform,
    SylvesterTransform,
)
from flowconductor_amd.transforms.nonlinearities import (  # noqa: F401
    CauchyCDF,
    CauchyCDFInverse,
    CompositeCDFTransform,
    Exp,
    ExtendedSoftplus,
    GatedLinearUnit,
    LeakyReLU,
    Logit,
    LogTanh,
    PiecewiseCubicCDF,
    PiecewiseLinearCDF,
    PiecewiseQuadraticCDF,
    PiecewiseRationalQuadraticCDF,
    Sigmoid,
    Softplus,
    Tanh,
)
from flowconductor_amd.transforms.normalization import ActNorm, BatchNorm  # noqa: F401
from flowconductor_amd.transforms.orthogonal import HouseholderSequence, ParametrizedHouseHolder  # noqa: F401
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
