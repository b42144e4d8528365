from flowconductor_amd.distributions.base import Distribution, NoMeanException  # noqa: F401
from flowconductor_amd.distributions.normal import (ConditionalDiagonalNormal, DiagonalNormal,  # noqa: F401
                                                     StandardNormal)
