"""The resident-weight instances of the general fused final-Linear + RQ-spline entry (fc_rq_fused4_body.h: hidden 64,
linear tails K = 4..7 and 9..11, no tails K = 4..10; K = 10 is the reference's default num_bins, coupling.py:507): operator level against a float64 Linear in
front of the oracle's spline, and against the streamed-weight kernel it replaces for this shape, over the kernel's
variants (32 / fewer transformed dims, padded / unpadded x rows, one / two float4 of x per thread, both directions,
running logabsdet total, more tiles than workgroups)."""
import pytest
import torch

from _util import maxdiff
from flowconductor_amd import ops
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _case(n, d, d_t, k, seed, h_decades=True, tails="linear"):
    torch.manual_seed(seed)
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    if tails == "linear":
        x = torch.randn(n, d) * 1.5
        x[1, :], x[2, :] = 3.0, -4.5                   # on and beyond the tail bound
    else:
        x = torch.rand(n, d)                           # the unit box
        x[1, :], x[2, :] = 0.0, 1.0                    # its ends
    h = torch.relu(torch.randn(n, 64)) * 2 + torch.randn(n, 64) * 0.3
    if h_decades:
        h *= torch.logspace(-3, 2, n).unsqueeze(1)     # row scales over five decades
    w = torch.randn(d_t * p, 64) * (1.2 / 8.0)
    w *= torch.logspace(-2, 0, d_t * p)[torch.randperm(d_t * p)].unsqueeze(1)
    b = torch.randn(d_t * p) * 0.3
    cols = torch.randperm(d)[:d_t].sort().values.to(torch.int32)
    return x, h, w, b, cols


def _reference(x, h, w, b, cols, k, inverse, tails="linear"):
    """float64 Linear + the oracle's spline in float64; the noise floor of the input: the same in float32 (f32 GEMM,
    f32 spline -- what the reference computes) against it."""
    n, d_t, p = x.shape[0], cols.numel(), w.shape[0] // cols.numel()
    rows64 = (h.double() @ w.double().T + b.double()).view(n, d_t, p).clone()
    rows32 = (h @ w.T + b).view(n, d_t, p).clone()
    xs = x[:, cols.long()]
    out, lad = O.rq_from_rows(xs.double(), rows64, k, tails, 3.0, inverse, wh_divisor=8.0)
    out32, lad32 = O.rq_from_rows(xs, rows32, k, tails, 3.0, inverse, wh_divisor=8.0)
    ref_y = x.double().clone()
    ref_y[:, cols.long()] = out
    return ref_y, lad.sum(dim=1), maxdiff(out32, out), maxdiff(lad32.sum(dim=1), lad.sum(dim=1))


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("n,d,d_t", [(256, 64, 32), (96, 64, 13), (160, 63, 32), (64, 37, 5), (128, 128, 32), (32, 126, 30),
                                     (32, 32, 32), (64, 2, 1)])
def test_resident_k10_against_float64_linear_and_streamed(n, d, d_t, inverse, device):
    _check_against_float64_and_streamed(10, n, d, d_t, inverse, device)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("k", [4, 5, 6, 7, 9, 11])
@pytest.mark.parametrize("n,d,d_t", [(96, 64, 32), (64, 37, 5), (32, 128, 32)])
def test_resident_other_bin_counts(k, n, d, d_t, inverse, device):
    _check_against_float64_and_streamed(k, n, d, d_t, inverse, device)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("k", [4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("n,d,d_t", [(96, 64, 32), (64, 37, 5), (32, 128, 32)])
def test_resident_box_form(k, n, d, d_t, inverse, device):
    """tails=None (the reference's default argument, coupling.py:508,543-547): 3K + 1 parameters, the unit box."""
    _check_against_float64_and_streamed(k, n, d, d_t, inverse, device, tails=None)


def test_resident_box_flags_only_transformed_columns(device):
    """Outside the box: InputOutsideDomain for a TRANSFORMED column (rational_quadratic.py:81-82), never for an identity
    column -- also when the transformed dims do not fill the last group of four (its spare lanes re-evaluate a
    transformed column, not column 0)."""
    from flowconductor_amd.transforms import InputOutsideDomain

    k, n, d, d_t = 10, 64, 9, 3
    x, h, w, b, _ = _case(n, d, d_t, k, seed=1, h_decades=False, tails=None)
    cols = torch.tensor([2, 5, 7], dtype=torch.int32)
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, None, 64)
    kw = dict(num_bins=k, tails=None, wh_divisor=8.0)
    x[:, 0] = 7.5                                      # identity column 0 far outside the box
    for streamed in (False, True):
        with torch.no_grad():
            y, _ = ops.rq_spline_fused_general(x.to(device), h.to(device), *packed, cols.to(device), streamed_weights=streamed, **kw)
        assert torch.equal(y[:, 0].cpu(), x[:, 0])
        bad = x.clone()
        bad[40, 5] = 1.5
        with pytest.raises(InputOutsideDomain):
            with torch.no_grad():
                ops.rq_spline_fused_general(bad.to(device), h.to(device), *packed, cols.to(device), streamed_weights=streamed, **kw)


def _check_against_float64_and_streamed(k, n, d, d_t, inverse, device, tails="linear"):
    # (inverse direction: rows of h at one scale -- with five decades of row scales the steepest splines make the
    #  inverse so ill-conditioned that the float32 oracle itself is off by more than any bound worth asserting)
    x, h, w, b, cols = _case(n, d, d_t, k, seed=n + d + d_t + k, h_decades=not inverse, tails=tails)
    ref_y, ref_lad, floor_y, floor_lad = _reference(x, h, w, b, cols, k, inverse, tails)
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, tails, 64)
    kw = dict(num_bins=k, tails=tails, tail_bound=3.0, wh_divisor=8.0, inverse=inverse)
    with torch.no_grad():
        y, lad = ops.rq_spline_fused_general(x.to(device), h.to(device), *packed, cols.to(device), **kw)
        ys, lads = ops.rq_spline_fused_general(x.to(device), h.to(device), *packed, cols.to(device), streamed_weights=True, **kw)
    scale, lscale = max(1.0, float(ref_y.abs().max())), max(1.0, float(ref_lad.abs().max()) / 10)
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()
    assert maxdiff(y, ref_y) <= 2e-5 * scale + 4 * floor_y
    assert maxdiff(lad, ref_lad) <= 2e-4 * lscale + 4 * floor_lad
    # the two kernels share the products' split but not the order of the spline arithmetic: each is within the bound
    # of the float64 reference, so they are within twice the bound of each other
    assert maxdiff(ys, ref_y) <= 2e-5 * scale + 4 * floor_y and maxdiff(lads, ref_lad) <= 2e-4 * lscale + 4 * floor_lad
    assert maxdiff(y, ys) <= 2 * (2e-5 * scale + 4 * floor_y) and maxdiff(lad, lads) <= 2 * (2e-4 * lscale + 4 * floor_lad)
    other = torch.ones(d, dtype=torch.bool)
    other[cols.long()] = False
    assert torch.equal(y.cpu()[:, other], x[:, other])          # identity columns pass through bit for bit


def test_resident_k10_many_tiles_and_running_total(device):
    """More 32-row tiles than workgroups (every workgroup walks its ring several times), logabsdet accumulated onto
    the caller's running total (transforms/base.py:51)."""
    k, n, d, d_t = 10, 1 << 15, 64, 32
    x, h, w, b, cols = _case(n, d, d_t, k, seed=5, h_decades=False)
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, "linear", 64)
    kw = dict(num_bins=k, tails="linear", tail_bound=3.0, wh_divisor=8.0)
    xd, hd, cd = x.to(device), h.to(device), cols.to(device)
    with torch.no_grad():
        y, lad = ops.rq_spline_fused_general(xd, hd, *packed, cd, **kw)
        ys, lads = ops.rq_spline_fused_general(xd, hd, *packed, cd, streamed_weights=True, **kw)
        total = torch.full((n,), 2.5, device=device)
        y2, lad2 = ops.rq_spline_fused_general(xd, hd, *packed, cd, logabsdet_accum=total, **kw)
        back, lad_inv = ops.rq_spline_fused_general(y, hd, *packed, cd, inverse=True, **kw)
    assert lad2.data_ptr() == total.data_ptr() and torch.equal(y2, y)
    assert maxdiff(lad2, lad + 2.5) <= 1e-5 * max(1.0, float(lad.abs().max()))
    assert maxdiff(y, ys) <= 2e-5 * max(1.0, float(ys.abs().max()))
    assert maxdiff(lad, lads) <= 5e-4 * max(1.0, float(lads.abs().max()) / 10)
    # sampled rows against the float64 reference
    rows = torch.arange(0, n, 61)
    ref_y, ref_lad, floor_y, floor_lad = _reference(x[rows], h[rows], w, b, cols, k, False)
    assert maxdiff(y.cpu()[rows], ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max())) + 4 * floor_y
    assert maxdiff(lad.cpu()[rows], ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * floor_lad
    inside = x.abs() <= 3.0
    assert float(((back.cpu() - x).abs() * inside).max()) <= 2e-3
    assert float((lad + lad_inv).abs().max()) <= 2e-2
