"""Round-4 regressions: advisor findings of round 3 (phantom rows of a partial 48-row tile, zero per-sample reflections,
fast-path predicates that outlive a change of the conditioner) and the one-launch fused backward's extra shapes."""
import pytest
import torch

from _util import Lib, maxdiff
from flowconductor_amd import ops
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
T, nets, utils = Lib.transforms, Lib.nets, Lib.utils


@pytest.mark.parametrize("hidden", [128, 256])
def test_partial_tile_phantom_rows_do_not_raise_outside_domain(hidden, device):
    """tails=None, hidden 128 / 256 (48-row tiles), a batch that is not a whole number of 48-row tiles: the rows beyond the
    batch in the last tile are phantom copies of that tile's first 16-byte piece.  An IDENTITY feature outside the box in it
    (the reference never range-checks identity features, coupling.py:73-100) must not raise; a transformed one must."""
    from flowconductor_amd.transforms import InputOutsideDomain

    torch.manual_seed(hidden)
    k, n, d, d_t = 10, 160, 12, 4          # 160 = 3 x 48 + 16: the last tile holds 16 real rows
    p = 3 * k + 1
    x = torch.rand(n, d)
    h = torch.randn(n, hidden)
    w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
    b = torch.randn(d_t * p) * 0.3
    cols = torch.tensor([4, 6, 9, 11], dtype=torch.int32)          # columns 0..3 (the tile's first piece) are identity features
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, None, hidden)
    kw = dict(num_bins=k, tails=None, wh_divisor=float(hidden) ** 0.5)
    x[144, 0:4] = torch.tensor([7.5, -3.0, 1.5, 2.0])              # first row of the last tile, identity columns
    with torch.no_grad():
        y, _ = ops.rq_spline_fused_general(x.to(device), h.to(device), *packed, cols.to(device), **kw)
    assert torch.equal(y[:, :4].cpu(), x[:, :4])
    bad = x.clone()
    bad[150, 6] = 1.25
    with pytest.raises(InputOutsideDomain):
        with torch.no_grad():
            ops.rq_spline_fused_general(bad.to(device), h.to(device), *packed, cols.to(device), **kw)


def test_per_sample_householder_with_zero_reflections_is_the_identity(device):
    x = torch.randn(70, 24)
    q = torch.zeros(70, 0, 24)
    with torch.no_grad():
        y = ops.householder(x.to(device), q.to(device))
        y_rev = ops.householder(x.to(device), q.to(device), reverse=True)
    assert torch.equal(y.cpu(), x) and torch.equal(y_rev.cpu(), x)


def test_one_kernel_predicate_follows_the_conditioner(device):
    """The memoised structure predicate of the one-kernel affine coupling layer re-derives itself when the conditioner
    changes under it: dropout switched on in training mode must leave the fused path (the kernel has no dropout)."""
    torch.manual_seed(3)
    mask = utils.create_alternating_binary_mask(16, even=True)
    t = T.AffineCouplingTransform(mask, lambda i, o: nets.ResidualNet(i, o, hidden_features=32, num_blocks=2)).to(device).eval()
    x = torch.randn(256, 16, device=device)
    with torch.no_grad():
        assert t._one_kernel_ok(x, None)
        y0, _ = t(x)
        for blk in t.transform_net.blocks:
            blk.dropout.p = 0.5
        t.transform_net.train()
        assert not t._one_kernel_ok(x, None)
        t.transform_net.eval()
        for blk in t.transform_net.blocks:
            blk.dropout.p = 0.0
        y1, _ = t(x)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("k,tails,d,d_t,n", [(8, "linear", 64, 20, 96), (9, "linear", 40, 17, 64), (11, "linear", 128, 32, 64),
                                            (8, None, 24, 9, 128), (6, "linear", 8, 2, 32)])
def test_one_launch_backward_odd_group_counts(k, tails, d, d_t, n, device):
    """fc_rq_fused_linear_backward (one launch, sweeps of four dim groups): dim counts that leave waves without a group in
    the second sweep, a last group of fewer than four dims, D = 128, the widest parameter rows (K = 11: 32 per dim)."""
    torch.manual_seed(11 * k + d_t)
    hidden = 64
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    x = torch.rand(n, d) if tails is None else torch.randn(n, d) * 1.5
    h = torch.relu(torch.randn(n, hidden)) + torch.randn(n, hidden) * 0.2
    w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
    b = torch.randn(d_t * p) * 0.3
    cols = torch.randperm(d)[:d_t].sort().values.to(torch.int32)
    gy, gl = torch.randn(n, d), torch.randn(n)
    kw = dict(wh_divisor=float(hidden) ** 0.5)
    x64, h64, w64, b64 = (v.double().requires_grad_(True) for v in (x, h, w, b))
    rows = (h64 @ w64.T + b64).view(n, d_t, p)
    out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
    y64 = x64.clone().index_copy(1, cols.long(), out)
    loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
    refs = torch.autograd.grad(loss, (x64, h64, w64, b64))
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, tails, 64)
    packed_t = ops.pack_final_layer_transposed(w.to(device), k, tails)
    got = ops.rq_fused_linear_backward(x.to(device), h.to(device), gy.to(device), gl.to(device), packed, packed_t,
                                       cols.to(device), num_bins=k, tails=tails, tail_bound=3.0, **kw)
    for g, r in zip(got, refs):
        assert float((g.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)) <= 2e-4
    # deterministic parts: bit-identical between two launches
    again = ops.rq_fused_linear_backward(x.to(device), h.to(device), gy.to(device), gl.to(device), packed, packed_t,
                                         cols.to(device), num_bins=k, tails=tails, tail_bound=3.0, **kw)
    assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])


def test_softplus_on_a_4d_batch_against_the_reference_fixture(device):
    """The reference's ``Softplus`` sums its log-Jacobian over the LAST dim only (nonlinearities.py:182,188): on a 4-D batch
    its logabsdet has shape [N, C, H].  This package returns [N] (the sum over all non-batch dims, what every other
    transform of the reference does and what CompositeTransform adds up): outputs identical to the reference fixture
    (tests/golden/fn_softplus_4d.npz, generated by importing the reference), logabsdet = the fixture's summed over C, H."""
    import os

    import numpy as np

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "fn_softplus_4d.npz"))
    t = T.Softplus().to(device)
    x = torch.from_numpy(fx["x"]).to(device)
    with torch.no_grad():
        y, lad = t(x)
        xi, ladi = t.inverse(torch.from_numpy(fx["y"]).to(device))
    assert lad.shape == (5,) and ladi.shape == (5,)
    assert maxdiff(y, torch.from_numpy(fx["y"])) <= 2e-6
    assert maxdiff(lad, torch.from_numpy(fx["lad"]).sum(dim=(1, 2))) <= 2e-5
    assert maxdiff(xi, torch.from_numpy(fx["xinv"])) <= 2e-5
    assert maxdiff(ladi, torch.from_numpy(fx["ladinv"]).sum(dim=(1, 2))) <= 2e-4


# ---- vector paths added late in round 4: the same answers as the one-element-per-lane forms ------------------------------
@pytest.mark.parametrize("d", [4, 12, 64, 100, 256, 30])
@pytest.mark.parametrize("per_sample", [False, True])
def test_planar_sub_wave_rows_match_float64(d, per_sample, device):
    """fc_planar on rows of 4k floats runs on sub-wave lane groups with float4 pieces (d = 30: the row-per-wave kernel);
    both against planar.py:30-49 in float64, shared and per-sample parameters, batches that do not fill the last wave."""
    from flowconductor_amd import ops

    torch.manual_seed(d)
    n = 1000 + 3
    x = torch.randn(n, d)
    rows = n if per_sample else 1
    w, u, b = torch.randn(rows, d) * 0.3, torch.randn(rows, d) * 0.3, torch.randn(rows) * 0.2
    y, lad = ops.planar(x.to(device), w.to(device), u.to(device), b.to(device), per_sample=per_sample)
    x64, w64, u64, b64 = x.double(), w.double(), u.double(), b.double()
    a = (x64 * w64).sum(-1) + b64
    t = torch.tanh(a)
    ref_y = x64 + u64 * t.unsqueeze(-1)
    ref_lad = torch.log(1e-7 + (1 + (1 - t ** 2) * (u64 * w64).sum(-1)).abs())
    a32 = (x * w).sum(-1) + b                     # the float32 sequence of the reference: the noise floor where |1 + s| is small
    t32 = torch.tanh(a32)
    lad32 = torch.log(1e-7 + (1 + (1 - t32 ** 2) * (u * w).sum(-1)).abs())
    assert maxdiff(y.cpu().double(), ref_y) <= 2e-6 * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad.cpu().double(), ref_lad) <= 2e-5 + 4 * maxdiff(lad32.double(), ref_lad)


@pytest.mark.parametrize("kind", ["exp", "tanh", "sigmoid", "leaky", "softplus", "cauchy", "logtanh"])
def test_elementwise_vector_rows_match_scalar_rows(kind, device):
    """fc_elementwise moves rows of 4k elements as 16-byte pieces; a view that starts 4 bytes into its buffer takes the
    one-element-per-lane kernel: same outputs bit for bit, row sums equal to the order of the additions."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(3)
    t = {"exp": T.Exp, "tanh": T.Tanh, "sigmoid": T.Sigmoid, "leaky": T.LeakyReLU, "softplus": T.Softplus,
         "cauchy": T.CauchyCDF, "logtanh": T.LogTanh}[kind]().to(device)
    n, m = 777, 64
    flat = torch.randn(n * m + 1, device=device) * 1.5
    aligned = flat[:n * m].reshape(n, m).clone()
    shifted = flat[1:].reshape(n, m)
    shifted.copy_(aligned)
    assert aligned.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    with torch.no_grad():
        y_v, lad_v = t(aligned)
        y_s, lad_s = t(shifted)
    assert torch.equal(y_v, y_s)
    assert maxdiff(lad_v, lad_s) <= 1e-5 * max(1.0, float(lad_s.abs().max()))


def test_masked_conditioner_keeps_small_columns_beside_a_large_one(device):
    """fc_split.h lifts a row's maximum to [2^14, 2^15): the two f16 pieces keep 22 bits of entries down to 2^-16 of the row
    maximum and lose one bit per further factor of two.  A MADE unit that reads only the small columns must see them beside a
    column 2^20 times larger that it does not read (the reference's masked float32 GEMM multiplies that column by an exact
    zero): 18 bits here; with the maximum at [2^10, 2^11) (rounds 1-3) 14 bits -- 6e-5 relative, outside this tolerance."""
    import copy

    from flowconductor_amd import transforms as T

    torch.manual_seed(11)
    d = 12
    t = T.MaskedAffineAutoregressiveTransform(d, 64, num_blocks=2).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = torch.randn(2048, d)
    x[:, d - 1] *= 2.0 ** 20          # the LAST column: no parameter of dims 0 .. d-2 may depend on it
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(copy.deepcopy(t).double(), x.double())
        f32_y, f32_lad = O.transform_apply(t, x)
        y, lad = t.to(device)(x.to(device))
    head = slice(0, d - 1)
    tol = 2e-5 * max(1.0, float(ref_y[:, head].abs().max())) + 4 * maxdiff(f32_y[:, head].double(), ref_y[:, head])
    assert maxdiff(y[:, head].cpu().double(), ref_y[:, head]) <= tol
    # (the last dim's own scale enters logabsdet at O(1): still float32-class)
    assert maxdiff(lad.cpu().double(), ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * maxdiff(f32_lad.double(), ref_lad)
