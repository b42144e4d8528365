"""Round-3 regressions on the GPU: module copies / checkpoints after a forward pass, the fused layer's autograd node
refusing stale weights, the product's InverseTransform."""
import copy
import io

import numpy as np
import pytest
import torch

from _util import GOLDEN_DIR, Lib, build_case, golden  # noqa: F401

pytestmark = pytest.mark.gpu
T, nets, utils = Lib.transforms, Lib.nets, Lib.utils


def _nsf(d=64, hidden=64, layers=2, bins=8):
    torch.manual_seed(5)
    stack = T.CompositeTransform([
        T.PiecewiseRationalQuadraticCouplingTransform(
            utils.create_alternating_binary_mask(d, even=(l % 2 == 0)),
            lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=2),
            num_bins=bins, tails="linear", tail_bound=3.0) for l in range(layers)])
    return Lib.flows.Flow(stack, Lib.distributions.StandardNormal([d]))


def test_deepcopy_and_torch_save_after_forward_and_backward(device):
    """DevicePack plans on the modules broke copy.deepcopy / torch.save once a flow had run."""
    flow = _nsf().to(device)
    x = torch.randn(256, 64, device=device)
    flow.train()
    (-flow.log_prob(x).mean()).backward()             # builds the training plans (_train_pack, _hip_packed_bwd)
    flow.eval()
    with torch.no_grad():
        ref = flow.log_prob(x)                         # builds the inference images (_hip_image)
    snap = copy.deepcopy(flow)
    buf = io.BytesIO()
    torch.save(flow, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    with torch.no_grad():
        assert torch.equal(snap.log_prob(x), ref)
        assert torch.equal(loaded.log_prob(x), ref)
        # the copy packs from ITS OWN parameters: changing the original must not leak into it
        for p in flow.parameters():
            p.mul_(1.5)
        assert torch.equal(snap.log_prob(x), ref)
        assert not torch.equal(flow.log_prob(x), ref)


def test_fused_backward_refuses_weights_changed_after_forward(device):
    """the fused autograd node reads the live weights in backward; an in-place update between forward
    and backward must raise like the plain torch graph would, not return silently wrong gradients."""
    flow = _nsf(layers=1).to(device).train()
    x = torch.randn(128, 64, device=device)
    loss = -flow.log_prob(x).mean()
    with torch.no_grad():
        flow._transform._transforms[0].transform_net.final_layer.weight.add_(0.01)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
    flow.zero_grad()
    (-flow.log_prob(x).mean()).backward()              # an untouched forward/backward pair still works
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in flow.parameters())


INVERSE_CASES = ["rq_coupling_linear_tails_d8_k8", "affine_coupling_d32", "maf_affine_d12_h32_ctx", "lu_linear_d9",
                 "householder_sequence_d16_k6", "reverse_permutation_d64", "rq_coupling_linear_tails_d64_k8_h64", "maf_rq_linear_tails_d6_k8", "rq_nsf_stack_d16_l4"]


@pytest.mark.parametrize("name", INVERSE_CASES)
def test_inverse_transform_wrapper_swaps_directions(device, name):
    """A3 (flowcon/transforms/base.py:215-231): forward of InverseTransform(t) is the reference's INVERSE golden,
    inverse of the wrapper is the reference's FORWARD golden -- through the product classes on the GPU."""
    from test_gpu_golden import _check

    g = golden(name)
    t, spec = build_case(name, g)
    assert spec["inverse"], name
    t = t.to(device)
    w = T.InverseTransform(t)
    exact = spec["tol"][0] == 0
    for n in (7, 257):
        ctx = torch.from_numpy(g["ctx_%d" % n]).to(device) if spec["context"] else None
        yin = torch.from_numpy(g["yin_%d" % n]).to(device)
        x = torch.from_numpy(g["x_%d" % n]).to(device)
        with torch.no_grad():
            out, lad = w(yin, ctx)
            back, lad_b = w.inverse(x, ctx)
        assert out.shape == yin.shape and lad.shape == (n,)
        _check(name, n, "InverseTransform.forward", out, g, "xinv", exact, 4.0)
        _check(name, n, "InverseTransform.forward logabsdet", lad, g, "ladinv", exact, 4.0)
        _check(name, n, "InverseTransform.inverse", back, g, "y", exact, 4.0)
        _check(name, n, "InverseTransform.inverse logabsdet", lad_b, g, "lad", exact, 4.0)
    # composition: InverseTransform inside a CompositeTransform runs the wrapped layers backwards
    comp = T.CompositeTransform([w, t])
    with torch.no_grad():
        z, total = comp(yin, ctx)
    scale = max(1.0, float(yin.abs().max()))
    assert (z - yin).abs().max().item() <= 2e-4 * scale
    assert total.abs().max().item() <= 5e-3


def test_inverse_transform_of_a_transform_without_inverse_raises(device):
    t = T.InverseTransform(T.PlanarTransform(features=4)).to(device)
    with pytest.raises(T.InverseNotAvailable):
        t(torch.zeros(3, 4, device=device))


def test_per_sample_sylvester_full_size_2_18(device):
    """BASELINE.json configs[4] in its conditional (per-sample) form at FULL size: x [2^18, 128], q [2^18, 32, 128],
    R1 / R2 [2^18, 128, 128] (38 GB of per-sample parameters).  512 rows spread over the batch against the float64
    formula of conditional.py:936-953; on ALL rows: finite, deterministic, and a row's result independent of where in
    the launch it sits (the same rows as a 512-row launch, bitwise)."""
    from flowconductor_amd import ops

    n, d, m = 1 << 18, 128, 32
    gen = torch.Generator(device=device).manual_seed(2718)
    x = torch.randn(n, d, device=device, generator=gen)
    q = torch.randn(n, m, d, device=device, generator=gen)
    r1 = torch.randn(n, d, d, device=device, generator=gen).mul_(1.0 / d ** 0.5)
    r2 = torch.randn(n, d, d, device=device, generator=gen).mul_(1.0 / d ** 0.5)
    r1.diagonal(dim1=1, dim2=2).tanh_()
    r2.diagonal(dim1=1, dim2=2).tanh_()        # (below the diagonal: random junk, never read)
    bias = torch.randn(n, d, device=device, generator=gen).mul_(0.1)
    with torch.no_grad():
        y, lad = ops.sylvester(x, q, r1, r2, bias)
        y2, lad2 = ops.sylvester(x, q, r1, r2, bias)
    assert y.shape == (n, d) and lad.shape == (n,)
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()
    assert torch.equal(y, y2) and torch.equal(lad, lad2)
    idx = torch.cat((torch.arange(0, 128), torch.arange(n // 3 + 5, n // 3 + 5 + 128), torch.arange(n // 2 - 64, n // 2 + 64),
                     torch.arange(n - 128, n))).to(device)
    with torch.no_grad():
        ys, lads = ops.sylvester(x[idx], q[idx].contiguous(), r1[idx].contiguous(), r2[idx].contiguous(), bias[idx])
    assert torch.equal(y[idx], ys) and torch.equal(lad[idx], lads)
    xd, qd = x[idx].double().cpu(), q[idx].double().cpu()
    r1d, r2d, bd = torch.triu(r1[idx].double().cpu()), torch.triu(r2[idx].double().cpu()), bias[idx].double().cpu()

    def reflect(v, reverse):
        for i in (range(m - 1, -1, -1) if reverse else range(m)):
            qi = qd[:, i]
            v = v - (v * qi).sum(-1, keepdim=True) * (2.0 / (qi * qi).sum(-1, keepdim=True)) * qi
        return v

    act = torch.tanh(torch.einsum("nij,nj->ni", r1d, reflect(xd, True)) + bd)
    ref_y = xd + reflect(torch.einsum("nij,nj->ni", r2d, act), False)
    ref_lad = torch.log(1 + (1 - act ** 2) * (r1d.diagonal(dim1=1, dim2=2) * r2d.diagonal(dim1=1, dim2=2))).sum(-1)
    assert float((ys.double().cpu() - ref_y).abs().max()) <= 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert float((lads.double().cpu() - ref_lad).abs().max()) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10)


def test_scalar_scale_and_shift(device):
    """flowcon/transforms/linear.py:232-266, including the reference's log-determinant factor (the SUM of the non-batch
    sizes) and gradients of the scalar parameters."""
    s, b = T.ScalarScale(scale=3.0, trainable=True).to(device), T.ScalarShift(0.25, trainable=True).to(device)
    x = torch.randn(9, 5, device=device)
    with torch.no_grad():
        y, lad = s(x)
        xb, ladb = s.inverse(y)
        z, lz = b(x)
        zb, _ = b.inverse(z)
    scale = float(torch.exp(s._scale) + s.eps)
    assert torch.allclose(y, x * scale, rtol=1e-6) and torch.allclose(lad, torch.full((9,), 5 * np.log(scale), device=device), rtol=1e-6)
    assert torch.allclose(xb, x, atol=1e-6) and torch.allclose(ladb, -lad)
    assert torch.allclose(z, x + 0.25) and float(lz.abs().max()) == 0.0 and torch.allclose(zb, x, atol=1e-6)
    x4 = torch.randn(3, 2, 4, 5, device=device)
    with torch.no_grad():
        _, lad4 = s(x4)
    assert torch.allclose(lad4, torch.full((3,), (2 + 4 + 5) * np.log(scale), device=device), rtol=1e-6)    # sum, not product
    (s(x)[0].sum() + s(x)[1].sum() + b(x)[0].sum()).backward()
    assert s._scale.grad is not None and b.shift.grad is not None
    assert abs(float(b.shift.grad) - x.numel()) < 1e-3


@pytest.mark.parametrize("hidden,bins", [(64, 8), (32, 10), (128, 10)])
def test_paranoid_caches_see_data_writes_without_invalidation(device, hidden, bins):
    """with options.paranoid_caches an eval-mode model whose weights are edited through `.data` gives
    fresh results WITHOUT ops.invalidate_hip_caches() -- on the hand-scheduled K = 8 path, the packed-fragment general
    path and the wide hidden stack; the default (off) keeps the documented caveat."""
    from flowconductor_amd import options

    torch.manual_seed(9)
    d = 16
    t = T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d), lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=2),
        num_bins=bins, tails="linear", tail_bound=3.0).eval().to(device)
    x = torch.randn(256, d, device=device)
    with torch.no_grad(), options.override(paranoid_caches=True):
        y0, _ = t(x)
        for p in t.transform_net.parameters():
            p.data.mul_(1.5)                       # EMA-style swap; no invalidate_hip_caches()
        y1, lad1 = t(x)
        with options.override(fused_final_layer=False, fused_hidden=False):
            y_ref, lad_ref = t(x)                  # conditioner on PyTorch kernels: always the live weights
    assert float((y0 - y_ref).abs().max()) > 1e-3, "the weight change must be visible"
    assert float((y1 - y_ref).abs().max()) <= 2e-5 and float((lad1 - lad_ref).abs().max()) <= 3e-4


@pytest.mark.parametrize("features,hidden,blocks,n", [(2, 4, 2, 4096), (12, 32, 2, 257), (32, 64, 3, 1000), (7, 50, 0, 64)])
def test_maf_affine_density_direction_in_one_kernel(device, features, hidden, blocks, n):
    """Round 3: the forward (density) direction of MaskedAffineAutoregressiveTransform runs hidden stack + masked final
    Linear + bijector in fc_affine_coupling_resnet on pre-masked weights (README flow: BASELINE.json configs[0]).
    Against the oracle, against the three-kernel path, with leftover rows, inside a CompositeTransform (in-kernel
    logabsdet accumulation), and after a weight update."""
    from flowconductor_amd import ops, options
    from oracle import torch_oracle as O

    torch.manual_seed(features + hidden)
    t_cpu = T.MaskedAffineAutoregressiveTransform(features=features, hidden_features=hidden, num_blocks=blocks).eval()
    with torch.no_grad():
        for p in t_cpu.parameters():
            p.mul_(1.5)
    t = copy.deepcopy(t_cpu).to(device)
    x = torch.randn(n, features)
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(t_cpu, x.clone())
        with ops.KernelTimer("fc_affine_coupling_resnet") as one, ops.KernelTimer("fc_affine") as plain:
            y, lad = t(x.to(device))
        assert len(one.pairs) == 1 and len(plain.pairs) == (1 if n % 16 else 0)
        with options.override(fused_final_layer=False):
            y3, lad3 = t(x.to(device))
        comp = T.CompositeTransform([t, T.ReversePermutation(features), t])
        z, total = comp(x.to(device))
        z_ref, total_ref = O.transform_apply(T.CompositeTransform([t_cpu, T.ReversePermutation(features), t_cpu]), x.clone())
        xb, lad_b = t.inverse(y)
    scale = max(1.0, float(y_ref.abs().max()))
    assert float((y.cpu() - y_ref).abs().max()) <= 2e-5 * scale and float((lad.cpu() - lad_ref).abs().max()) <= 1e-4
    assert float((y - y3).abs().max()) <= 2e-5 * scale and float((lad - lad3).abs().max()) <= 1e-4
    assert float((z.cpu() - z_ref).abs().max()) <= 1e-4 * max(1.0, float(z_ref.abs().max()))
    assert float((total.cpu() - total_ref).abs().max()) <= 3e-4
    assert float((xb - x.to(device)).abs().max()) <= 1e-3 and float((lad + lad_b).abs().max()) <= 1e-3
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(0.5)                            # in-place update: version counters move, the image is re-packed
        for p in t_cpu.parameters():
            p.mul_(0.5)
        y2, _ = t(x.to(device))
        y2_ref, _ = O.transform_apply(t_cpu, x.clone())
    assert float((y2.cpu() - y2_ref).abs().max()) <= 2e-5 * max(1.0, float(y2_ref.abs().max()))
