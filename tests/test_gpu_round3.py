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
