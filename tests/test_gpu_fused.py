"""Final-conditioner-layer + RQ-spline fused kernel (fc_rq_spline_fused_linear) vs the unfused HIP path
and the CPU oracle."""
import os

import pytest
import torch

from _util import build_case, maxdiff
from flowconductor_amd import ops
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [32, 64, 95, 4096, 100003])
@pytest.mark.parametrize("inverse", [False, True])
def test_fused_matches_unfused_and_oracle(n, inverse, device, monkeypatch):
    t, _ = build_case("rq_coupling_linear_tails_d64_k8_h64")
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, 64, generator=gen) * 1.5
    rows = min(n, 512)
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x[:rows].clone(), inverse=inverse)
    t = t.to(device)
    xd = x.to(device)
    fn = t.inverse if inverse else t.forward
    with torch.no_grad():
        assert t._fused_ok(xd)
        with ops.KernelTimer("fc_rq_spline_fused_linear") as timer:
            y_f, lad_f = fn(xd)
        assert len(timer.pairs) == 1, "the fused kernel did not run"
        monkeypatch.setenv("FC_FUSED", "0")
        assert not t._fused_ok(xd)
        y_u, lad_u = fn(xd)
    scale = max(1.0, float(ref_y.abs().max()))
    # fused vs unfused: same spline arithmetic, GEMM by exact-f32 MFMA vs hipBLASLt
    assert maxdiff(y_f, y_u) <= 5e-5 * scale
    # (the spline inverse is ill-conditioned on a few elements: tools/noise_floor.py, max 1.9e-3 at 2^14 rows)
    assert maxdiff(lad_f, lad_u) <= (1e-3 if not inverse else 1e-2)
    tol_y, tol_l = (2e-5, 2e-4) if not inverse else (3e-4, 3e-3)
    assert maxdiff(y_f[:rows], ref_y) <= tol_y * scale
    assert maxdiff(lad_f[:rows], ref_lad) <= tol_l * max(1.0, float(ref_lad.abs().max()) / 10)
    assert torch.isfinite(y_f).all() and torch.isfinite(lad_f).all()




@pytest.mark.parametrize("in_f,blocks,d,n", [(32, 2, 64, 64), (32, 2, 64, 6400), (16, 1, 32, 128), (6, 0, 12, 64),
                                            (64, 2, 128, 192)])
def test_resnet_hidden_kernel_matches_torch(in_f, blocks, d, n, device):
    """fc_resnet_hidden vs the same nn.Module evaluated by PyTorch on the CPU."""
    from flowconductor_amd.nn import nets

    torch.manual_seed(in_f + blocks)
    net = nets.ResidualNet(in_f, 8, hidden_features=64, num_blocks=blocks).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)  # make the residual blocks matter (their last layer is initialised ~1e-3)
    ids = torch.randperm(d)[:in_f].sort().values
    x = torch.randn(n, d)
    with torch.no_grad():
        ref = net.hidden(x[:, ids])
        assert net.hip_hidden_supported(d)
        got = net.to(device).hidden_hip(x.to(device), ids.to(device))
    assert got.shape == (n, 64)
    assert maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_fused_hidden_path_in_coupling(device, monkeypatch):
    t, _ = build_case("rq_coupling_linear_tails_d64_k8_h64")
    x = torch.randn(1000, 64) * 1.5
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone())
    t = t.to(device)
    with torch.no_grad():
        with ops.KernelTimer("fc_resnet_hidden") as timer:
            y, lad = t(x.to(device))
        assert len(timer.pairs) == 1, "the hidden-layer kernel did not run"
        monkeypatch.setenv("FC_FUSED_HIDDEN", "0")
        y2, lad2 = t(x.to(device))
    assert maxdiff(y, ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad, ref_lad) <= 3e-4
    assert maxdiff(y, y2) <= 2e-5 and maxdiff(lad, lad2) <= 3e-4
