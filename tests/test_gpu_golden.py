"""GPU parity: flowconductor_amd (HIP kernels through the C ABI) vs the reference's golden
vectors and vs the CPU oracle, on identical weights and inputs."""
import pytest
import torch

import cases
from _util import SIZES, build_case, golden, maxdiff, rel_close
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_gpu_matches_reference_golden(name, device):
    g = golden(name)
    t, spec = build_case(name, g)
    t = t.to(device)
    tol_y, tol_lad, tol_xi, tol_ladi = spec["tol"]
    for n in SIZES:
        x = torch.from_numpy(g["x_%d" % n]).to(device)
        ctx = torch.from_numpy(g["ctx_%d" % n]).to(device) if spec["context"] else None
        with torch.no_grad():
            y, lad = t(x, ctx)
        assert y.shape == x.shape and lad.shape == (n,)
        assert torch.isfinite(y).all() and torch.isfinite(lad).all()
        if tol_y == 0:
            assert torch.equal(y.cpu(), torch.from_numpy(g["y_%d" % n])), (name, n)
            assert torch.equal(lad.cpu(), torch.from_numpy(g["lad_%d" % n])), (name, n)
        else:
            ok, worst = rel_close(y, g["y_%d" % n], rtol=1e-5, atol=tol_y)
            assert ok, (name, n, "outputs", worst, maxdiff(y, g["y_%d" % n]))
            ok, worst = rel_close(lad, g["lad_%d" % n], rtol=1e-5, atol=tol_lad)
            assert ok, (name, n, "logabsdet", worst, maxdiff(lad, g["lad_%d" % n]))
        if spec["inverse"]:
            yin = torch.from_numpy(g["y_%d" % n]).to(device)
            with torch.no_grad():
                xi, ladi = t.inverse(yin, ctx)
            if tol_xi == 0:
                assert torch.equal(xi.cpu(), torch.from_numpy(g["xinv_%d" % n])), (name, n)
            else:
                ok, worst = rel_close(xi, g["xinv_%d" % n], rtol=1e-5, atol=tol_xi)
                assert ok, (name, n, "inverse outputs", worst, maxdiff(xi, g["xinv_%d" % n]))
                ok, worst = rel_close(ladi, g["ladinv_%d" % n], rtol=1e-5, atol=tol_ladi)
                assert ok, (name, n, "inverse logabsdet", worst, maxdiff(ladi, g["ladinv_%d" % n]))


@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 257, 4099])
def test_rq_coupling_ragged_batches_vs_oracle(n, device):
    """Empty, tiny and non-multiple-of-tile batches through the north-star kernel."""
    name = "rq_coupling_linear_tails_d64_k8_h64"
    t, spec = build_case(name)
    gen = torch.Generator().manual_seed(n + 1)
    x = torch.randn(n, 64, generator=gen) * 1.5
    with torch.no_grad():
        if n > 0:
            y_ref, lad_ref = O.transform_apply(t, x.clone())
        y, lad = t.to(device)(x.to(device))
    assert y.shape == (n, 64) and lad.shape == (n,)
    if n > 0:
        ok, worst = rel_close(y, y_ref, rtol=1e-5, atol=2e-5)
        assert ok, worst
        ok, worst = rel_close(lad, lad_ref, rtol=1e-5, atol=1e-4)
        assert ok, worst


def test_cpu_inputs_fail_loudly(device):
    t, _ = build_case("affine_coupling_d32")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        with torch.no_grad():
            t(torch.randn(4, 32))


def test_rq_no_tails_raises_outside_domain(device):
    from flowconductor_amd.transforms import InputOutsideDomain

    t, _ = build_case("rq_coupling_no_tails_d6_k10")
    t = t.to(device)
    x = torch.rand(16, 6, device=device)
    x[3, 1] = 1.5  # a transformed column (odd index) outside [0, 1]
    with pytest.raises(InputOutsideDomain):
        with torch.no_grad():
            t(x)
    # the error word is cleared: a valid call afterwards succeeds
    with torch.no_grad():
        t(torch.rand(16, 6, device=device))
