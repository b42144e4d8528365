"""Column-at-a-time inverse of the masked autoregressive transforms against the reference's
D-full-passes scheme (autoregressive.py:44-53), restated by the oracle and by this package with
options ar_incremental = "off"."""
import copy

import pytest
import torch

from _util import maxdiff
from flowconductor_amd import options
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _build(kind, features, hidden, random_mask=False):
    from flowconductor_amd import transforms as T

    blocks = dict(num_blocks=2, use_residual_blocks=not random_mask, random_mask=random_mask)
    if kind == "maf":
        return T.MaskedAffineAutoregressiveTransform(features, hidden, **blocks)
    if kind == "shift":
        return T.MaskedShiftAutoregressiveTransform(features, hidden, **blocks)
    if kind == "rq_linear_tails":
        return T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            features, hidden, num_bins=8, tails="linear", tail_bound=3.0, **blocks)
    if kind == "rq_box":
        return T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features, hidden, num_bins=5, **blocks)
    if kind == "sos":
        return T.MaskedSumOfSigmoidsTransform(features, hidden, n_sigmoids=6, **blocks)
    if kind == "linear":
        return T.MaskedPiecewiseLinearAutoregressiveTransform(8, features, hidden, **blocks)
    if kind == "quadratic":
        return T.MaskedPiecewiseQuadraticAutoregressiveTransform(6, features, hidden, tails="linear",
                                                                 tail_bound=3.0, **blocks)
    if kind == "cubic":
        return T.MaskedPiecewiseCubicAutoregressiveTransform(6, features, hidden, **blocks)
    raise KeyError(kind)


def _inputs(kind, n, features):
    g = torch.Generator().manual_seed(5)
    if kind in ("linear", "cubic"):
        return torch.rand(n, features, generator=g) * 0.98 + 0.01
    if kind == "rq_box":
        return torch.rand(n, features, generator=g) * 2.2 - 1.1
    if kind == "sos":
        return torch.randn(n, features, generator=g) * 0.6
    return torch.randn(n, features, generator=g) * 1.2


@pytest.mark.parametrize("kind", ["maf", "shift", "rq_linear_tails", "rq_box", "sos", "linear", "quadratic", "cubic"])
@pytest.mark.parametrize("features,hidden,n", [(6, 64, 1000), (5, 24, 77)])
def test_incremental_inverse_matches_full_passes(kind, features, hidden, n, device, monkeypatch):
    torch.manual_seed(31)
    t = _build(kind, features, hidden).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = _inputs(kind, n, features)
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone(), inverse=True)
        ref_y64, ref_lad64 = O.transform_apply(copy.deepcopy(t).double(), x.double(), inverse=True)
    t = t.to(device)
    monkeypatch.setitem(options._values, "ar_device_loop", False)      # (this test: the HOST loops; the device loop below)
    with torch.no_grad():
        # by default only where it pays: >= 8 parameters per dim (not shift / affine) and >= 8192 rows
        assert not t._incremental_ok(x.to(device))
        big = torch.zeros(8192, features, device=device)
        assert t._incremental_ok(big) == (kind not in ("maf", "shift"))
        monkeypatch.setitem(options._values, "ar_incremental", "force")
        assert t._incremental_ok(x.to(device))
        y, lad = t.inverse(x.to(device))
        monkeypatch.setitem(options._values, "ar_incremental", "off")
        assert not t._incremental_ok(x.to(device))
        y_full, lad_full = t.inverse(x.to(device))
    # D chained conditioner passes amplify rounding (the quadratic spline's inverse is ill-conditioned near its
    # knots: the f32 oracle itself is 3e-3 from float64 there), so the bound carries the f32 noise floor
    tol_y = 1e-4 * max(1.0, float(ref_y.abs().max())) + 4 * maxdiff(ref_y, ref_y64)
    tol_l = 1e-3 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * maxdiff(ref_lad, ref_lad64)
    assert maxdiff(y, y_full) <= tol_y
    assert maxdiff(lad, lad_full) <= tol_l
    assert maxdiff(y, ref_y64) <= tol_y
    assert maxdiff(lad, ref_lad64) <= tol_l


def test_incremental_inverse_random_masks_round_trip(device, monkeypatch):
    """Random hidden degrees (feed-forward blocks): inputs/outputs keep degrees 1..D, so the column order is still
    the natural one.  forward(inverse(z)) = z and the log-determinants cancel."""
    torch.manual_seed(3)
    t = _build("rq_linear_tails", 7, 32, random_mask=True).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    t = t.to(device)
    z = torch.randn(500, 7, device=device)
    monkeypatch.setitem(options._values, "ar_incremental", "force")
    monkeypatch.setitem(options._values, "ar_device_loop", False)
    with torch.no_grad():
        assert t._incremental_ok(z)
        x, lad_inv = t.inverse(z)
        z2, lad_fwd = t.forward(x)
    assert maxdiff(z2, z) <= 2e-4 * max(1.0, float(z.abs().max()))
    assert maxdiff(lad_inv + lad_fwd, torch.zeros_like(lad_inv)) <= 2e-3


def test_incremental_inverse_not_used_when_gradients_are_needed(device):
    """The column writes are in place: with autograd recording, the reference's out-of-place D-pass scheme stays."""
    t = _build("rq_linear_tails", 4, 16).to(device)
    x = torch.randn(8192, 4, device=device)
    assert not t._incremental_ok(x)
    with torch.no_grad():
        assert t._incremental_ok(x)


@pytest.mark.parametrize("features,n", [(6, 1000), (32, 4096), (17, 77), (3, 32)])
def test_rq_autoregressive_forward_on_fused_kernels(features, n, device, monkeypatch):
    """Density direction of the RQ-spline AR layer (autoregressive.py:529-621; K = 8, linear tails, hidden 64): MADE
    hidden stack in fc_resnet_hidden, masked final Linear + spline (identity-init softplus beta, no 1/sqrt(H)
    division) in fc_rq_spline_fused_linear.  Against the oracle in float32 / float64 and the unfused path."""
    from flowconductor_amd import ops

    torch.manual_seed(features)
    t = _build("rq_linear_tails", features, 64).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = torch.randn(n, features) * 1.4
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone())
        ref_y64, ref_lad64 = O.transform_apply(copy.deepcopy(t).double(), x.double())
    t = t.to(device)
    with torch.no_grad():
        with ops.KernelTimer("fc_rq_spline_fused_linear") as timer:
            y, lad = t(x.to(device))
        assert len(timer.pairs) == 1, "the fused final-layer + spline kernel did not run"
        monkeypatch.setitem(options._values, "fused_final_layer", False)
        y_unfused, lad_unfused = t(x.to(device))
    tol_y = 2e-5 * max(1.0, float(ref_y.abs().max())) + 4 * maxdiff(ref_y, ref_y64)
    tol_l = 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * maxdiff(ref_lad, ref_lad64)
    assert maxdiff(y, ref_y64) <= tol_y and maxdiff(lad, ref_lad64) <= tol_l
    assert maxdiff(y, y_unfused) <= tol_y and maxdiff(lad, lad_unfused) <= tol_l


@pytest.mark.parametrize("bins,tails,features,n", [(10, None, 6, 1000), (5, None, 32, 4096), (10, "linear", 17, 77),
                                                    (16, None, 3, 64), (10, None, 4, 40)])
def test_rq_autoregressive_forward_general_shapes(bins, tails, features, n, device, monkeypatch):
    """The reference's DEFAULT AR-RQ layer (num_bins = 10, tails = None: the [-1.2, 1.2] box, autoregressive.py:536,595)
    and other bin counts: masked final Linear + spline in fc_rq_spline_fused_general.  Against the oracle in
    float32 / float64 and the unfused path; outside the box the layer raises like the reference."""
    from flowconductor_amd import ops, transforms as T

    torch.manual_seed(bins + features)
    t = T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features, 64, num_bins=bins, tails=tails, tail_bound=3.0,
                                                                  num_blocks=2).eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, features, generator=g) * 1.4 if tails == "linear" else torch.rand(n, features, generator=g) * 2.3 - 1.15
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone())
        ref_y64, ref_lad64 = O.transform_apply(copy.deepcopy(t).double(), x.double())
    t = t.to(device)
    with torch.no_grad():
        with ops.KernelTimer("fc_rq_spline_fused_general") as timer:
            y, lad = t(x.to(device))
        assert len(timer.pairs) == 1, "the general fused final-layer + spline kernel did not run"
        monkeypatch.setitem(options._values, "fused_final_layer", False)
        y_unfused, lad_unfused = t(x.to(device))
        monkeypatch.setitem(options._values, "fused_final_layer", True)
    tol_y = 2e-5 * max(1.0, float(ref_y.abs().max())) + 4 * maxdiff(ref_y, ref_y64)
    tol_l = 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * maxdiff(ref_lad, ref_lad64)
    assert maxdiff(y, ref_y64) <= tol_y and maxdiff(lad, ref_lad64) <= tol_l
    assert maxdiff(y, y_unfused) <= tol_y and maxdiff(lad, lad_unfused) <= tol_l
    if tails is None:
        bad = x.clone()
        bad[3, 1] = 1.3
        with pytest.raises(T.InputOutsideDomain):
            with torch.no_grad():
                t(bad.to(device))


def test_readme_maf_flow_runs_its_made_on_the_hidden_kernel(device):
    """BASELINE configs[0] (the README flow: 2 x [MAF(features=2, hidden_features=4), RandomPermutation]): a MADE with
    4 hidden units runs zero-padded in the 64-wide matrix-core kernels -- the density direction as ONE kernel per layer
    (fc_affine_coupling_resnet on pre-masked weights, round 3), the sampling direction's D passes on fc_resnet_hidden;
    log_prob and samples against the oracle."""
    from flowconductor_amd import distributions, flows, ops, transforms

    torch.manual_seed(0)
    layers = []
    for _ in range(2):
        layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
        layers.append(transforms.RandomPermutation(features=2))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).eval()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(2.0)
    x = torch.randn(4096, 2)
    with torch.no_grad():
        ref = O.flow_log_prob(flow, x)
        z_ref, _ = O.transform_apply(flow._transform, x.clone())
    flow = flow.to(device)
    with torch.no_grad():
        assert flow._transform._transforms[0].autoregressive_net.hip_hidden_supported()
        with ops.KernelTimer("fc_affine_coupling_resnet") as timer, ops.KernelTimer("fc_resnet_hidden") as hidden_timer:
            lp = flow.log_prob(x.to(device))
        assert len(timer.pairs) == 2 and not hidden_timer.pairs, "the MAF layers did not run as one kernel each"
        z, _ = flow._transform(x.to(device))
        with ops.KernelTimer("fc_made_inverse") as loop_timer:
            back, _ = flow._transform.inverse(z)
        assert len(loop_timer.pairs) == 2, "the inverse of the two MAF layers did not run as one fc_made_inverse each"
        with options.override(ar_device_loop=False), ops.KernelTimer("fc_resnet_hidden") as hidden_timer:
            back_host, _ = flow._transform.inverse(z)
        assert len(hidden_timer.pairs) == 4, "the MADE hidden stacks of the host-loop passes did not run in fc_resnet_hidden"
        assert maxdiff(back, back_host) <= 1e-4 * max(1.0, float(x.abs().max()))
    assert maxdiff(lp, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert maxdiff(z, z_ref) <= 2e-5 * max(1.0, float(z_ref.abs().max()))
    assert maxdiff(back, x) <= 1e-4 * max(1.0, float(x.abs().max()))


# ---- round 4: the D passes inside one kernel (fc_made_inverse) -------------------------------------------------------
def _rq(features, hidden, k, tails, blocks=2):
    from flowconductor_amd import transforms as T

    return T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features, hidden, num_bins=k, tails=tails, tail_bound=3.0,
                                                                     num_blocks=blocks)


@pytest.mark.parametrize("kind,features,hidden,n", [
    ("maf", 6, 64, 1000), ("maf", 2, 4, 4096), ("maf", 40, 50, 333), ("rq_linear_tails", 6, 64, 1000),
    ("rq_linear_tails", 33, 24, 200), ("rq_box", 5, 24, 77), ("rq_k10_box", 8, 64, 512), ("rq_k16_tails", 7, 32, 160),
    ("rq_k4_3blocks", 64, 64, 96), ("maf_1block", 9, 16, 50)])
def test_device_loop_inverse_matches_the_reference_scheme(kind, features, hidden, n, device, monkeypatch):
    """fc_made_inverse against the oracle's D full passes (autoregressive.py:44-53) in float32 and float64, and against this
    package's own host loop; affine and RQ forms, D up to 64 (two k-steps of the initial layer), every parameter-tile count
    (P = 2 .. 47), 1-3 blocks, batches that are not whole 16-row blocks."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(features + hidden)
    t = {"maf": lambda: T.MaskedAffineAutoregressiveTransform(features, hidden, num_blocks=2),
         "maf_1block": lambda: T.MaskedAffineAutoregressiveTransform(features, hidden, num_blocks=1),
         "rq_linear_tails": lambda: _rq(features, hidden, 8, "linear"), "rq_box": lambda: _rq(features, hidden, 5, None),
         "rq_k10_box": lambda: _rq(features, hidden, 10, None), "rq_k16_tails": lambda: _rq(features, hidden, 16, "linear"),
         "rq_k4_3blocks": lambda: _rq(features, hidden, 4, "linear", blocks=3)}[kind]().eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = _inputs("rq_box" if "box" in kind else "maf", n, features)
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone(), inverse=True)
        ref_y64, ref_lad64 = O.transform_apply(copy.deepcopy(t).double(), x.double(), inverse=True)
    t = t.to(device)
    from flowconductor_amd import ops
    with torch.no_grad():
        assert t._device_loop_ok(x.to(device), None)
        with ops.KernelTimer("fc_made_inverse") as timer:
            y, lad = t.inverse(x.to(device))
        assert len(timer.pairs) == 1
        monkeypatch.setitem(options._values, "ar_device_loop", False)
        assert not t._device_loop_ok(x.to(device), None)
        y_host, lad_host = t.inverse(x.to(device))
        z, lad_fwd = t.forward(y)
    tol_y = 1e-4 * max(1.0, float(ref_y.abs().max())) + 4 * maxdiff(ref_y, ref_y64)
    tol_l = 1e-3 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * maxdiff(ref_lad, ref_lad64)
    assert maxdiff(y, ref_y64) <= tol_y and maxdiff(lad, ref_lad64) <= tol_l
    assert maxdiff(y, y_host) <= tol_y and maxdiff(lad, lad_host) <= tol_l
    assert maxdiff(z, x) <= 2 * tol_y and maxdiff(lad + lad_fwd, torch.zeros_like(lad)) <= 2 * tol_l


def test_device_loop_steps_aside(device):
    """Forms and conditioners fc_made_inverse does not know keep the host loops: sum-of-sigmoids, a context, autograd."""
    from flowconductor_amd import transforms as T

    x = torch.randn(64, 6, device=device)
    assert not _build("sos", 6, 32).to(device)._device_loop_ok(x, None)
    assert not _build("maf", 6, 32, random_mask=True).to(device)._device_loop_ok(x, None)        # feed-forward blocks
    t = T.MaskedAffineAutoregressiveTransform(6, 32, context_features=3).to(device)
    assert not t._device_loop_ok(x, torch.randn(64, 3, device=device))
    t = _build("maf", 6, 32).to(device)
    assert not t._device_loop_ok(x, None)                     # parameters require gradients, autograd on
    with torch.no_grad():
        assert t._device_loop_ok(x, None)


@pytest.mark.parametrize("features,hidden,blocks", [(8, 64, 2), (64, 64, 2), (33, 40, 3), (5, 24, 1)])
def test_device_loop_prefix_passes_equal_whole_passes(features, hidden, blocks, device):
    """units_needed (each pass computes the leading hidden units it reads) against the same kernel computing all 64 units
    in every pass: the skipped units only meet zeroed weights, so the rows agree to the rounding of the row scales."""
    from flowconductor_amd import ops

    torch.manual_seed(features)
    t = _rq(features, hidden, 8, "linear", blocks=blocks).to(device).eval()
    kind, per_dim, rq = t._device_loop_form()
    assert kind == ops.MADE_RQ and per_dim == 23
    z = torch.randn(4096 + 16, features, device=device)
    with torch.no_grad():
        packed = ops.pack_made_inverse(t.autoregressive_net, features, per_dim)
        need = packed[-1]
        assert int(need[0]) == 0 and int(need.max()) <= hidden and bool((need[1:] >= need[:-1]).all())
        y, lad = ops.made_inverse(z, packed, blocks, per_dim, kind, rq)
        whole = packed[:-1] + (torch.full_like(need, 64),)
        y_all, lad_all = ops.made_inverse(z, whole, blocks, per_dim, kind, rq)
        y_t, lad_t = t.inverse(z)
    assert maxdiff(y, y_all) <= 2e-6 * max(1.0, float(y_all.abs().max()))
    assert maxdiff(lad, lad_all) <= 2e-5 * max(1.0, float(lad_all.abs().max()) / 10)
    assert maxdiff(y, y_t) <= 1e-6 and maxdiff(lad, lad_t) <= 1e-5      # (the transform's own call: same kernel, same pack)
