"""The general fused final-Linear + RQ-spline kernel (fc_rq_spline_fused_general): K = 4..16, linear tails or none,
hidden 64 / 128 / 256 (the reference's default layer has num_bins = 10, coupling.py:507, and any hidden_features,
nn/nets/resnet.py:62) -- against the CPU oracle, against the unfused HIP path, and at operator level against a float64
Linear in front of the oracle's spline."""
import pytest
import torch

from _util import Lib, maxdiff
from flowconductor_amd import ops, options
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
T, nets, utils = Lib.transforms, Lib.nets, Lib.utils


def _layer(d, hidden, k, tails, seed, blocks=2, even=True):
    torch.manual_seed(seed)
    t = T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=even),
        lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=blocks),
        num_bins=k, tails=tails, tail_bound=3.0).eval()
    with torch.no_grad():          # away from the near-identity default initialisation: every bin in use
        lin = t.transform_net.final_layer
        lin.weight.copy_(torch.randn(lin.weight.shape) * (1.5 / hidden ** 0.5))
        lin.bias.copy_(torch.randn(lin.bias.shape) * 0.3)
    return t


def _inputs(n, d, tails, seed):
    gen = torch.Generator().manual_seed(seed)
    if tails is None:
        return torch.rand(n, d, generator=gen)           # the unit box
    x = torch.randn(n, d, generator=gen) * 1.6
    if n >= 8:
        x[2], x[3], x[4], x[5] = 3.0, -3.0, 4.5, -6.0      # on and beyond the tail bound
    return x


@pytest.mark.parametrize("k", [4, 5, 7, 8, 10, 13, 16])
@pytest.mark.parametrize("tails", ["linear", None])
@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_general_fused_layer_matches_oracle_and_unfused(k, tails, hidden, device, monkeypatch):
    """(K, tails, hidden) grid at D = 16: forward and inverse, the fused kernel must be the one that runs."""
    if k == 8 and tails == "linear" and hidden == 64:
        pytest.skip("the north-star shape runs in fc_rq_spline_fused_linear (tests/test_gpu_fused.py)")
    d, n = 16, 1000          # 1000 = 31 tiles of 32 rows + 8 leftover rows
    t = _layer(d, hidden, k, tails, seed=100 * k + hidden)
    x = _inputs(n, d, tails, seed=k)
    import copy
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(t, x.clone())
        xi_ref, ladi_ref = O.transform_apply(t, y_ref.clone(), inverse=True)
        # the reference's own f32 noise floor on these (steep, every-bin) splines: |f32 - f64| of the oracle
        t64 = copy.deepcopy(t).double()
        y64, lad64 = O.transform_apply(t64, x.double())
        xi64, ladi64 = O.transform_apply(t64, y_ref.double(), inverse=True)
    fy, fl, fxi, fli = (maxdiff(a, b) for a, b in ((y_ref, y64), (lad_ref, lad64), (xi_ref, xi64), (ladi_ref, ladi64)))
    td, xd = t.to(device), x.to(device)
    with torch.no_grad():
        assert td._fused_mode(xd) == "general"
        with ops.KernelTimer("fc_rq_spline_fused_general") as timer:
            y, lad = td(xd)
            xi, ladi = td.inverse(y_ref.to(device))
        assert len(timer.pairs) == 2, "the general fused kernel did not run"
        monkeypatch.setitem(options._values, "fused_final_layer", False)
        y_u, lad_u = td(xd)
    scale = max(1.0, float(y_ref.abs().max()))
    lscale = max(1.0, float(lad_ref.abs().max()))
    # 1e-5 relative + 8 x the reference's own float32 noise floor (the bound of tests/test_gpu_golden.py)
    assert maxdiff(y, y_ref) <= 1e-5 * scale + 8 * fy
    assert maxdiff(lad, lad_ref) <= 1e-5 * lscale + 8 * fl
    assert maxdiff(y, y_u) <= 5e-5 * scale and maxdiff(lad, lad_u) <= 1e-3
    assert maxdiff(xi, xi_ref) <= 1e-5 * scale + 8 * fxi
    assert maxdiff(ladi, ladi_ref) <= 1e-5 * lscale + 8 * fli
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()


@pytest.mark.parametrize("d,hidden,k,tails,n", [(64, 256, 10, "linear", 4096), (64, 128, 10, None, 2048),
                                                (128, 64, 10, "linear", 640), (6, 48, 10, None, 96),
                                                (63, 200, 12, "linear", 352), (10, 100, 6, "linear", 33)])
def test_general_fused_layer_widths(d, hidden, k, tails, n, device):
    """Wide inputs (two launches of 32 dims each at D = 128), odd feature counts (unpadded x rows), hidden widths that
    are zero-padded to 64 / 128 / 256, few transformed dims (waves without spline work)."""
    t = _layer(d, hidden, k, tails, seed=d + hidden)
    x = _inputs(n, d, tails, seed=d)
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(t, x.clone())
        td = t.to(device)
        with ops.KernelTimer("fc_rq_spline_fused_general") as timer:
            y, lad = td(x.to(device))
        assert len(timer.pairs) == -(-t.num_transform_features // 32)
        xb, ladb = td.inverse(y)
    scale = max(1.0, float(y_ref.abs().max()))
    assert maxdiff(y, y_ref) <= 2e-5 * scale
    assert maxdiff(lad, lad_ref) <= 2e-4 * max(1.0, float(lad_ref.abs().max()) / 10)
    inside = (x.abs() <= 3.0) if tails == "linear" else torch.ones_like(x, dtype=torch.bool)
    assert float(((xb.cpu() - x).abs() * inside).max()) <= 2e-3      # ill-conditioned elements of the inverse
    assert float((lad + ladb).abs().max()) <= 2e-2


def test_general_fused_box_raises_outside_domain(device):
    from flowconductor_amd.transforms import InputOutsideDomain

    t = _layer(8, 64, 10, None, seed=3).to(device)
    x = torch.rand(64, 8, device=device)
    x[5, 0] = 1.25      # a transformed column (even mask) outside [0, 1]
    with pytest.raises(InputOutsideDomain):
        with torch.no_grad():
            t(x)
    with torch.no_grad():
        t(torch.rand(64, 8, device=device))       # the error word is cleared


@pytest.mark.parametrize("k,tails,hidden", [(10, "linear", 256), (10, None, 64), (16, "linear", 128), (4, None, 256)])
def test_general_fused_operator_against_float64_linear(k, tails, hidden, device):
    """Operator level: h, W, b given; reference = float64 Linear + the oracle's spline.  The f32-GEMM result in front
    of the same spline sets the noise floor of this input (as in test_fused_linear_scaling_edge_cases)."""
    torch.manual_seed(11 * k + hidden)
    n, d, d_t = 256, 64, 32
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    x = torch.rand(n, d) if tails is None else torch.randn(n, d) * 1.5
    h = torch.relu(torch.randn(n, hidden)) * 2 + torch.randn(n, hidden) * 0.3
    h *= torch.logspace(-3, 2, n).unsqueeze(1)                 # row scales over five decades
    w = torch.randn(d_t * p, hidden) * (0.1 / hidden ** 0.5) * torch.logspace(-3, 0, d_t * p)[torch.randperm(d_t * p)].unsqueeze(1) * 5
    b = torch.randn(d_t * p) * 0.2
    cols = torch.arange(0, d, 2, dtype=torch.int32)
    kw = dict(wh_divisor=float(hidden) ** 0.5)
    rows64 = (h.double() @ w.double().T + b.double()).float().view(n, d_t, p).clone()
    rows32 = (h @ w.T + b).view(n, d_t, p).clone()
    out, lad_e = O.rq_from_rows(x[:, cols.long()], rows64, k, tails, 3.0, False, **kw)
    out32, lad32 = O.rq_from_rows(x[:, cols.long()], rows32, k, tails, 3.0, False, **kw)
    ref_y = x.clone()
    ref_y[:, cols.long()] = out
    ref_lad = lad_e.sum(dim=1)
    floor_y, floor_lad = maxdiff(out32, out), maxdiff(lad32.sum(dim=1), ref_lad)
    frag, wun, bpad = ops.pack_final_layer_general(w.to(device), b.to(device), k, tails, hidden)
    with torch.no_grad():
        y, lad = ops.rq_spline_fused_general(x.to(device), h.to(device), frag, wun, bpad, cols.to(device), num_bins=k,
                                             tails=tails, tail_bound=3.0, **kw)
    assert maxdiff(y, ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max())) + 4 * floor_y
    assert maxdiff(lad, ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * floor_lad


def test_default_constructed_nsf_layer_takes_the_fused_path(device):
    """The reference's DEFAULT layer (num_bins = 10, tails = None, coupling.py:507-547) in a small flow, against the
    oracle; every layer must run the general fused kernel (missing #1)."""
    torch.manual_seed(0)
    d = 12
    layers = [T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=128, num_blocks=2)) for i in range(4)]
    stack = T.CompositeTransform(layers).eval()
    x = torch.rand(500, d)
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(stack, x.clone())
        with ops.KernelTimer("fc_rq_spline_fused_general") as timer:
            y, lad = stack.to(device)(x.to(device))
    assert len(timer.pairs) == 4
    assert maxdiff(y, y_ref) <= 2e-5 and maxdiff(lad, lad_ref) <= 3e-4


# ---- the wide hidden-layer kernel (fc_resnet_hidden_wide) -----------------------------------------------------------

@pytest.mark.parametrize("hidden,in_f,blocks,d,n", [(256, 32, 2, 64, 64), (256, 32, 2, 64, 6400), (128, 32, 2, 64, 1024),
                                                   (128, 16, 1, 32, 128), (256, 64, 3, 128, 192), (200, 7, 2, 15, 64),
                                                   (100, 33, 4, 70, 256), (256, 6, 0, 12, 64), (128, 32, 2, 64, 100032)])
def test_wide_hidden_kernel_matches_torch_float64(hidden, in_f, blocks, d, n, device):
    """fc_resnet_hidden_wide against the same nn.Module evaluated in float64 on the CPU; widths that are zero-padded to
    128 / 256 keep their padding columns exactly 0."""
    import copy

    torch.manual_seed(hidden + in_f + blocks)
    net = nets.ResidualNet(in_f, 8, hidden_features=hidden, num_blocks=blocks).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)      # make the residual blocks matter (their last layer is initialised ~1e-3)
    ids = torch.randperm(d)[:in_f].sort().values
    x = torch.randn(n, d) * torch.logspace(-2, 1, n).unsqueeze(1)      # row scales over three decades
    with torch.no_grad():
        ref = copy.deepcopy(net).double().hidden(x[:, ids].double())
        ref32 = net.hidden(x[:, ids])
        assert net.hip_hidden_wide_supported(d)
        with ops.KernelTimer("fc_resnet_hidden_wide") as timer:
            got = net.to(device).hidden_hip_wide(x.to(device), ids.to(device))
        assert len(timer.pairs) == 1
    width = 128 if hidden <= 128 else 256
    assert got.shape == (n, width)
    assert float(got[:, hidden:].abs().max()) == 0.0 if hidden < width else True
    floor = maxdiff(ref32, ref)      # an f32 GEMM stack's own distance from the float64 result
    assert maxdiff(got[:, :hidden], ref) <= 1e-5 * max(1.0, float(ref.abs().max())) + 4 * floor


@pytest.mark.parametrize("name", ["tanh", "silu", "elu", "leaky_relu", "sigmoid"])
def test_wide_hidden_kernel_other_activations(name, device):
    import copy

    from torch import nn
    from torch.nn import functional as F

    acts = {"tanh": torch.tanh, "silu": F.silu, "elu": nn.ELU(alpha=1.3), "leaky_relu": nn.LeakyReLU(0.2),
            "sigmoid": nn.Sigmoid()}
    torch.manual_seed(5)
    net = nets.ResidualNet(20, 8, hidden_features=160, num_blocks=2, activation=acts[name]).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.7)
    ids = torch.arange(1, 41, 2)
    x = torch.randn(320, 41)
    with torch.no_grad():
        ref = copy.deepcopy(net).double().hidden(x[:, ids].double())
        got = net.to(device).hidden_hip_wide(x.to(device), ids.to(device))
    assert maxdiff(got[:, :160], ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert float(got[:, 160:].abs().max()) == 0.0


def test_wide_nsf_flow_runs_both_wide_kernels(device):
    """K = 10, hidden 256 (the shape of bench.py's `nsf_k10_h256` block): every layer = one fc_resnet_hidden_wide + one
    fc_rq_spline_fused_general launch; against the oracle, with leftover rows (n % 64 != 0)."""
    torch.manual_seed(0)
    d = 64
    layers = [T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=256, num_blocks=2), num_bins=10, tails="linear",
        tail_bound=3.0) for i in range(4)]
    stack = T.CompositeTransform(layers).eval()
    x = torch.randn(64 * 9 + 37, d) * 1.3
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(stack, x.clone())
        with ops.KernelTimer("fc_rq_spline_fused_general") as tf, ops.KernelTimer("fc_resnet_hidden_wide") as th:
            y, lad = stack.to(device)(x.to(device))
    assert len(tf.pairs) == 4 and len(th.pairs) == 4
    assert maxdiff(y, y_ref) <= 2e-5 * max(1.0, float(y_ref.abs().max())) and maxdiff(lad, lad_ref) <= 3e-4
