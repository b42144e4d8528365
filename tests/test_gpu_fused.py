"""Final-conditioner-layer + RQ-spline fused kernel (fc_rq_spline_fused_linear) vs the unfused HIP path
and the CPU oracle."""

import copy

import pytest
import torch

from _util import build_case, maxdiff
from flowconductor_amd import ops, options
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
        monkeypatch.setitem(options._values, "fused_final_layer", False)
        assert not t._fused_ok(xd)
        y_u, lad_u = fn(xd)
    scale = max(1.0, float(ref_y.abs().max()))
    # fused vs unfused: same spline arithmetic, GEMM by split-f16 MFMA vs hipBLASLt
    assert maxdiff(y_f, y_u) <= 5e-5 * scale
    # (the spline inverse is ill-conditioned on a few elements: tools/noise_floor.py, max 1.9e-3 at 2^14 rows)
    assert maxdiff(lad_f, lad_u) <= (1e-3 if not inverse else 1e-2)
    tol_y, tol_l = (2e-5, 2e-4) if not inverse else (3e-4, 3e-3)
    assert maxdiff(y_f[:rows], ref_y) <= tol_y * scale
    assert maxdiff(lad_f[:rows], ref_lad) <= tol_l * max(1.0, float(ref_lad.abs().max()) / 10)
    assert torch.isfinite(y_f).all() and torch.isfinite(lad_f).all()




@pytest.mark.parametrize("in_f,blocks,d,n", [(32, 2, 64, 64), (32, 2, 64, 6400), (16, 1, 32, 128), (6, 0, 12, 64),
                                            (64, 2, 128, 192), (7, 1, 15, 48), (33, 2, 70, 16), (32, 2, 64, 100000), (32, 3, 64, 512),
                                            (24, 4, 48, 1024), (64, 4, 128, 256)])
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


@pytest.mark.parametrize("hidden,in_f,blocks,n", [(64, 32, 2, 1 << 17), (64, 5, 0, 48), (48, 33, 1, 4112), (64, 64, 4, 2048),
                                                  (20, 7, 3, 160)])
def test_resnet_hidden_packed_image_is_bit_identical(hidden, in_f, blocks, n, device):
    """fc_resnet_hidden_packed (weight image made by fc_pack_fragments, copied into LDS) against fc_resnet_hidden (image
    built inside the kernel from the f32 weights): the same bits, on both the one- and two-blocks-per-wave kernels; and
    the image follows in-place parameter updates."""
    from flowconductor_amd.nn import nets

    torch.manual_seed(hidden + in_f)
    d = 2 * in_f
    net = nets.ResidualNet(in_f, 8, hidden_features=hidden, num_blocks=blocks).eval().to(device)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    ids = torch.arange(0, d, 2, device=device)
    x = torch.randn(n, d, device=device)
    with torch.no_grad():
        a = net.hidden_hip(x, ids)                                         # packed image
        b = ops.resnet_hidden(x, ids, ops.pack_resnet_hidden(net), in_f, blocks)
        assert torch.equal(a, b)
        net.initial_layer.weight.mul_(1.5)                                 # in place: the version counter moves
        a2 = net.hidden_hip(x, ids)
        b2 = ops.resnet_hidden(x, ids, ops.pack_resnet_hidden(net), in_f, blocks)
    assert torch.equal(a2, b2) and not torch.equal(a, a2)


@pytest.mark.parametrize("hidden,in_f,blocks,ctx_f", [(32, 16, 2, None), (48, 20, 1, None), (7, 3, 4, None),
                                                     (32, 16, 2, 6), (50, 10, 3, 32)])
def test_resnet_hidden_kernel_narrow_nets_zero_padded(hidden, in_f, blocks, ctx_f, device):
    """Conditioners with fewer than 64 hidden units run in the 64-wide kernel on zero-padded weights: the padding
    units stay exactly 0, the real ones match the module in float64."""
    from flowconductor_amd.nn import nets

    torch.manual_seed(hidden + in_f)
    net = nets.ResidualNet(in_f, 8, hidden_features=hidden, context_features=ctx_f, num_blocks=blocks).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    d, n = 2 * in_f + 1, 4096
    ids = torch.randperm(d)[:in_f].sort().values
    x = torch.randn(n, d)
    c = None if ctx_f is None else torch.randn(n, ctx_f)
    with torch.no_grad():
        ref = copy.deepcopy(net).double().hidden(x.double()[:, ids], None if c is None else c.double())
        ref32 = net.hidden(x[:, ids], c)
        net = net.to(device)
        cd = None if c is None else c.to(device)
        assert net.hip_hidden_supported(d, cd)
        got = net.hidden_hip(x.to(device), ids.to(device), cd)
        out = net.final_from_padded(got)
        out_ref = net.final_layer(net.hidden(x.to(device)[:, ids.to(device)], cd))
    assert got.shape == (n, 64)
    assert float(got[:, hidden:].abs().max()) == 0.0
    assert maxdiff(got[:, :hidden], ref) <= 1e-5 * max(1.0, float(ref.abs().max())) + 4 * maxdiff(ref32, ref)
    assert maxdiff(out, out_ref) <= 2e-5 * max(1.0, float(out_ref.abs().max()))


@pytest.mark.parametrize("hidden_features,n", [(32, 1000), (20, 77)])
def test_fused_flow_narrow_conditioners(hidden_features, n, device, monkeypatch):
    """RQ and affine coupling flows whose conditioners have fewer than 64 hidden units take the same fused kernels
    (1/sqrt(hidden_features) division with the real width).  Against the oracle."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(hidden_features)
    features = 24

    def net(a, b):
        return nets.ResidualNet(a, b, hidden_features=hidden_features, num_blocks=2)

    layers = []
    for i in range(4):
        mask = utils.create_alternating_binary_mask(features, even=(i % 2 == 0))
        layers.append(transforms.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=8, tails="linear",
                                                                             tail_bound=3.0))
        layers.append(transforms.AffineCouplingTransform(mask, net))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([features])).eval()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.5)
    x = torch.randn(n, features) * 1.2
    with torch.no_grad():
        ref = O.flow_log_prob(flow, x)
    flow = flow.to(device)
    with torch.no_grad(), ops.KernelTimer("fc_rq_spline_fused_linear") as fused, \
            ops.KernelTimer("fc_resnet_hidden") as hid, ops.KernelTimer("fc_affine_coupling_resnet") as one:
        lp = flow.log_prob(x.to(device))
    # RQ layers: hidden-stack kernel + fused final layer / spline kernel; affine layers: one kernel each
    assert len(fused.pairs) == 4 and len(hid.pairs) == 4 and len(one.pairs) == 4
    assert maxdiff(lp, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def _activations():
    from torch import nn
    from torch.nn import functional as F

    return {"tanh": torch.tanh, "silu": F.silu, "elu": nn.ELU(alpha=1.3), "leaky_relu": nn.LeakyReLU(0.2),
            "sigmoid": nn.Sigmoid(), "f_elu": F.elu, "f_leaky_relu": F.leaky_relu, "nn_tanh": nn.Tanh()}


@pytest.mark.parametrize("name", ["tanh", "silu", "elu", "leaky_relu", "sigmoid", "f_elu", "f_leaky_relu", "nn_tanh"])
@pytest.mark.parametrize("blocks,ctx_f", [(1, None), (2, None), (2, 5), (4, None)])
def test_resnet_hidden_kernel_other_activations(name, blocks, ctx_f, device):
    """ResidualNets built with tanh / SiLU / ELU / LeakyReLU / sigmoid run in the same kernel family (uniform
    switch); against the module in float64."""
    from flowconductor_amd.nn import nets

    torch.manual_seed(len(name) + blocks)
    net = nets.ResidualNet(12, 8, hidden_features=64, context_features=ctx_f, num_blocks=blocks,
                           activation=_activations()[name]).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    n, d = 2048, 30
    ids = torch.randperm(d)[:12].sort().values
    x = torch.randn(n, d) * 1.5
    x[::7] *= 1e-3      # small arguments: tanh / ELU cancellation regions
    c = None if ctx_f is None else torch.randn(n, ctx_f)
    with torch.no_grad():
        ref = copy.deepcopy(net).double().hidden(x.double()[:, ids], None if c is None else c.double())
        ref32 = net.hidden(x[:, ids], c)
        net = net.to(device)
        cd = None if c is None else c.to(device)
        assert net.hip_hidden_supported(d, cd)
        got = net.hidden_hip(x.to(device), ids.to(device), cd)
    assert maxdiff(got, ref) <= 1e-5 * max(1.0, float(ref.abs().max())) + 4 * maxdiff(ref32, ref)


def test_resnet_hidden_unknown_activation_stays_on_pytorch(device):
    from flowconductor_amd.nn import nets

    net = nets.ResidualNet(12, 8, hidden_features=64, activation=torch.nn.Softsign()).to(device).eval()
    assert not net.hip_hidden_supported(30)
    mixed = nets.ResidualNet(12, 8, hidden_features=64).to(device).eval()
    mixed.blocks[1].activation = torch.nn.Tanh()
    assert not mixed.hip_hidden_supported(30)


@pytest.mark.parametrize("xscale", [1e-6, 1.0, 3e5])
def test_resnet_hidden_kernel_input_scales(xscale, device):
    """Rows far outside the f16 range: the per-row power-of-two scaling keeps the split products at f32-GEMM
    accuracy.  Reference: the same layers in float64."""
    from flowconductor_amd.nn import nets

    torch.manual_seed(3)
    net = nets.ResidualNet(32, 8, hidden_features=64, num_blocks=2).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
        net.initial_layer.weight.div_(xscale)   # activations stay O(1) behind inputs of any magnitude
    ids = torch.arange(0, 64, 2)
    x = torch.randn(256, 64) * xscale
    x[::3] *= 1e-3   # rows of very different magnitude side by side
    with torch.no_grad():
        ref = net.double().hidden(x.double()[:, ids]).float()
        net = net.float()
        ref32 = net.hidden(x[:, ids])
        got = net.to(device).hidden_hip(x.to(device), ids.to(device))
    floor = maxdiff(ref32, ref)   # what f32 GEMMs lose on this input
    assert maxdiff(got, ref) <= 1e-5 * max(1.0, float(ref.abs().max())) + 4 * floor


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
        monkeypatch.setitem(options._values, "fused_hidden", False)
        y2, lad2 = t(x.to(device))
    assert maxdiff(y, ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad, ref_lad) <= 3e-4
    assert maxdiff(y, y2) <= 2e-5 and maxdiff(lad, lad2) <= 3e-4


@pytest.mark.parametrize("case", ["typical", "zero_weights", "huge_h_rows", "tiny_h_rows", "mixed_row_scales",
                                  "tiny_weights", "large_weights", "mixed_weight_rows"])
def test_fused_linear_scaling_edge_cases(case, device):
    """The fused kernel computes W h on the f16 matrix cores after power-of-two scaling (per wave for the
    weights, per row for h).  Parity of the whole op against a float64 Linear + the oracle's spline on
    inputs that stress the scaling: zero-initialised final layer, rows of h far outside the f16 range."""
    torch.manual_seed(7)
    n, d, d_t, k, hidden = 256, 64, 32, 8, 64
    x = torch.randn(n, d) * 1.5
    h = torch.relu(torch.randn(n, hidden)) * 2 + torch.randn(n, hidden) * 0.3
    w = torch.randn(d_t * (3 * k - 1), hidden) * 0.2
    b = torch.randn(d_t * (3 * k - 1)) * 0.1
    if case == "zero_weights":
        w.zero_()
    elif case == "huge_h_rows":
        h *= 3.0e5       # beyond the f16 maximum; the weights shrink so that the logits stay O(1)
        w /= 3.0e5
    elif case == "tiny_h_rows":
        h *= 1.0e-6
        w *= 1.0e6
    elif case == "mixed_row_scales":
        s = torch.logspace(-6, 2, n).unsqueeze(1)
        h *= s
        w *= 0.05       # logits of the large rows stay moderate: the spline amplifies logit errors by their size
    elif case == "tiny_weights":
        w *= 1.0e-7
    elif case == "large_weights":
        w *= 40.0
        h *= 0.025
    elif case == "mixed_weight_rows":
        # magnitudes from 1e-5 to 1 BETWEEN the 96 weight rows that share one wave's power-of-two scale (rows of one
        # dim, and of the four dims of a wave): the small rows keep only the bits above 2^-22 of the wave's maximum
        w *= torch.logspace(-5, 0, d_t * (3 * k - 1))[torch.randperm(d_t * (3 * k - 1))].unsqueeze(1) * 5.0
    cols = torch.arange(0, d, 2, dtype=torch.int32)
    params64 = (h.double() @ w.double().T + b.double())
    rows = params64.float().view(n, d_t, 3 * k - 1).clone()
    out, lad_e = O.rq_from_rows(x[:, cols.long()], rows, k, "linear", 3.0, False, wh_divisor=float(hidden) ** 0.5)
    ref_y = x.clone()
    ref_y[:, cols.long()] = out
    ref_lad = lad_e.sum(dim=1)
    # what an f32 GEMM in front of the same spline gives: its distance from the float64-GEMM result is the
    # noise floor of this input
    rows32 = (h @ w.T + b).view(n, d_t, 3 * k - 1).clone()
    out32, lad32 = O.rq_from_rows(x[:, cols.long()], rows32, k, "linear", 3.0, False, wh_divisor=float(hidden) ** 0.5)
    floor_y = maxdiff(out32, out)
    floor_lad = maxdiff(lad32.sum(dim=1), ref_lad)
    wp, bp = ops.pack_final_layer(w.to(device), b.to(device))
    with torch.no_grad():
        y, lad = ops.rq_spline_fused_linear(x.to(device), h.to(device), wp, bp, cols.to(device), num_bins=k,
                                            tail_bound=3.0, wh_divisor=float(hidden) ** 0.5)
    # same bounds as the flow-level fused test above (f32 spline arithmetic on both sides)
    assert maxdiff(y, ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max())) + 4 * floor_y
    assert maxdiff(lad, ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * floor_lad
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()


@pytest.mark.parametrize("n", [64, 1000])
@pytest.mark.parametrize("inverse", [False, True])
def test_composite_accumulates_logabsdet_in_kernel(n, inverse, device):
    """CompositeTransform hands its running total to the fused kernel (FC_RQ_ACCUMULATE_LOGABSDET); the result
    must be bit-identical to summing the layers' logabsdets with separate adds (base.py:45-52 order)."""
    from flowconductor_amd import transforms

    layers = [build_case("rq_coupling_linear_tails_d64_k8_h64")[0] for _ in range(3)]
    comp = transforms.CompositeTransform(layers).to(device).eval()
    x = (torch.randn(n, 64, generator=torch.Generator().manual_seed(5)) * 1.5).to(device)
    with torch.no_grad():
        with ops.KernelTimer("fc_rq_spline_fused_linear") as timer:
            y, total = (comp.inverse if inverse else comp)(x)
        assert len(timer.pairs) == 3
        out, ref_total = x, torch.zeros(n, device=device)
        for t in (layers[::-1] if inverse else layers):
            out, lad = (t.inverse if inverse else t)(out)
            ref_total += lad
    assert torch.equal(y, out)
    assert torch.equal(total, ref_total)


@pytest.mark.parametrize("d,n,d_t", [(36, 96, 32), (48, 4096, 32), (64, 32, 32), (96, 2080, 32), (112, 640, 32),
                                     (128, 1056, 32), (32, 4096, 16), (8, 160, 4), (40, 992, 20), (64, 2048, 28), (44, 2048, 21),
                                     (12, 96, 1), (64, 320, 31), (63, 2048, 31), (21, 640, 10), (43, 1056, 22), (6, 64, 3)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fused_linear_input_widths(d, n, d_t, inverse, device):
    """The fused kernel's tile variants: 64-row tiles with 1 / 2 / 4 float4 of x per thread (D <= 32 / 64 /
    112), 32-row tiles for wider inputs and for the last 32 rows of an odd multiple of 32; any number of
    transformed dims up to 32 (wave w owns dims 4w..4w+3; the last group may be partly padding)."""
    torch.manual_seed(d)
    k, hidden = 8, 64
    x = torch.randn(n, d) * 1.5
    h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
    w = torch.randn(d_t * (3 * k - 1), hidden) * 0.2
    b = torch.randn(d_t * (3 * k - 1)) * 0.1
    cols = torch.randperm(d)[:d_t].sort().values
    rows = (h.double() @ w.double().T + b.double()).float().view(n, d_t, 3 * k - 1).clone()
    out, lad_e = O.rq_from_rows(x[:, cols], rows, k, "linear", 3.0, inverse, wh_divisor=float(hidden) ** 0.5)
    ref_y = x.clone()
    ref_y[:, cols] = out
    ref_lad = lad_e.sum(dim=1)
    wp, bp = ops.pack_final_layer(w.to(device), b.to(device))
    with torch.no_grad():
        y, lad = ops.rq_spline_fused_linear(x.to(device), h.to(device), wp, bp, cols.to(device), num_bins=k,
                                            tail_bound=3.0, wh_divisor=float(hidden) ** 0.5, inverse=inverse)
    tol_y, tol_l = (2e-5, 2e-4) if not inverse else (3e-4, 3e-3)
    assert maxdiff(y, ref_y) <= tol_y * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad, ref_lad) <= tol_l * max(1.0, float(ref_lad.abs().max()) / 10)


@pytest.mark.parametrize("kind,d,hidden,blocks,n", [("affine", 32, 64, 2, 4096), ("affine", 10, 20, 1, 100), ("general", 64, 64, 3, 1040),
                                                    ("additive", 24, 48, 0, 333), ("affine", 80, 64, 2, 512)])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_coupling_layer_in_one_kernel(kind, d, hidden, blocks, n, inverse, device):
    """fc_affine_coupling_resnet (hidden stack + final Linear + affine / additive bijector, one launch per layer) against
    the oracle and against the three-kernel path; leftover rows, narrow nets, 48 identity features (two k-steps), the
    clamped-softplus scale, the running logabsdet total of a CompositeTransform."""
    from flowconductor_amd import options, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(d + blocks)
    mask = utils.create_alternating_binary_mask(d, even=True)
    if d == 80:       # 48 identity features (two k-steps of the initial layer), 32 transformed
        mask = torch.cat((torch.zeros(48), torch.ones(32)))

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=blocks)

    if kind == "additive":
        t = transforms.AdditiveCouplingTransform(mask, net)
    else:
        act = (transforms.AffineCouplingTransform.GENERAL_SCALE_ACTIVATION if kind == "general"
               else transforms.AffineCouplingTransform.DEFAULT_SCALE_ACTIVATION)
        t = transforms.AffineCouplingTransform(mask, net, scale_activation=act)
    t = t.eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.7)
        t.transform_net.final_layer.bias.add_(torch.randn_like(t.transform_net.final_layer.bias) * 0.3)
    x = torch.randn(n, d) * 1.3
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x, inverse=inverse)
    tg = t.to(device)
    xd = x.to(device)
    with torch.no_grad(), ops.KernelTimer("fc_affine_coupling_resnet") as one, ops.KernelTimer("fc_affine") as three:
        y, lad = (tg.inverse if inverse else tg)(xd)
    assert len(one.pairs) == 1 and len(three.pairs) == (1 if n % 16 else 0)
    tol = 1e-4 if inverse else 2e-5
    assert maxdiff(y, ref_y) <= tol * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad, ref_lad) <= 1e-4 * max(1.0, float(ref_lad.abs().max()))
    with torch.no_grad(), options.override(fused_final_layer=False):
        y3, lad3 = (tg.inverse if inverse else tg)(xd)
    assert maxdiff(y, y3) <= tol * max(1.0, float(ref_y.abs().max()))
    ident = tg.identity_features.to(device)
    assert torch.equal(y[:, ident], xd[:, ident])
    # a row-sliced (possibly 16-byte-misaligned) view gives the rows of the full batch
    with torch.no_grad():
        yv, ladv = (tg.inverse if inverse else tg)(xd[1:])
    assert maxdiff(yv, y[1:]) <= tol * max(1.0, float(ref_y.abs().max())) and maxdiff(ladv, lad[1:]) <= 1e-5 * max(1.0, float(ref_lad.abs().max()))
    # inside a CompositeTransform: the kernel adds onto the running total
    comp = transforms.CompositeTransform([tg, transforms.ReversePermutation(d), tg])
    with torch.no_grad():
        yc, ladc = (comp.inverse if inverse else comp)(xd)
        a1, l1 = (tg.inverse if inverse else tg)(xd)
        a2, _ = transforms.ReversePermutation(d).to(device)(a1)
        a3, l3 = (tg.inverse if inverse else tg)(a2)
    assert torch.equal(yc, a3) and maxdiff(ladc, l1 + l3) <= 1e-6 * max(1.0, float((l1 + l3).abs().max()))


@pytest.mark.parametrize("d,d_t,n", [(64, 32, 4096), (44, 21, 2080), (12, 3, 96)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fused_linear_raw_weights_equal_padded(d, d_t, n, inverse, device):
    """FC_RQ_RAW_WEIGHTS: the final Linear's [d_t * 23, 64] weight and bias as they are give the same bits as the
    zero-padded [ceil(d_t / 4) * 4 * 24, 64] arrays (the kernel reads the padding rows as zeros either way)."""
    torch.manual_seed(d + d_t)
    x = torch.randn(n, d, device=device) * 1.5
    h = torch.randn(n, 64, device=device)
    w = torch.randn(d_t * 23, 64, device=device) * 0.2
    b = torch.randn(d_t * 23, device=device) * 0.1
    cols = torch.randperm(d)[:d_t].sort().values.to(device)
    kw = dict(num_bins=8, tail_bound=3.0, wh_divisor=8.0, inverse=inverse)
    wp, bp = ops.pack_final_layer(w, b)
    with torch.no_grad():
        y0, l0 = ops.rq_spline_fused_linear(x, h, wp, bp, cols, **kw)
        y1, l1 = ops.rq_spline_fused_linear(x, h, w, b, cols, **kw)
    assert torch.equal(y0, y1) and torch.equal(l0, l1)


@pytest.mark.parametrize("kind", ["maf", "rq_ar"])
@pytest.mark.parametrize("inverse", [False, True])
def test_made_hidden_stack_on_hip_kernel(kind, inverse, device, monkeypatch):
    """A MADE with hidden 64 / residual blocks runs its hidden stack in fc_resnet_hidden on pre-masked weights
    (both for the single forward pass and for each of the D passes of the inverse); result vs the oracle and vs
    the PyTorch-ROCm conditioner."""
    from flowconductor_amd import transforms

    torch.manual_seed(21)
    if kind == "maf":
        t = transforms.MaskedAffineAutoregressiveTransform(features=6, hidden_features=64, num_blocks=2)
    else:
        t = transforms.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            features=6, hidden_features=64, num_blocks=2, num_bins=8, tails="linear", tail_bound=3.0)
    t.eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = torch.randn(1000, 6) * 1.2
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t, x.clone(), inverse=inverse)
    t = t.to(device)
    fn = t.inverse if inverse else t.forward
    with torch.no_grad():
        assert t.autoregressive_net.hip_hidden_supported()
        with ops.KernelTimer("fc_resnet_hidden") as timer, ops.KernelTimer("fc_affine_coupling_resnet") as one, \
                ops.KernelTimer("fc_made_inverse") as loop:
            y, lad = fn(x.to(device))
        if kind == "maf" and not inverse:      # round 3: the affine form's density direction is ONE kernel (hidden stack inside)
            assert len(one.pairs) == 1 and not timer.pairs, "the one-kernel MAF path did not run"
        elif inverse:                          # round 4: the D passes of the inverse run inside ONE kernel
            assert len(loop.pairs) == 1 and not timer.pairs, "the device-loop inverse did not run"
            with options.override(ar_device_loop=False), ops.KernelTimer("fc_resnet_hidden") as host_timer:
                y_host, lad_host = fn(x.to(device))
            assert len(host_timer.pairs) == 6, "the hidden-layer kernel did not run in the host loop"
            assert maxdiff(y, y_host) <= 3e-4 * max(1.0, float(ref_y.abs().max()))
        else:
            assert len(timer.pairs) == 1, "the hidden-layer kernel did not run"
        monkeypatch.setitem(options._values, "fused_hidden", False)
        y2, lad2 = fn(x.to(device))
    scale = max(1.0, float(ref_y.abs().max()))
    tol = 3e-4 if inverse else 2e-5      # the inverse chains D conditioner passes
    assert maxdiff(y, ref_y) <= tol * scale
    assert maxdiff(lad, ref_lad) <= 10 * tol * max(1.0, float(ref_lad.abs().max()) / 10)
    assert maxdiff(y, y2) <= tol * scale


@pytest.mark.parametrize("features,n", [(32, 1000), (20, 333), (63, 200), (128, 500), (100, 333), (70, 64)])
def test_fused_flow_other_widths(features, n, device, monkeypatch):
    """Flows whose coupling layers transform fewer than 32 dims (D = 32 -> 16, D = 20 -> 10) take the fused kernels
    too, also with an odd feature count (63: rows that are not a whole number of float4); layers that transform more
    than 32 dims (D = 128 -> 64, 100 -> 50, 70 -> 35) chain one launch per 32 dims.  Against the oracle."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(features)
    layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(features, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2), num_bins=8, tails="linear",
        tail_bound=3.0) for i in range(4)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([features])).eval()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.5)
    x = torch.randn(n, features) * 1.3
    with torch.no_grad():
        ref = O.flow_log_prob(flow, x)
    flow = flow.to(device)
    with torch.no_grad(), ops.KernelTimer("fc_rq_spline_fused_linear") as timer:
        lp = flow.log_prob(x.to(device))
    transformed = [int(l.num_transform_features) for l in layers]
    assert len(timer.pairs) == sum(-(-t // 32) for t in transformed), len(timer.pairs)
    assert maxdiff(lp, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # sampling direction through the same chained launches
    with torch.no_grad():
        z, lad = flow._transform(x.to(device))
        back, lad_inv = flow._transform.inverse(z)
    assert maxdiff(back, x) <= 2e-4 * max(1.0, float(x.abs().max()))
    assert maxdiff(lad + lad_inv, torch.zeros_like(lad)) <= 2e-3
