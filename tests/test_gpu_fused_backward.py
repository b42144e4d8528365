"""Backward of the fused final-Linear + RQ-spline layer (fc_rq_fused_linear_backward) and the training path built on
it, against torch.autograd on the oracle in float64 (the gradient-consistency oracle of the reference's
tests/transforms/transform_test.py:29-37 is autograd itself)."""
import copy

import pytest
import torch

from _util import Lib, maxdiff
from flowconductor_amd import ops, options
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
T, nets, utils, flows, distributions = Lib.transforms, Lib.nets, Lib.utils, Lib.flows, Lib.distributions


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("k,tails,d,d_t,n", [(8, "linear", 64, 32, 256), (10, "linear", 64, 32, 160), (10, None, 16, 8, 96),
                                            (4, "linear", 12, 6, 64), (8, "linear", 63, 31, 96), (5, None, 10, 3, 32)])
def test_fused_linear_backward_operator_vs_float64_autograd(k, tails, d, d_t, n, device):
    torch.manual_seed(7 * k + d)
    hidden = 64
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    x = torch.rand(n, d) if tails is None else torch.randn(n, d) * 1.5
    if tails == "linear" and n >= 8:
        x[3], x[4] = 4.0, -3.5                                  # rows outside the tail interval: identity, no gradient
    h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
    h *= torch.logspace(-2, 1, n).unsqueeze(1)
    w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
    b = torch.randn(d_t * p) * 0.3
    cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
    gy = torch.randn(n, d)
    gl = torch.randn(n)
    kw = dict(wh_divisor=float(hidden) ** 0.5)
    # float64 autograd reference
    x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
    rows = (h64 @ w64.T + b64).view(n, d_t, p)
    out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
    y64 = x64.clone()
    y64 = y64.index_copy(1, cols.long(), out)
    loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
    gx_ref, gh_ref, gw_ref, gb_ref = torch.autograd.grad(loss, (x64, h64, w64, b64))
    packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, tails, 64)
    packed_t = ops.pack_final_layer_transposed(w.to(device), k, tails)
    gx, gh, gw, gb = ops.rq_fused_linear_backward(x.to(device), h.to(device), gy.to(device), gl.to(device), packed,
                                                  packed_t, cols.to(device), num_bins=k, tails=tails, tail_bound=3.0,
                                                  **kw)
    # f32 arithmetic against a float64 reference: relative to the largest entry of each gradient
    assert _relerr(gx, gx_ref) <= 2e-4
    assert _relerr(gh, gh_ref) <= 2e-4
    assert _relerr(gw, gw_ref) <= 2e-4
    assert _relerr(gb, gb_ref) <= 2e-4
    # the identity columns pass the upstream gradient through bit for bit
    ident = [c for c in range(d) if c not in set(cols.tolist())]
    assert torch.equal(gx[:, ident].cpu(), gy[:, ident])


def _layer(d, hidden, k, tails, seed, blocks=2, mask="alternating"):
    torch.manual_seed(seed)
    t = T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=True) if mask == "alternating" else utils.create_mid_split_binary_mask(d),
        lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=blocks),
        num_bins=k, tails=tails, tail_bound=3.0)
    with torch.no_grad():
        lin = t.transform_net.final_layer
        lin.weight.copy_(torch.randn(lin.weight.shape) * (1.0 / hidden ** 0.5))
        lin.bias.copy_(torch.randn(lin.bias.shape) * 0.3)
        for p in t.transform_net.blocks.parameters():
            p.mul_(1.5)
    return t


@pytest.mark.parametrize("d,hidden,k,tails,n,mask", [(64, 64, 8, "linear", 256, "alternating"), (64, 64, 10, "linear", 200, "alternating"),
                                                     (16, 32, 10, None, 77, "alternating"), (128, 64, 8, "linear", 96, "alternating"),
                                                     (10, 20, 5, "linear", 33, "alternating"), (64, 64, 8, "linear", 160, "mid_split"),
                                                     (54, 32, 8, "linear", 512, "alternating")])   # K = 8 with a narrow net
def test_coupling_layer_trains_through_fused_kernels(d, hidden, k, tails, n, mask, device):
    """Parameter and input gradients of one RQ coupling layer on the fused training path (forward: fc_resnet_hidden +
    fc_rq_spline_fused_general; backward: fc_rq_fused_linear_backward once per 32 transformed dims) against float64
    autograd on the oracle.  Batches that are not whole 32-row tiles, D = 128 (two groups of 32 dims), narrow nets; a
    contiguous-half mask (the hidden stack's input gradient then takes the scalar path of fc_resnet_hidden_backward_accum,
    the alternating masks the 16-byte one)."""
    t_cpu = _layer(d, hidden, k, tails, seed=d + k, mask=mask)
    t_gpu = copy.deepcopy(t_cpu).to(device).train()
    t_cpu = t_cpu.double().train()
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(n, d, generator=gen) if tails is None else torch.randn(n, d, generator=gen) * 1.5
    gy = torch.randn(n, d, generator=gen)
    gl = torch.randn(n, generator=gen)
    x64 = x.double().requires_grad_(True)
    y_ref, lad_ref = O.transform_apply(t_cpu, x64)
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()

    xd = x.to(device).requires_grad_(True)
    groups = -(-t_gpu.num_transform_features // 32)
    with ops.KernelTimer("fc_rq_fused_linear_backward") as tb, ops.KernelTimer("fc_rq_spline_fused_general") as tf, \
            ops.KernelTimer("fc_rq_spline_fused_linear") as tf8, ops.KernelTimer("fc_rq_spline_backward") as told:
        y, lad = t_gpu(xd)
        ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    # forward: the hand-scheduled K = 8 kernel for the north-star shape (it takes the 64-wide nn.Linear weights as they
    # are), the general one otherwise
    # backward: one launch per 32 dims (fc_rq_fused_backward512.h)
    assert len(tb.pairs) == groups and len(tf.pairs) + len(tf8.pairs) == groups and not told.pairs
    assert bool(tf8.pairs) == (k == 8 and tails == "linear" and hidden == 64)
    assert maxdiff(y.detach(), y_ref.detach()) <= 2e-5 * max(1.0, float(y_ref.detach().abs().max()))
    assert maxdiff(lad.detach(), lad_ref.detach()) <= 3e-4
    assert _relerr(xd.grad, x64.grad) <= 2e-4
    for (name, p_ref), (_, p) in zip(t_cpu.named_parameters(), t_gpu.named_parameters()):
        assert p.grad is not None, name
        assert _relerr(p.grad, p_ref.grad) <= 3e-4, name


def test_flow_trains_on_the_fused_path_like_the_unfused_one(device):
    """-log_prob(x).mean().backward() through a 6-layer flow (reference default K = 10): fused training path vs the
    conditioner-on-PyTorch path, same weights: the same loss and the same gradients to rounding; an Adam step lowers the
    loss (examples/toy_2d.py:57-68)."""
    torch.manual_seed(0)
    d = 16
    layers = [T.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2), num_bins=10, tails="linear",
        tail_bound=3.0) for i in range(6)]
    flow = flows.Flow(T.CompositeTransform(layers), distributions.StandardNormal([d])).to(device).train()
    x = (torch.randn(1000, d) * 0.6 + 0.4).to(device)
    loss_f = -flow.log_prob(x).mean()
    loss_f.backward()
    grads_f = [p.grad.clone() for p in flow.parameters()]
    flow.zero_grad()
    with options.override(fused_training=False):
        loss_u = -flow.log_prob(x).mean()
        loss_u.backward()
    assert abs(float(loss_f.detach()) - float(loss_u.detach())) <= 1e-5 * max(1.0, abs(float(loss_u.detach())))
    for gf, p in zip(grads_f, flow.parameters()):
        assert _relerr(gf, p.grad) <= 5e-4
    opt = torch.optim.Adam(flow.parameters(), lr=2e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] and all(l == l for l in losses)


@pytest.mark.parametrize("hidden,in_f,blocks,d,n", [(64, 32, 2, 64, 256), (64, 32, 2, 64, 128 * 9), (32, 16, 1, 32, 128),
                                                   (64, 64, 2, 128, 384), (20, 7, 2, 15, 128), (64, 6, 0, 12, 128)])
def test_hidden_backward_kernel_vs_float64_autograd(hidden, in_f, blocks, d, n, device):
    """fc_resnet_hidden_backward: gradients wrt the identity inputs and every weight / bias of the hidden stack against
    float64 autograd through the same nn.Module (activations recomputed in the kernel, nothing saved but x)."""
    torch.manual_seed(hidden + in_f + blocks)
    net = nets.ResidualNet(in_f, 8, hidden_features=hidden, num_blocks=blocks)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    ids = torch.randperm(d)[:in_f].sort().values
    x = torch.randn(n, d) * torch.logspace(-1, 0.5, n).unsqueeze(1)
    gh = torch.randn(n, 64) * torch.logspace(-2, 1, n).flip(0).unsqueeze(1)
    gh[:, hidden:] = 0
    net64 = copy.deepcopy(net).double()
    xid = x[:, ids].double().requires_grad_(True)
    h = net64.hidden(xid)
    params = [net64.initial_layer.weight, net64.initial_layer.bias] + [p for b in net64.blocks for l in b.linear_layers
                                                                      for p in (l.weight, l.bias)]
    ref = torch.autograd.grad(h, [xid] + params, gh[:, :hidden].double())
    netd = net.to(device)
    assert netd.hip_hidden_backward_supported()
    gxid, gw0, gwb, gb = ops.resnet_hidden_backward(x.to(device), gh.to(device), ids.to(device), netd.hidden_backward_packed(),
                                                    in_f, blocks)
    assert _relerr(gxid, ref[0]) <= 2e-4
    assert _relerr(gw0[:hidden], ref[1]) <= 2e-4 and _relerr(gb[0, :hidden], ref[2]) <= 2e-4
    for i in range(2 * blocks):
        assert _relerr(gwb[i, :hidden, :hidden], ref[3 + 2 * i]) <= 2e-4, i
        assert _relerr(gb[1 + i, :hidden], ref[4 + 2 * i]) <= 2e-4, i
    if hidden < 64:      # zero-padded units: no gradient leaks into them
        assert float(gw0[hidden:].abs().max()) == 0.0 and float(gb[:, hidden:].abs().max()) == 0.0


@pytest.mark.parametrize("k,tails,d_t,hidden", [(8, "linear", 32, 64), (10, "linear", 70, 64), (10, None, 5, 20), (4, "linear", 3, 64)])
def test_device_pack_equals_the_tensor_op_packers(k, tails, d_t, hidden, device):
    """fc_pack_fragments (one launch; training re-packs every step) against the torch packers the inference path caches:
    bit-identical fragments, scales and biases, for the final layer (forward + W^T) and the hidden stack."""
    torch.manual_seed(k + d_t)
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    w = (torch.randn(d_t * p, hidden) * torch.logspace(-3, 1, d_t * p).unsqueeze(1)).to(device)
    b = torch.randn(d_t * p, device=device)
    spec = [(slice(lo * p, min(lo + 32, d_t) * p), torch.arange(lo, min(lo + 32, d_t), dtype=torch.int32, device=device))
            for lo in range(0, d_t, 32)]
    pack, chunks = ops.device_pack_final_layer(w, b, k, tails, spec)
    pack.run()
    for (w_frag, w_un, bias_pad, wt_frag, _, rows) in chunks:
        ref_f, ref_un, ref_b = ops.pack_final_layer_general(w[rows], b[rows], k, tails, 64)
        ref_t = ops.pack_final_layer_transposed(w[rows], k, tails)
        assert torch.equal(w_frag, ref_f) and torch.equal(w_un, ref_un) and torch.equal(bias_pad, ref_b)
        assert torch.equal(wt_frag, ref_t)
    net = nets.ResidualNet(min(d_t, 60), 8, hidden_features=hidden, num_blocks=2).to(device)
    pack2, packed = ops.device_pack_resnet_hidden_backward(net)
    pack2.run()
    ref = ops.pack_resnet_hidden_backward(net)
    for got, want in zip(packed[:4], ref[:4]):
        assert torch.equal(got.reshape(-1), want.reshape(-1))
    assert packed[4] == ref[4]
    with torch.no_grad():       # optimizers update in place: a re-run picks the new values up
        for prm in net.parameters():
            prm.mul_(0.5)
    pack2.run()
    ref = ops.pack_resnet_hidden_backward(net)
    for got, want in zip(packed[:4], ref[:4]):
        assert torch.equal(got.reshape(-1), want.reshape(-1))


def test_fused_linear_backward_repeated_fresh_batches(device):
    """gW / gb / gh / gx at N = 4096 (one 32-row tile per workgroup on every CU) for eight independently drawn batches, fresh
    tensors each time.  Regression test: with spilled fragment addresses the dw role returned, now and then, a wrong
    [dim, widths] slice of gW for one tile (tools/probe/cold_launch_gw.py, tools/probe/fuzz_backward.py found it)."""
    k, tails, d, d_t, n, hidden = 8, "linear", 64, 32, 4096, 64
    p = 3 * k - 1
    kw = dict(wh_divisor=float(hidden) ** 0.5)
    cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)
    for trial in range(9):
        torch.manual_seed(100 + trial)
        if trial == 8:     # several tiles per workgroup, magnitudes growing along the batch: the per-feature scale of the gW
            n = 3 * 256 * 32 + 64     # accumulators is lowered (and the accumulators rescaled) from tile to tile
        x = torch.randn(n, d) * 1.5
        h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
        if trial == 8:
            h *= torch.logspace(-3, 0.5, n).unsqueeze(1)
            x[:, ::2] *= torch.linspace(0.05, 1.0, n).unsqueeze(1)
        w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
        b = torch.randn(d_t * p) * 0.3
        gy, gl = torch.randn(n, d), torch.randn(n)
        x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
        rows = (h64 @ w64.T + b64).view(n, d_t, p)
        out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
        y64 = x64.clone().index_copy(1, cols.long(), out)
        loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
        gx_ref, gh_ref, gw_ref, gb_ref = torch.autograd.grad(loss, (x64, h64, w64, b64))
        packed = ops.pack_final_layer_general(w.to(device), b.to(device), k, tails, 64)
        packed_t = ops.pack_final_layer_transposed(w.to(device), k, tails)
        gx, gh, gw, gb = ops.rq_fused_linear_backward(x.to(device), h.to(device), gy.to(device), gl.to(device), packed,
                                                      packed_t, cols.to(device), num_bins=k, tails=tails, tail_bound=3.0, **kw)
        assert _relerr(gw, gw_ref) <= 2e-4, trial
        assert _relerr(gb, gb_ref) <= 2e-4 and _relerr(gh, gh_ref) <= 2e-4 and _relerr(gx, gx_ref) <= 2e-4, trial
