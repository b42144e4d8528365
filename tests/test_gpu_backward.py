"""Gradients through the HIP RQ-spline path (fc_rq_spline_backward behind torch.autograd) against
torch.autograd on the CPU oracle -- the gradient-consistency check of SURVEY section 8(f) #3."""
import copy

import pytest
import torch

from _util import build_case, maxdiff
from flowconductor_amd import ops
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_grads(x, params, cols, gy, gl, dtype, **kw):
    """dL/dx[:, cols], dL/dparams for L = sum(gy * y) + sum(gl * logabsdet) by autograd on the oracle."""
    n, d_t = x.shape[0], len(cols)
    xt = x[:, cols].to(dtype).clone().requires_grad_(True)
    p = params.to(dtype).clone().requires_grad_(True)
    rows = p.view(n, d_t, -1) * 1.0     # (rq_from_rows divides slices of its argument in place)
    out, lad = O.rq_from_rows(xt, rows, kw["num_bins"], kw.get("tails"), kw.get("tail_bound", 1.0), False,
                              wh_divisor=kw.get("wh_divisor"), box=kw.get("box", (0.0, 1.0, 0.0, 1.0)),
                              enable_identity_init=kw.get("enable_identity_init", False))
    loss = (out * gy[:, cols].to(dtype)).sum() + (lad.sum(dim=1) * gl.to(dtype)).sum()
    loss.backward()
    return xt.grad, p.grad


@pytest.mark.parametrize("case", ["k8_tails", "k5_tails_generic", "k8_box_identity_init", "k16_tails"])
def test_rq_spline_backward_matches_oracle_autograd(case, device):
    torch.manual_seed(11)
    n, d = 512, 12
    cols = [1, 2, 4, 5, 7, 8, 9, 11]
    d_t = len(cols)
    if case == "k8_tails":
        kw = dict(num_bins=8, tails="linear", tail_bound=3.0, wh_divisor=8.0)
        x = torch.randn(n, d) * 1.8     # some inputs beyond the tail bound
    elif case == "k5_tails_generic":
        kw = dict(num_bins=5, tails="linear", tail_bound=2.0, wh_divisor=1.0)
        x = torch.randn(n, d) * 1.2
    elif case == "k16_tails":
        kw = dict(num_bins=16, tails="linear", tail_bound=4.0, wh_divisor=4.0)
        x = torch.randn(n, d) * 2.0
    else:
        kw = dict(num_bins=8, tails=None, left=-1.2, right=1.2, bottom=-1.2, top=1.2, enable_identity_init=True)
        x = (torch.rand(n, d) * 2 - 1) * 1.15
    mult = 3 * kw["num_bins"] - 1 if kw.get("tails") else 3 * kw["num_bins"] + 1
    params = torch.randn(n, d_t * mult)
    gy, gl = torch.randn(n, d), torch.randn(n)
    okw = dict(kw)
    if kw.get("tails") is None:
        okw["box"] = (kw["left"], kw["right"], kw["bottom"], kw["top"])
        okw["wh_divisor"] = None
    ref_gx, ref_gp = _oracle_grads(x, params, cols, gy, gl, torch.float64, **okw)
    f32_gx, f32_gp = _oracle_grads(x, params, cols, gy, gl, torch.float32, **okw)

    xd = x.to(device).requires_grad_(True)
    pd = params.to(device).requires_grad_(True)
    cd = torch.tensor(cols, dtype=torch.int32, device=device)
    y, lad = ops.rq_spline_autograd(xd, pd, cd, **kw)
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    gx, gp = xd.grad.cpu(), pd.grad.cpu()
    # identity columns: the bijector copies them, so their gradient is the upstream one
    idc = [c for c in range(d) if c not in cols]
    assert torch.equal(gx[:, idc], gy[:, idc])
    # tolerance: 1e-4 of the gradient scale + 8 x what float32 autograd on the oracle loses against float64
    sx, sp = float(ref_gx.abs().max()), float(ref_gp.abs().max())
    assert maxdiff(gx[:, cols].double(), ref_gx) <= 1e-4 * sx + 8 * maxdiff(f32_gx.double(), ref_gx)
    assert maxdiff(gp.double(), ref_gp) <= 1e-4 * sp + 8 * maxdiff(f32_gp.double(), ref_gp)


def test_coupling_layer_trains_through_hip_path(device):
    """Parameter gradients of one RQ coupling layer (conditioner on PyTorch autograd, spline forward + backward in
    HIP) against the oracle walked by torch.autograd on the CPU."""
    t_cpu, _ = build_case("rq_coupling_linear_tails_d64_k8_h64")
    t_gpu = copy.deepcopy(t_cpu).to(device).train()
    t_cpu = t_cpu.double().train()
    x = torch.randn(256, 64, generator=torch.Generator().manual_seed(3)) * 1.5
    gy = torch.randn(256, 64, generator=torch.Generator().manual_seed(4))
    gl = torch.randn(256, generator=torch.Generator().manual_seed(5))

    y_ref, lad_ref = O.transform_apply(t_cpu, x.double())
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()

    from flowconductor_amd import options

    # (the fused training path, which this layer shape would take by default, has its own tests in
    #  tests/test_gpu_fused_backward.py; here: conditioner on PyTorch autograd + the stand-alone spline backward kernel)
    with options.override(fused_training=False), ops.KernelTimer("fc_rq_spline_backward") as timer:
        y, lad = t_gpu(x.to(device))
        ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert len(timer.pairs) == 1, "the backward kernel did not run"
    assert maxdiff(y.detach().cpu().double(), y_ref.detach()) <= 2e-5 * float(y_ref.detach().abs().max())
    for (name, p_ref), (_, p) in zip(t_cpu.named_parameters(), t_gpu.named_parameters()):
        assert p.grad is not None, name
        scale = max(1e-6, float(p_ref.grad.abs().max()))
        assert maxdiff(p.grad.cpu().double(), p_ref.grad) <= 2e-4 * scale, name


def test_flow_loss_backward_step(device):
    """-log_prob(x).mean().backward() + an SGD step through a 4-layer HIP flow lowers the loss (toy_2d.py:57-68)."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(0)
    layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(8, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=16, num_blocks=1), num_bins=6, tails="linear",
        tail_bound=3.0) for i in range(4)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([8])).to(device).train()
    x = (torch.randn(2048, 8) * 0.5 + 0.7).to(device)
    opt = torch.optim.SGD(flow.parameters(), lr=0.05)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("act", [ops.AFFINE_SIGMOID_PLUS2, ops.AFFINE_SOFTPLUS_CLAMP3, ops.AFFINE_SCALE_GIVEN,
                                 ops.AFFINE_ADDITIVE, ops.AFFINE_MAF_SOFTPLUS, ops.AFFINE_SHIFT_TANH2,
                                 ops.AFFINE_SCALE_SOFTPLUS])
def test_affine_backward_matches_torch_autograd(act, device):
    """fc_affine_backward against autograd on the same formulas written with torch ops (coupling.py:234-252,
    autoregressive.py:97-129, 164-196, conditional.py:155-272)."""
    import torch.nn.functional as F

    torch.manual_seed(act)
    n, d = 300, 10
    cols = [0, 2, 3, 5, 8, 9]
    d_t = len(cols)
    one = act in (ops.AFFINE_ADDITIVE, ops.AFFINE_SHIFT_TANH2, ops.AFFINE_SCALE_SOFTPLUS)
    x = torch.randn(n, d)
    params = torch.randn(n, d_t if one else 2 * d_t) * 1.5
    if act == ops.AFFINE_SCALE_GIVEN:
        params[:, d_t:] = params[:, d_t:].abs() + 0.2
    gy, gl = torch.randn(n, d), torch.randn(n)

    xt = x[:, cols].double().requires_grad_(True)
    p = params.double().requires_grad_(True)
    if act == ops.AFFINE_SIGMOID_PLUS2:
        shift, s = p[:, :d_t], torch.sigmoid(p[:, d_t:] + 2) + 1e-3
    elif act == ops.AFFINE_SOFTPLUS_CLAMP3:
        shift, s = p[:, :d_t], torch.clamp(F.softplus(p[:, d_t:]) + 1e-3, 0, 3)
    elif act == ops.AFFINE_SCALE_GIVEN:
        shift, s = p[:, :d_t], p[:, d_t:]
    elif act == ops.AFFINE_MAF_SOFTPLUS:
        v = p.view(n, d_t, 2)
        shift, s = v[..., 1], F.softplus(v[..., 0]) + 1e-3
    elif act == ops.AFFINE_SCALE_SOFTPLUS:
        shift, s = torch.zeros_like(p), F.softplus(p) + 1e-5
    elif act == ops.AFFINE_SHIFT_TANH2:
        shift, s = 2 * torch.tanh(p), torch.ones_like(p)
    else:
        shift, s = p, torch.ones_like(p)
    y, lad = xt * s + shift, torch.log(s).sum(dim=1)
    ((y * gy[:, cols].double()).sum() + (lad * gl.double()).sum()).backward()

    xd, pd = x.to(device).requires_grad_(True), params.to(device).requires_grad_(True)
    yd, ladd = ops.affine_coupling(xd, pd, torch.tensor(cols, dtype=torch.int32, device=device), activation=act)
    ((yd * gy.to(device)).sum() + (ladd * gl.to(device)).sum()).backward()
    idc = [c for c in range(d) if c not in cols]
    assert torch.equal(xd.grad.cpu()[:, idc], gy[:, idc])
    assert maxdiff(xd.grad.cpu()[:, cols].double(), xt.grad) <= 2e-5 * max(1.0, float(xt.grad.abs().max()))
    assert maxdiff(pd.grad.cpu().double(), p.grad) <= 2e-5 * max(1.0, float(p.grad.abs().max()))


def test_readme_maf_flow_trains(device):
    """The reference README flow (examples/toy_2d.py: MaskedAffineAutoregressiveTransform + RandomPermutation,
    D = 2, hidden 4) takes optimisation steps through the HIP path: affine and permutation gradients."""
    from flowconductor_amd import distributions, flows, transforms

    torch.manual_seed(0)
    layers = []
    for _ in range(2):
        layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
        layers.append(transforms.RandomPermutation(features=2))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).to(device).train()
    x = (torch.randn(4096, 2) * torch.tensor([0.3, 1.5]) + torch.tensor([1.0, -0.5])).to(device)
    opt = torch.optim.Adam(flow.parameters(), lr=0.05)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("kind", ["rq_coupling", "affine_coupling", "maf"])
def test_inverse_direction_gradients_match_oracle_autograd(kind, device):
    """Gradients through ``inverse`` (sampling direction; reverse-KL / variational training): the kernels only have
    backward kernels for the forward direction, the inverse is made differentiable by the implicit-function step of
    ``ops._inverse_through_forward``.  Parameter and input gradients of L = sum(gy * y) + sum(gl * logabsdet) against
    torch.autograd walking the oracle's inverse in float64."""
    from flowconductor_amd import transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(41)
    d, n = 8, 300

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=16, num_blocks=1)

    if kind == "rq_coupling":
        t = transforms.PiecewiseRationalQuadraticCouplingTransform(utils.create_alternating_binary_mask(d), net,
                                                                   num_bins=6, tails="linear", tail_bound=3.0)
    elif kind == "affine_coupling":
        t = transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(d), net)
    else:
        t = transforms.MaskedAffineAutoregressiveTransform(d, 16, num_blocks=1)
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    t_ref = copy.deepcopy(t).double().train()
    t_gpu = copy.deepcopy(t).to(device).train()
    z = torch.randn(n, d) * 1.2
    gy, gl = torch.randn(n, d), torch.randn(n)

    z_ref = z.double().requires_grad_(True)
    y_ref, lad_ref = O.transform_apply(t_ref, z_ref, inverse=True)
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()

    z_gpu = z.to(device).requires_grad_(True)
    y, lad = t_gpu.inverse(z_gpu)
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()

    assert maxdiff(y.detach(), y_ref.detach()) <= 3e-5 * max(1.0, float(y_ref.detach().abs().max()))
    assert maxdiff(lad.detach(), lad_ref.detach()) <= 3e-4
    assert maxdiff(z_gpu.grad, z_ref.grad) <= 3e-4 * max(1.0, float(z_ref.grad.abs().max()))
    for (name, p_ref), (_, p) in zip(t_ref.named_parameters(), t_gpu.named_parameters()):
        assert p.grad is not None, name
        scale = max(1e-6, float(p_ref.grad.abs().max()))
        assert maxdiff(p.grad.cpu().double(), p_ref.grad) <= 5e-4 * scale, (name, maxdiff(p.grad.cpu().double(), p_ref.grad), scale)


def test_sampling_with_autograd_enabled_and_reverse_kl_step(device):
    """``flow.sample`` outside ``torch.no_grad()`` (the reference allows it) and one reverse-KL step: samples and their
    log-density from ``sample_and_log_prob`` carry gradients to the flow's parameters."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(43)
    d = 6
    layers = []
    for i in range(3):
        layers.append(transforms.PiecewiseRationalQuadraticCouplingTransform(
            utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
            lambda a, b: nets.ResidualNet(a, b, hidden_features=16, num_blocks=1), num_bins=6, tails="linear",
            tail_bound=3.0))
        layers.append(transforms.ReversePermutation(d))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([d])).to(device).train()
    s = flow.sample(500)
    assert s.shape == (500, d) and s.requires_grad
    opt = torch.optim.SGD(flow.parameters(), lr=0.02)
    target_mean = torch.full((d,), 1.5, device=device)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        samples, log_q = flow.sample_and_log_prob(2048)
        log_p = -0.5 * ((samples - target_mean) ** 2).sum(dim=1)       # unnormalised N(1.5, I) target
        loss = (log_q - log_p).mean()                                   # reverse KL up to a constant
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in flow.parameters() if p.requires_grad)
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


def test_nsf_style_stack_trains_through_actnorm_and_lu(device):
    """[ActNorm, LULinear, RQ coupling] x 2 (the NSF layer pattern): parameter gradients of -log_prob.mean() against
    torch.autograd on the oracle in float64 -- the ActNorm / LU kernels sit behind autograd nodes as well."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(47)
    d, n = 6, 512
    layers = []
    for i in range(2):
        layers += [transforms.ActNorm(d), transforms.LULinear(d, identity_init=False),
                   transforms.PiecewiseRationalQuadraticCouplingTransform(
                       utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
                       lambda a, b: nets.ResidualNet(a, b, hidden_features=16, num_blocks=1), num_bins=6,
                       tails="linear", tail_bound=3.0)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([d]))
    with torch.no_grad():
        for m in flow.modules():
            if isinstance(m, transforms.ActNorm):
                m.initialized.fill_(True)
                m.log_scale.normal_(0, 0.3)
                m.shift.normal_(0, 0.5)
    ref = copy.deepcopy(flow).double().train()
    gpu = copy.deepcopy(flow).to(device).train()
    x = torch.randn(n, d)

    def reference_log_prob(flow64, v):
        # the oracle restates ActNorm / LU for inference (detached parameters): their two formulas in torch ops here,
        # normalization.py:171-204 and lu.py:56-68; the coupling layers through the oracle
        total = v.new_zeros(v.shape[0])
        for t in flow64._transform._transforms:
            if isinstance(t, transforms.ActNorm):
                v, lad = torch.exp(t.log_scale) * v + t.shift, t.log_scale.sum().expand(v.shape[0])
            elif isinstance(t, transforms.LULinear):
                lower, upper = t._create_lower_upper()
                v, lad = torch.nn.functional.linear(v, lower @ upper, t.bias), torch.log(t.upper_diag).sum().expand(v.shape[0])
            else:
                v, lad = O.transform_apply(t, v)
            total = total + lad
        return O.standard_normal_log_prob(v) + total

    loss_ref = -reference_log_prob(ref, x.double()).mean()
    loss_ref.backward()
    loss = -gpu.log_prob(x.to(device)).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    for (name, p_ref), (_, p) in zip(ref.named_parameters(), gpu.named_parameters()):
        assert p.grad is not None, name
        scale = max(1e-5, float(p_ref.grad.abs().max()))
        assert maxdiff(p.grad.cpu().double(), p_ref.grad) <= 1e-3 * scale, (name, maxdiff(p.grad.cpu().double(), p_ref.grad), scale)
    # sampling direction with gradients through LU / ActNorm inverses
    s, lq = gpu.sample_and_log_prob(256)
    (s.sum() + lq.sum()).backward()


def test_batch_shared_cdfs_train(device):
    """Batch-shared linear / quadratic / cubic CDFs under autograd (expanded rows through the spline node): gradients of
    the shared parameters against central finite differences of the oracle's float64 forward."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(73)
    d, n = 5, 200
    x = torch.rand(n, d) * 0.9 + 0.05
    gy, gl = torch.randn(n, d).double(), torch.randn(n).double()
    for t in (T.PiecewiseLinearCDF([d], num_bins=4), T.PiecewiseQuadraticCDF([d], num_bins=4),
              T.PiecewiseCubicCDF([d], num_bins=4)):
        def loss64(module):
            with torch.no_grad():
                y, lad = O.transform_apply(module, x.double())
            return float((y * gy).sum() + (lad * gl).sum())

        tg = copy.deepcopy(t).to(device).train()
        y, lad = tg(x.to(device))
        ((y * gy.float().to(device)).sum() + (lad * gl.float().to(device)).sum()).backward()
        t64 = copy.deepcopy(t).double()
        g = torch.Generator().manual_seed(2)
        for (name, p64), (_, pg) in zip(t64.named_parameters(), tg.named_parameters()):
            assert pg.grad is not None, (type(t).__name__, name)
            flat, gflat = p64.data.view(-1), pg.grad.cpu().double().view(-1)
            for idx in torch.randperm(flat.numel(), generator=g)[:3].tolist():
                old, h = float(flat[idx]), 1e-5
                flat[idx] = old + h
                up = loss64(t64)
                flat[idx] = old - h
                down = loss64(t64)
                flat[idx] = old
                fd = (up - down) / (2 * h)
                assert abs(float(gflat[idx]) - fd) <= 3e-3 * max(1.0, abs(fd)) + 2e-3, (type(t).__name__, name, float(gflat[idx]), fd)


def test_realnvp_with_batch_norm_trains(device):
    """SimpleRealNVP-style stack with BatchNorm between the coupling layers in training mode: the batch-norm layer is
    the reference's torch expression under autograd (cross-batch statistics), everything else the HIP path; loss and
    parameter gradients against the same stack evaluated with torch ops in float64."""
    from flowconductor_amd import transforms, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(53)
    d, n = 6, 400
    layers = []
    for i in range(2):
        layers += [transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
                                                      lambda a, b: nets.ResidualNet(a, b, hidden_features=16, num_blocks=1)),
                   transforms.BatchNorm(d)]
    stack = transforms.CompositeTransform(layers)
    ref = copy.deepcopy(stack).double().train()
    gpu = copy.deepcopy(stack).to(device).train()
    x = torch.randn(n, d) * 1.3 + 0.4

    v, total = x.double(), torch.zeros(n, dtype=torch.float64)
    for t in ref._transforms:
        if isinstance(t, transforms.BatchNorm):
            mean, var = v.mean(0), v.var(0)
            v, lad = t.weight * ((v - mean) / torch.sqrt(var + t.eps)) + t.bias, (torch.log(t.weight) - 0.5 * torch.log(var + t.eps)).sum().expand(n)
        else:
            v, lad = O.transform_apply(t, v)
        total = total + lad
    loss_ref = (v ** 2).sum(1).mean() - total.mean()
    loss_ref.backward()

    y, lad = gpu(x.to(device))
    loss = (y ** 2).sum(1).mean() - lad.mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    for (name, p_ref), (_, p) in zip(ref.named_parameters(), gpu.named_parameters()):
        if p_ref.grad is None:
            continue
        scale = max(1e-5, float(p_ref.grad.abs().max()))
        # (some gradients are exactly zero in exact arithmetic, e.g. of a bias that the next layer's mean
        # subtraction removes: absolute floor at the float32 noise of the loss)
        assert p.grad is not None and maxdiff(p.grad.cpu().double(), p_ref.grad) <= 1e-3 * scale + 1e-6, name
    bn = [t for t in gpu._transforms if isinstance(t, transforms.BatchNorm)][0]
    assert float(bn.running_var.abs().sum()) > 0      # running statistics were updated
    with torch.no_grad():
        gpu.eval()
        y_eval, _ = gpu(x.to(device))                 # eval mode, no autograd: the HIP kernel with running statistics
        back, _ = gpu.inverse(y_eval)
    assert maxdiff(back, x) <= 1e-4 * max(1.0, float(x.abs().max()))


def test_sum_of_sigmoids_gradients(device):
    """FlowConductor's sum-of-sigmoids bijector under autograd: HIP forward kernel, gradients from the same map in torch
    ops.  Module-owned parameters (batch-shared) against autograd on the oracle's formula in float64; the masked
    autoregressive form against autograd through the oracle; the inverse direction runs and carries gradients."""
    from flowconductor_amd import transforms

    torch.manual_seed(59)
    d, s_, n = 5, 6, 300
    t = transforms.SumOfSigmoids(d, n_sigmoids=s_)
    with torch.no_grad():
        t.shift_preact.normal_(0, 0.5)
        t.log_scale_preact.normal_(0, 0.5)
        t.raw_softmax.normal_(0, 0.5)
    x = torch.randn(n, d)
    gy, gl = torch.randn(n, d), torch.randn(n)
    leaves = [p.detach().double().clone().requires_grad_(True)
              for p in (t.shift_preact, t.log_scale_preact, t.raw_softmax, t.extended_softplus.shift)]
    y_ref, lad_ref = O.sos_forward(x.double(), leaves[0], leaves[1], leaves[2], leaves[3])
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    tg = copy.deepcopy(t).to(device)
    y, lad = tg(x.to(device))
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 2e-5 * max(1.0, float(y_ref.detach().abs().max()))
    for got, ref in zip((tg.shift_preact, tg.log_scale_preact, tg.raw_softmax, tg.extended_softplus.shift), leaves):
        scale = max(1e-5, float(ref.grad.abs().max()))
        assert got.grad is not None and maxdiff(got.grad.cpu().double().reshape(ref.grad.shape), ref.grad) <= 1e-3 * scale + 1e-6

    ar = transforms.MaskedSumOfSigmoidsTransform(d, 16, n_sigmoids=s_, num_blocks=1)
    ref_ar = copy.deepcopy(ar).double().train()
    gpu_ar = copy.deepcopy(ar).to(device).train()
    yr, lr = O.transform_apply(ref_ar, x.double())
    ((yr * gy.double()).sum() + (lr * gl.double()).sum()).backward()
    yg, lg = gpu_ar(x.to(device))
    ((yg * gy.to(device)).sum() + (lg * gl.to(device)).sum()).backward()
    for (name, p_ref), (_, p) in zip(ref_ar.named_parameters(), gpu_ar.named_parameters()):
        if p_ref.grad is None:
            continue
        scale = max(1e-5, float(p_ref.grad.abs().max()))
        assert p.grad is not None and maxdiff(p.grad.cpu().double(), p_ref.grad) <= 1e-3 * scale + 1e-6, name

    z = torch.randn(64, d, device=device).requires_grad_(True)
    xi, ladi = tg.inverse(z)
    (xi.sum() + ladi.sum()).backward()
    assert z.grad is not None and torch.isfinite(z.grad).all() and tg.shift_preact.grad is not None


@pytest.mark.parametrize("kind", ["householder", "planar", "sylvester"])
def test_orthogonal_and_planar_family_gradients_by_finite_differences(kind, device):
    """Householder / planar / Sylvester layers under autograd (kernel forward, gradients from the same map in torch ops)
    against central finite differences of the ORACLE's float64 forward -- an independent check of those expressions."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(61)
    d, n = 6, 200
    t = {"householder": lambda: T.HouseholderSequence(d, 3), "planar": lambda: T.PlanarTransform(d),
         "sylvester": lambda: T.SylvesterTransform(d, num_householder=2, device="cpu")}[kind]()
    with torch.no_grad():
        for p in t.parameters():
            p.add_(torch.randn_like(p) * 0.3)
    x = torch.randn(n, d)
    gy, gl = torch.randn(n, d).double(), torch.randn(n).double()

    def loss64(module):
        with torch.no_grad():
            y, lad = O.transform_apply(module, x.double())
        return float((y * gy).sum() + (lad * gl).sum())

    tg = copy.deepcopy(t).to(device).train()
    xg = x.to(device).requires_grad_(True)
    y, lad = tg(xg)
    ((y * gy.float().to(device)).sum() + (lad * gl.float().to(device)).sum()).backward()
    t64 = copy.deepcopy(t).double()
    g = torch.Generator().manual_seed(1)
    for (name, p64), (_, pg) in zip(t64.named_parameters(), tg.named_parameters()):
        assert pg.grad is not None, name
        flat, gflat = p64.data.view(-1), pg.grad.cpu().double().view(-1)
        for idx in torch.randperm(flat.numel(), generator=g)[:4].tolist():
            old, h = float(flat[idx]), 1e-5
            flat[idx] = old + h
            up = loss64(t64)
            flat[idx] = old - h
            down = loss64(t64)
            flat[idx] = old
            fd = (up - down) / (2 * h)
            assert abs(float(gflat[idx]) - fd) <= 2e-3 * max(1.0, abs(fd)) + 1e-3, (name, idx, float(gflat[idx]), fd)
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_rq_cdf_trains(device):
    """Batch-shared RQ-spline CDF under autograd: gradients of the shared parameters against torch.autograd on the
    oracle's spline with the row expanded, float64."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(67)
    d, k, n = 5, 6, 300
    t = T.PiecewiseRationalQuadraticCDF([d], num_bins=k, tails="linear", tail_bound=3.0)
    x = torch.randn(n, d) * 1.3
    gy, gl = torch.randn(n, d), torch.randn(n)
    leaves = [p.detach().double().clone().requires_grad_(True)
              for p in (t.unnormalized_widths, t.unnormalized_heights, t.unnormalized_derivatives)]
    rows = torch.cat(leaves, dim=-1).reshape(1, d, -1).expand(n, d, 3 * k - 1) * 1.0
    out, lad_e = O.rq_from_rows(x.double(), rows, k, "linear", 3.0, False)
    ((out * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()).backward()
    tg = copy.deepcopy(t).to(device).train()
    y, lad = tg(x.to(device))
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), out.detach()) <= 2e-5 * max(1.0, float(out.detach().abs().max()))
    for got, ref in zip((tg.unnormalized_widths, tg.unnormalized_heights, tg.unnormalized_derivatives), leaves):
        scale = max(1e-5, float(ref.grad.abs().max()))
        assert got.grad is not None and maxdiff(got.grad.cpu().double(), ref.grad) <= 1e-3 * scale + 1e-6


@pytest.mark.parametrize("kind", ["linear", "quadratic", "quadratic_tails", "cubic", "maf_quadratic", "maf_cubic"])
def test_sibling_spline_layers_train(kind, device):
    """Linear / quadratic / cubic spline layers under autograd: kernel forward behind a node whose gradients come from
    the same spline in torch ops; parameter and input gradients against torch.autograd through the oracle in float64;
    the inverse direction carries gradients too."""
    from flowconductor_amd import transforms as T, utils
    from flowconductor_amd.nn import nets

    torch.manual_seed(71)
    d, n, k = 6, 300, 5
    mask = utils.create_alternating_binary_mask(d)

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=16, num_blocks=1)

    unit = kind in ("linear", "quadratic", "cubic", "maf_cubic")
    t = {"linear": lambda: T.PiecewiseLinearCouplingTransform(mask, net, num_bins=k),
         "quadratic": lambda: T.PiecewiseQuadraticCouplingTransform(mask, net, num_bins=k),
         "quadratic_tails": lambda: T.PiecewiseQuadraticCouplingTransform(mask, net, num_bins=k, tails="linear", tail_bound=2.0),
         "cubic": lambda: T.PiecewiseCubicCouplingTransform(mask, net, num_bins=k),
         "maf_quadratic": lambda: T.MaskedPiecewiseQuadraticAutoregressiveTransform(k, d, 16, num_blocks=1, tails="linear", tail_bound=3.0),
         "maf_cubic": lambda: T.MaskedPiecewiseCubicAutoregressiveTransform(k, d, 16, num_blocks=1)}[kind]()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    x = torch.rand(n, d) * 0.9 + 0.05 if unit else torch.randn(n, d) * 1.3
    gy, gl = torch.randn(n, d), torch.randn(n)
    ref = copy.deepcopy(t).double().train()
    x_ref = x.double().requires_grad_(True)
    y_ref, lad_ref = O.transform_apply(ref, x_ref)
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    gpu = copy.deepcopy(t).to(device).train()
    x_gpu = x.to(device).requires_grad_(True)
    y, lad = gpu(x_gpu)
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 3e-5 * max(1.0, float(y_ref.detach().abs().max()))
    assert maxdiff(x_gpu.grad, x_ref.grad) <= 1e-3 * max(1.0, float(x_ref.grad.abs().max()))
    for (name, p_ref), (_, p) in zip(ref.named_parameters(), gpu.named_parameters()):
        if p_ref.grad is None:
            continue
        scale = max(1e-5, float(p_ref.grad.abs().max()))
        assert p.grad is not None and maxdiff(p.grad.cpu().double(), p_ref.grad) <= 2e-3 * scale + 1e-6, name
    z = y.detach().clone().requires_grad_(True)
    back, lad_inv = gpu.inverse(z)
    (back.sum() + lad_inv.sum()).backward()
    assert z.grad is not None and torch.isfinite(z.grad).all()
    assert maxdiff(back.detach(), x) <= 1e-3 * max(1.0, float(x.abs().max()))


@pytest.mark.parametrize("kind", ["rq", "sos", "linear_spline", "shift", "scale"])
def test_hyper_network_transforms_train(kind, device):
    """Conditional (hyper-network) transforms under autograd: gradients of the hyper-network's parameters against
    torch.autograd through the oracle in float64."""
    from flowconductor_amd import transforms as T

    torch.manual_seed(79)
    d, ctx_f, n = 5, 4, 300
    t = {"rq": lambda: T.ConditionalPiecewiseRationalQuadraticTransform(d, 16, ctx_f, num_bins=6, tails="linear", tail_bound=3.0),
         "sos": lambda: T.ConditionalSumOfSigmoidsTransform(d, 16, ctx_f, n_sigmoids=5),
         "linear_spline": lambda: T.PiecewiseLinearConditionalTransform(5, d, 16, ctx_f),
         "shift": lambda: T.ConditionalShiftTransform(d, 16, ctx_f),
         "scale": lambda: T.ConditionalScaleTransform(d, 16, ctx_f)}[kind]()
    x = (torch.randn(n, d) * 1.2).clamp(-3.5, 3.5)       # (the conditional linear spline lives on the box [-4, 4])
    c = torch.randn(n, ctx_f)
    gy, gl = torch.randn(n, d), torch.randn(n)
    ref = copy.deepcopy(t).double().train()
    y_ref, lad_ref = O.transform_apply(ref, x.double(), c.double())
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    gpu = copy.deepcopy(t).to(device).train()
    y, lad = gpu(x.to(device), c.to(device))
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 3e-5 * max(1.0, float(y_ref.detach().abs().max()))
    checked = 0
    for (name, p_ref), (_, p) in zip(ref.named_parameters(), gpu.named_parameters()):
        if p_ref.grad is None:
            continue
        scale = max(1e-5, float(p_ref.grad.abs().max()))
        assert p.grad is not None and maxdiff(p.grad.cpu().double(), p_ref.grad) <= 2e-3 * scale + 1e-6, name
        checked += 1
    assert checked > 0


@pytest.mark.parametrize("d,k,reverse", [(6, 3, False), (64, 8, True), (100, 11, False), (200, 17, True), (384, 2, False)])
def test_householder_backward_kernel_matches_float64_autograd(d, k, reverse, device):
    """fc_householder_backward (shared q, walks the saved output back through the involutions, more reflections than one
    register chunk holds) against float64 autograd through the oracle's reflections."""
    torch.manual_seed(71 + d)
    n = 777
    x, q, gy = torch.randn(n, d), torch.randn(k, d), torch.randn(n, d)
    x64 = x.double().requires_grad_(True)
    q64 = q.double().requires_grad_(True)
    y_ref = O.householder_apply(x64, q64.flip(0) if reverse else q64)
    (y_ref * gy.double()).sum().backward()
    xg, qg = x.to(device).requires_grad_(True), q.to(device).requires_grad_(True)
    y, lad = ops.householder_autograd(xg, qg, reverse=reverse)
    assert type(y.grad_fn).__name__ == "_HouseholderFunctionBackward"
    (y * gy.to(device)).sum().backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 1e-5 * float(y_ref.detach().abs().max()) * k
    assert float(lad.abs().max()) == 0.0
    assert maxdiff(xg.grad, x64.grad) <= 2e-5 * float(x64.grad.abs().max()) * k
    assert maxdiff(qg.grad, q64.grad) <= 2e-5 * float(q64.grad.abs().max()) * k + 1e-5


@pytest.mark.parametrize("d", [3, 64, 130, 500])
def test_planar_backward_kernel_matches_float64_autograd(d, device):
    """fc_planar_backward against float64 autograd on planar.py:30-49 written out (u_hat given)."""
    torch.manual_seed(73 + d)
    n = 1001
    x = torch.randn(n, d)
    w, u, b = torch.randn(1, d) / d ** 0.5, torch.randn(1, d) / d ** 0.5, torch.randn(1)
    gy, gl = torch.randn(n, d), torch.randn(n)
    leaves = [t.double().clone().requires_grad_(True) for t in (x, w, u, b)]
    x64, w64, u64, b64 = leaves
    t = torch.tanh(x64 @ w64.T + b64)
    y_ref = x64 + u64 * t
    lad_ref = torch.log(1e-7 + (1 + (u64 @ ((1 - t ** 2) * w64).T)).abs()).reshape(-1)
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    dev = [t.to(device).requires_grad_(True) for t in (x, w, u, b)]
    y, lad = ops.planar_autograd(*dev)
    assert type(y.grad_fn).__name__ == "_PlanarFunctionBackward"
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 1e-5 * float(y_ref.detach().abs().max())
    for got, ref, name in zip(dev, leaves, "xwub"):
        scale = max(1e-5, float(ref.grad.abs().max()))
        assert maxdiff(got.grad.reshape(ref.grad.shape), ref.grad) <= 1e-4 * scale + 1e-5, name
    # grad_logabsdet absent (only the outputs feed the loss)
    dev2 = [t.to(device).requires_grad_(True) for t in (x, w, u, b)]
    y2, _ = ops.planar_autograd(*dev2)
    (y2 * gy.to(device)).sum().backward()
    assert torch.isfinite(dev2[1].grad).all()


@pytest.mark.parametrize("s_,d", [(1, 1), (6, 5), (30, 3)])
def test_sos_backward_kernel_matches_float64_autograd(s_, d, device):
    """fc_sum_of_sigmoids_backward, per-sample raw rows, against float64 autograd on the oracle's formula."""
    torch.manual_seed(79 + s_)
    n = 515
    x, raw = torch.randn(n, d) * 3, torch.randn(n, d, 3 * s_ + 1)
    gy, gl = torch.randn(n, d), torch.randn(n)
    x64, r64 = x.double().requires_grad_(True), raw.double().requires_grad_(True)
    y_ref, lad_ref = O.sos_forward(x64, r64[..., :s_], r64[..., s_:2 * s_], r64[..., 2 * s_:3 * s_], r64[..., 3 * s_])
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    xg, rg = x.to(device).requires_grad_(True), raw.to(device).reshape(n, -1).requires_grad_(True)
    y, lad = ops.sum_of_sigmoids_autograd(xg, rg, s_)
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 2e-5 * max(1.0, float(y_ref.detach().abs().max()))
    assert maxdiff(xg.grad, x64.grad) <= 2e-4 * float(x64.grad.abs().max()) + 1e-6
    assert maxdiff(rg.grad.reshape(r64.grad.shape), r64.grad) <= 2e-4 * float(r64.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("d,m", [(6, 2), (64, 4), (130, 3)])
def test_sylvester_backward_matches_float64_autograd(d, m, device):
    """ops._SylvesterFunction (fc_householder_backward x 2, fc_sylvester_mid_backward, library GEMMs) against float64
    autograd on planar.py:144-166 written out with the oracle's reflections."""
    torch.manual_seed(83 + d)
    n = 515
    x, q = torch.randn(n, d), torch.randn(m, d)
    r1 = torch.triu(torch.randn(d, d)) / d ** 0.5
    r2 = torch.triu(torch.randn(d, d)) / d ** 0.5
    r1.diagonal().copy_(torch.rand(d) * 0.5 + 0.2)      # diag R1 diag R2 > -1: invertible map
    r2.diagonal().copy_(torch.rand(d) * 0.5 + 0.2)
    b = torch.randn(d) * 0.3
    gy, gl = torch.randn(n, d), torch.randn(n)
    leaves = [t.double().clone().requires_grad_(True) for t in (x, q, r1, r2, b)]
    x64, q64, r164, r264, b64 = leaves
    qtz = O.householder_apply(x64, q64.flip(0))
    act = torch.tanh(qtz @ r164.T + b64)
    y_ref = x64 + O.householder_apply(act @ r264.T, q64)
    lad_ref = torch.log(1 + (1 - act ** 2) * (torch.diag(r164) * torch.diag(r264))).sum(-1)
    ((y_ref * gy.double()).sum() + (lad_ref * gl.double()).sum()).backward()
    dev = [t.to(device).requires_grad_(True) for t in (x, q, r1, r2, b)]
    y, lad = ops.sylvester_autograd(*dev)
    assert type(y.grad_fn).__name__ == "_SylvesterFunctionBackward"
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach(), y_ref.detach()) <= 2e-5 * float(y_ref.detach().abs().max())
    assert maxdiff(lad.detach(), lad_ref.detach()) <= 2e-5 * max(1.0, float(lad_ref.detach().abs().max()))
    for got, ref, name in zip(dev, leaves, ("x", "q", "r1", "r2", "b")):
        scale = max(1e-5, float(ref.grad.abs().max()))
        assert maxdiff(got.grad.reshape(ref.grad.shape), ref.grad) <= 2e-4 * scale + 1e-5, name


@pytest.mark.parametrize("kind,k,tails", [("linear", 7, None), ("quadratic", 6, None), ("quadratic", 9, "linear"), ("cubic", 8, None),
                                          ("cubic", 5, "linear")])
def test_piecewise_spline_backward_kernel_matches_float64_autograd(kind, k, tails, device):
    """fc_piecewise_spline_backward (forward-mode derivative inside the kernel, one thread per element and parameter)
    against float64 autograd through the oracle's spline functions, raw per-sample rows, a subset of columns."""
    torch.manual_seed(89 + k)
    n, d = 400, 5
    cols = torch.tensor([0, 2, 3])
    code = {"linear": ops.SPLINE_LINEAR, "quadratic": ops.SPLINE_QUADRATIC, "cubic": ops.SPLINE_CUBIC}[kind]
    mult = ops.spline_multiplier(code, k, tails)
    bound = 2.0
    x = torch.randn(n, d) * 1.2 if tails else torch.rand(n, d) * 0.96 + 0.02
    raw = torch.randn(n, len(cols) * mult)
    gy, gl = torch.randn(n, d), torch.randn(n)
    kw = dict(kind=code, num_bins=k, tails=tails, tail_bound=bound)
    x64, r64 = x.double().requires_grad_(True), raw.double().requires_grad_(True)
    rows = r64.view(n, len(cols), mult)
    parts = {"linear": [rows], "quadratic": [rows[..., :k], rows[..., k:]],
             "cubic": [rows[..., :k], rows[..., k:2 * k], rows[..., 2 * k:2 * k + 1], rows[..., 2 * k + 1:]]}[kind]
    fn = {"linear": O.linear_spline, "quadratic": O.quadratic_spline, "cubic": O.cubic_spline}[kind]
    if tails:
        y_ref, lad_ref = O._unconstrained(fn, x64[:, cols], bound, tails, parts)
    else:
        y_ref, lad_ref = fn(x64[:, cols], *parts)
    ((y_ref * gy[:, cols].double()).sum() + (lad_ref.sum(-1) * gl.double()).sum()).backward()
    xg, rg = x.to(device).requires_grad_(True), raw.to(device).requires_grad_(True)
    y, lad = ops.piecewise_spline_autograd(xg, rg, cols.to(device), **kw)
    assert type(y.grad_fn).__name__ == "_PiecewiseSplineFunctionBackward"
    ((y * gy.to(device)).sum() + (lad * gl.to(device)).sum()).backward()
    assert maxdiff(y.detach()[:, cols], y_ref.detach()) <= 3e-5 * max(1.0, float(y_ref.detach().abs().max()))
    ident = [c for c in range(d) if c not in cols.tolist()]
    assert torch.equal(xg.grad[:, ident].cpu(), gy[:, ident])                      # identity columns pass through
    gx_ref = x64.grad[:, cols] 
    assert maxdiff(xg.grad[:, cols], gx_ref) <= 1e-3 * max(1.0, float(gx_ref.abs().max()))
    assert maxdiff(rg.grad, r64.grad) <= 1e-3 * max(1e-3, float(r64.grad.abs().max()))
