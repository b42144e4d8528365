"""Flow-level behaviour and the BASELINE.json configurations at (or near) their full sizes,
checked through size-independent properties plus oracle spot-checks on row subsets."""
import pytest
import torch

from _util import Lib, maxdiff
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
T, nets, utils, flows, distributions = Lib.transforms, Lib.nets, Lib.utils, Lib.flows, Lib.distributions


def _rq_stack(d, layers, hidden, bins=8, context=None):
    return T.CompositeTransform([
        T.PiecewiseRationalQuadraticCouplingTransform(
            utils.create_alternating_binary_mask(d, even=(l % 2 == 0)),
            lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=2, context_features=context),
            num_bins=bins, tails="linear", tail_bound=3.0)
        for l in range(layers)])


def test_flow_shapes_and_sample_consistency(device):
    """reference tests/flows/base_test.py:13-127."""
    torch.manual_seed(0)
    d, ctx = 6, 3
    flow = flows.Flow(_rq_stack(d, 3, 16, context=ctx), distributions.StandardNormal([d]),
                      embedding_net=torch.nn.Linear(4, ctx)).to(device).eval()
    x = torch.randn(10, d, device=device)
    c = torch.randn(10, 4, device=device)
    with torch.no_grad():
        lp = flow.log_prob(x, context=c)
        assert lp.shape == (10,)
        s = flow.sample(5, context=c)
        assert s.shape == (10, 5, d)
        samples, lps = flow.sample_and_log_prob(4, context=c)
        assert samples.shape == (10, 4, d) and lps.shape == (10, 4)
        lp_again = flow.log_prob(samples.reshape(40, d), context=c.repeat_interleave(4, dim=0)).reshape(10, 4)
        noise = flow.transform_to_noise(x, context=c)
        assert noise.shape == (10, d)
    assert maxdiff(lps, lp_again) <= 2e-3
    with pytest.raises(ValueError):
        flow.log_prob(x, context=c[:5])
    # without context
    flow2 = flows.Flow(_rq_stack(d, 2, 16), distributions.StandardNormal([d])).to(device).eval()
    with torch.no_grad():
        assert flow2.log_prob(x).shape == (10,)
        assert flow2.sample(7).shape == (7, d)
        assert flow2.sample(7, batch_size=3).shape == (7, d)
        s2, l2 = flow2.sample_and_log_prob(9)
        assert maxdiff(l2, flow2.log_prob(s2)) <= 2e-3


def test_canned_flows_match_oracle(device):
    torch.manual_seed(1)
    maf = flows.MaskedAutoregressiveFlow(features=4, hidden_features=16, num_layers=2, num_blocks_per_layer=2).eval()
    nvp = flows.SimpleRealNVP(features=6, hidden_features=16, num_layers=3, num_blocks_per_layer=1).eval()
    for flow, d in ((maf, 4), (nvp, 6)):
        x = torch.randn(33, d)
        with torch.no_grad():
            ref = O.flow_log_prob(flow, x.clone())
            got = flow.to(device).log_prob(x.to(device))
        assert maxdiff(got, ref) <= 1e-4
        flow.cpu()


def test_coupling_4d_inputs_match_oracle(device):
    """coupling layers on [N, C, H, W] inputs split on C (reference coupling_test.py 4-D shape [2, 4, 4])."""
    torch.manual_seed(2)

    class ConvNet(torch.nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.hidden_channels = 8
            self.a = torch.nn.Conv2d(i, 8, 3, padding=1)
            self.b = torch.nn.Conv2d(8, o, 3, padding=1)

        def forward(self, x, context=None):
            return self.b(torch.relu(self.a(x)))

    mask = utils.create_alternating_binary_mask(4)
    cases = [
        T.AffineCouplingTransform(mask, ConvNet),
        T.AdditiveCouplingTransform(mask, ConvNet),
        T.PiecewiseRationalQuadraticCouplingTransform(mask, ConvNet, num_bins=5, tails="linear", tail_bound=3.0),
        T.PiecewiseQuadraticCouplingTransform(mask, ConvNet, num_bins=5, tails="linear", tail_bound=3.0),
    ]
    x = torch.randn(3, 4, 4, 4)
    for t in cases:
        t.eval()
        with torch.no_grad():
            y_ref, lad_ref = O.transform_apply(t, x.clone())
            y, lad = t.to(device)(x.to(device))
            xb, ladb = t.inverse(y)
        assert y.shape == x.shape and lad.shape == (3,)
        assert maxdiff(y, y_ref) <= 2e-5 * max(1.0, float(y_ref.abs().max())), type(t).__name__
        assert maxdiff(lad, lad_ref) <= 2e-4, type(t).__name__
        assert maxdiff(xb, x) <= 1e-3 and maxdiff(lad + ladb, torch.zeros(3)) <= 2e-3
        # identity half untouched (reference coupling_test.py:50)
        assert torch.equal(y[:, t.identity_features].cpu(), x[:, t.identity_features.cpu()])


def test_actnorm_data_dependent_init_and_state_dict(device):
    """reference normalization_test.py:76-143."""
    t = T.ActNorm(5).to(device)
    t.train()
    x = torch.randn(4096, 5, device=device) * 3 + 1.5
    with torch.no_grad():
        y, lad = t(x)
    assert bool(t.initialized)
    assert maxdiff(y.mean(0), torch.zeros(5)) <= 1e-4 and maxdiff(y.std(0), torch.ones(5)) <= 1e-4
    assert maxdiff(lad, torch.full((4096,), float(t.log_scale.detach().sum()))) <= 1e-6
    t2 = T.ActNorm(5)
    t2.load_state_dict(t.state_dict())
    assert bool(t2.initialized)
    with torch.no_grad():
        xb, ladb = t.inverse(y)
    assert maxdiff(xb, x) <= 1e-4 and maxdiff(lad + ladb, torch.zeros(4096)) == 0.0


def test_batchnorm_train_mode_statistics(device):
    """reference normalization_test.py:13-59: training mode normalises with batch mean / unbiased var."""
    t = T.BatchNorm(4).to(device)
    t.train()
    x = torch.randn(512, 4, device=device) * 2 + 0.7
    with torch.no_grad():
        y, lad = t(x)
    mean, var = x.mean(0), x.var(0)
    w = t.weight.detach()
    expect = w * ((x - mean) / torch.sqrt(var + t.eps)) + t.bias.detach()
    assert maxdiff(y, expect) <= 1e-5
    assert maxdiff(t.running_mean, 0.1 * mean) <= 1e-6
    assert maxdiff(lad, torch.full((512,), float((torch.log(w) - 0.5 * torch.log(var + t.eps)).sum()))) <= 1e-5


# ---- BASELINE.json configurations -----------------------------------------------------------------

def test_config1_readme_maf_flow(device):
    """configs[0]: MAF(2, hidden 4) + RandomPermutation, batch 4096, vs the CPU oracle."""
    torch.manual_seed(0)
    flow = flows.Flow(T.CompositeTransform([T.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4),
                                            T.RandomPermutation(features=2)]),
                      distributions.StandardNormal([2])).eval()
    x = torch.randn(4096, 2)
    with torch.no_grad():
        ref = O.flow_log_prob(flow, x.clone())
        got = flow.to(device).log_prob(x.to(device))
        s, lp = flow.sample_and_log_prob(1000)
        assert maxdiff(lp, flow.log_prob(s)) <= 1e-4
    assert maxdiff(got, ref) <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_config2_affine_coupling_bruteforce_jacobian(device):
    """configs[1]: 8-layer affine coupling, D=32, N=2^18; logabsdet vs slogdet of the brute-force
    autograd Jacobian (reference transform_test.py:29-37) on a 16-row slice, 1e-5."""
    torch.manual_seed(0)
    d, n = 32, 1 << 18
    stack = T.CompositeTransform([
        T.AffineCouplingTransform(utils.create_alternating_binary_mask(d, even=(l % 2 == 0)),
                                  lambda i, o: nets.ResidualNet(i, o, hidden_features=64, num_blocks=2))
        for l in range(8)]).eval()
    x = torch.randn(n, d)
    xs = x[:16].clone().requires_grad_(True)
    ys, _ = O.transform_apply(stack, xs)  # differentiable CPU restatement
    jac = utils.batch_jacobian(ys, xs)
    brute = torch.linalg.slogdet(jac.detach().double())[1].float()
    with torch.no_grad():
        y, lad = stack.to(device)(x.to(device))
        xb, ladb = stack.inverse(y)
    assert maxdiff(lad[:16], brute) <= 1e-5 * max(1.0, float(brute.abs().max()))
    assert maxdiff(y[:16], ys.detach()) <= 1e-5 * max(1.0, float(ys.detach().abs().max()))
    assert maxdiff(xb, x) <= 1e-4 and float((lad + ladb).abs().max()) <= 1e-4
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()


def test_config3_rq_nsf_full_depth_properties(device):
    """configs[2]: 32-layer RQ-NSF coupling, D=64, K=8 at N=2^18 (2^20 is bench.py's job):
    forward/inverse round trip, logabsdet antisymmetry, oracle spot-check on 256 rows."""
    torch.manual_seed(0)
    stack = _rq_stack(64, 32, 64).eval()
    n = 1 << 18
    x = torch.randn(n, 64)
    with torch.no_grad():
        z_ref, lad_ref = O.transform_apply(stack, x[:256].clone())
        stack = stack.to(device)
        xd = x.to(device)
        z, lad = stack(xd)
        xb, ladb = stack.inverse(z)
    assert torch.isfinite(z).all() and torch.isfinite(lad).all()
    assert maxdiff(z[:256], z_ref) <= 1e-4 and maxdiff(lad[:256], lad_ref) <= 1e-3
    assert float((xb - xd).abs().max()) <= 5e-3
    assert float((lad + ladb).abs().max()) <= 5e-2
    # deterministic: same inputs, same bits
    with torch.no_grad():
        z2, lad2 = stack(xd)
    assert torch.equal(z, z2) and torch.equal(lad, lad2)


def test_config5_sylvester_d128_m32_large_batch(device):
    """configs[4] (non-conditional class, SURVEY.md headline facts): N=2^18, D=128, M=32."""
    torch.manual_seed(0)
    t = T.SylvesterTransform(features=128, num_householder=32, device="cpu")
    with torch.no_grad():
        t.Q_orth.q_vectors.copy_(torch.randn(32, 128))
    n = 1 << 18
    x = torch.randn(n, 128)
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(t, x[:128].clone())
        y, lad = t.to(device)(x.to(device))
    assert torch.isfinite(y).all() and torch.isfinite(lad).all()
    assert maxdiff(y[:128], y_ref) <= 3e-5 * max(1.0, float(y_ref.abs().max()))
    assert maxdiff(lad[:128], lad_ref) <= 1e-4 * max(1.0, float(lad_ref.abs().max()))
    # the tail rows go through the same code path as the head rows
    with torch.no_grad():
        y_tail, lad_tail = t(x[-128:].to(device))
    assert torch.equal(y_tail, y[-128:]) and torch.equal(lad_tail, lad[-128:])


def test_hip_graph_replay_of_log_prob(device):
    """A captured HIP graph of Flow.log_prob (utils/graphs.GraphedCall) replays bit-identically to eager calls on
    new inputs of the same shape, and still raises the reference's domain exception after a replay."""
    from flowconductor_amd import distributions, flows, transforms
    from flowconductor_amd.utils.graphs import GraphedCall

    torch.manual_seed(0)
    layers = []
    for _ in range(4):
        layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
        layers.append(transforms.RandomPermutation(features=2))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).to(device).eval()
    x0 = torch.randn(4096, 2, device=device)
    graphed = GraphedCall(flow.log_prob, x0)
    for seed in (1, 2):
        x = torch.randn(4096, 2, device=device, generator=torch.Generator(device=device).manual_seed(seed))
        with torch.no_grad():
            eager = flow.log_prob(x)
        assert torch.equal(graphed(x), eager)
    with pytest.raises(ValueError):
        graphed(torch.randn(100, 2, device=device))

    # error word after a replay: a box-domain spline fed values outside its interval
    t = transforms.PiecewiseRationalQuadraticCDF(shape=[3], num_bins=4, tails=None).to(device).eval()
    inside = torch.rand(64, 3, device=device)
    g2 = GraphedCall(t.forward, inside)
    y, lad = g2(inside)
    assert torch.isfinite(y).all()
    with pytest.raises(transforms.InputOutsideDomain):
        g2(inside + 5.0)


@pytest.mark.parametrize("n", [0, 1, 15, 31, 33])
def test_empty_and_tiny_batches_through_every_fast_path(n, device):
    """N = 0 and batches smaller than one kernel tile (16 rows of the hidden kernel, 32 of the fused one) through the
    coupling (RQ fused / affine / conditional), autoregressive (MAF, RQ-AR) and base-distribution paths: shapes kept,
    values equal to the oracle's."""
    torch.manual_seed(5)
    d, ctx_f = 12, 3

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=64, num_blocks=2)

    def cnet(i, o):
        return nets.ResidualNet(i, o, hidden_features=64, context_features=ctx_f, num_blocks=2)

    mask = utils.create_alternating_binary_mask(d)
    stacks = {
        "rq": T.CompositeTransform([T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=8, tails="linear",
                                                                                  tail_bound=3.0),
                                    T.AffineCouplingTransform(mask, net), T.ReversePermutation(d)]),
        "conditional": T.CompositeTransform([T.PiecewiseRationalQuadraticCouplingTransform(
            mask, cnet, num_bins=8, tails="linear", tail_bound=3.0)]),
        "ar": T.CompositeTransform([T.MaskedAffineAutoregressiveTransform(d, 64),
                                    T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                                        d, 64, num_bins=8, tails="linear", tail_bound=3.0)]),
    }
    for name, stack in stacks.items():
        stack.eval()
        with torch.no_grad():
            for p in stack.parameters():
                p.mul_(1.5)
        flow = Lib.flows.Flow(stack, Lib.distributions.StandardNormal([d])).eval()
        x = torch.randn(n, d)
        c = torch.randn(n, ctx_f) if name == "conditional" else None
        with torch.no_grad():
            ref = O.flow_log_prob(flow, x, c) if n else torch.zeros(0)
            flow = flow.to(device)
            xd = x.to(device)
            cd = None if c is None else c.to(device)
            lp = flow.log_prob(xd, cd) if c is not None else flow.log_prob(xd)
            z, lad = flow._transform(xd, cd)
            back, lad_inv = flow._transform.inverse(z, cd)
        assert lp.shape == (n,) and z.shape == (n, d) and lad.shape == (n,) and back.shape == (n, d), name
        if n:
            assert maxdiff(lp, ref) <= 3e-5 * max(1.0, float(ref.abs().max())), name
            assert maxdiff(back, x) <= 2e-4 * max(1.0, float(x.abs().max())), name


def test_strided_inputs_and_contexts(device):
    """Non-contiguous views (every other column of a wider tensor, a transposed buffer, an expanded context) give
    the results of their contiguous copies on every fast path."""
    torch.manual_seed(9)
    d, ctx_f, n = 16, 4, 1000

    def cnet(i, o):
        return nets.ResidualNet(i, o, hidden_features=64, context_features=ctx_f, num_blocks=2)

    mask = utils.create_alternating_binary_mask(d)
    stack = T.CompositeTransform([
        T.PiecewiseRationalQuadraticCouplingTransform(mask, cnet, num_bins=8, tails="linear", tail_bound=3.0),
        T.AffineCouplingTransform(mask, cnet), T.RandomPermutation(d),
        T.MaskedAffineAutoregressiveTransform(d, 64, context_features=ctx_f)]).to(device).eval()
    wide = torch.randn(n, 2 * d, device=device)
    x_view = wide[:, ::2]
    x_t = torch.randn(d, n, device=device).t()
    c_row = torch.randn(1, ctx_f, device=device)
    with torch.no_grad():
        for x in (x_view, x_t):
            assert not x.is_contiguous()
            for c in (c_row.expand(n, ctx_f), torch.randn(n, 2 * ctx_f, device=device)[:, ::2]):
                assert not c.is_contiguous()
                y, lad = stack(x, c)
                y_ref, lad_ref = stack(x.contiguous(), c.contiguous())
                assert torch.equal(y, y_ref) and torch.equal(lad, lad_ref)
                back, _ = stack.inverse(y[:, :], c)
                assert maxdiff(back, x) <= 2e-4 * max(1.0, float(x.abs().max()))


def test_non_finite_rows_do_not_leak_into_other_rows(device):
    """NaN / +-inf in some rows of the input: those rows come out non-finite (the reference would propagate or raise),
    every other row is bit-identical to a run without them -- through the fused coupling path, the stand-alone spline
    kernels and an autoregressive layer."""
    torch.manual_seed(13)
    d, n = 16, 512

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=64, num_blocks=2)

    mask = utils.create_alternating_binary_mask(d)
    layers = {
        "rq_fused": T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=8, tails="linear", tail_bound=3.0),
        "rq_k10": T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=10, tails="linear", tail_bound=3.0),
        "quadratic": T.PiecewiseQuadraticCouplingTransform(mask, net, num_bins=8, tails="linear", tail_bound=3.0),
        "affine": T.AffineCouplingTransform(mask, net),
        "maf": T.MaskedAffineAutoregressiveTransform(d, 64),
    }
    x = torch.randn(n, d, device=device)
    bad = x.clone()
    bad[3, 0] = float("nan")        # identity column -> poisons the conditioner of that row
    bad[10, 1] = float("nan")       # transformed column
    bad[20, 1] = float("inf")
    bad[30, 0] = float("-inf")
    rows = torch.ones(n, dtype=torch.bool, device=device)
    rows[[3, 10, 20, 30]] = False
    for name, t in layers.items():
        t = t.to(device).eval()
        with torch.no_grad():
            y, lad = t(x)
            try:
                yb, ladb = t(bad)
            except T.InputOutsideDomain:
                continue            # also a legal answer (the reference's domain check), as long as nothing hangs
        assert torch.equal(yb[rows], y[rows]) and torch.equal(ladb[rows], lad[rows]), name
        assert not torch.isfinite(yb[3]).all() or not torch.isfinite(ladb[3]), name


def test_diagonal_normal_base_distributions(device):
    """DiagonalNormal / ConditionalDiagonalNormal (distributions/normal.py:53-175) on the HIP kernels: log_prob, sampling
    and parameter gradients against the reference's formulas in float64."""
    import math

    from flowconductor_amd import distributions as D

    torch.manual_seed(3)
    d, n = 7, 500
    x = torch.randn(n, d)
    dist = D.DiagonalNormal([d])
    with torch.no_grad():
        dist.mean_.normal_(0, 1)
        dist.log_std_.normal_(0, 0.5)
    dist = dist.to(device)
    xd = x.to(device)
    lp = dist.log_prob(xd)
    mean, log_std = dist.mean_.detach().cpu().double().requires_grad_(True), dist.log_std_.detach().cpu().double().requires_grad_(True)
    ref = (-0.5 * (((x.double() - mean) * torch.exp(-log_std)) ** 2).sum(1) - log_std.sum() - 0.5 * d * math.log(2 * math.pi))
    assert maxdiff(lp.detach(), ref.detach()) <= 1e-5 * max(1.0, float(ref.detach().abs().max()))
    (-lp.mean()).backward()
    (-ref.mean()).backward()
    assert maxdiff(dist.mean_.grad, mean.grad) <= 1e-5 and maxdiff(dist.log_std_.grad, log_std.grad) <= 1e-4
    assert set(dist.state_dict()) == {"mean_", "log_std_"}

    enc = torch.nn.Linear(3, 2 * d).to(device)
    cdist = D.ConditionalDiagonalNormal([d], context_encoder=enc).to(device)
    c = torch.randn(n, 3, device=device)
    with torch.no_grad():
        lp = cdist.log_prob(xd, c)
        params = enc(c).cpu().double()
        means, log_stds = params[:, :d], params[:, d:]
        ref = (-0.5 * (((x.double() - means) * torch.exp(-log_stds)) ** 2).sum(1) - log_stds.sum(1) - 0.5 * d * math.log(2 * math.pi))
        assert maxdiff(lp, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
        s = cdist.sample(50, context=c[:4])
        assert s.shape == (4, 50, d)
        z = (s.cpu().double() - means[:4, None]) * torch.exp(-log_stds[:4, None])
        assert abs(float(z.mean())) < 0.15 and abs(float(z.std()) - 1.0) < 0.15
    with pytest.raises(ValueError):
        cdist.log_prob(xd, None)
    # trains: gradients reach the context encoder through the per-sample affine kernel
    (-cdist.log_prob(xd, c).mean()).backward()
    assert enc.weight.grad is not None and torch.isfinite(enc.weight.grad).all()
