"""Random coupling / autoregressive flows through the fast-path dispatch (hidden-layer kernel with / without context,
zero-padded widths, activations, chained fused launches, leftover rows, column-at-a-time inverses) against the oracle.
Not part of the test suite; run on the GPU box:  python tools/probe/fuzz_flows.py [seed] [cases]"""
import os
import sys

import torch
from torch.nn import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from flowconductor_amd import distributions, flows, ops, transforms as T, utils  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(seed)
dev = "cuda"
ACTS = [F.relu, torch.tanh, F.silu, F.elu, F.leaky_relu, torch.nn.ReLU(), torch.nn.ELU(0.7)]


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def md(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max()) if a.numel() else 0.0


worst = 0.0
for c in range(cases):
    d = ri(2, 128)
    n = ri(1, 1500)
    hidden = ri(3, 64)
    blocks = ri(0, 4)
    ctx_f = ri(1, 32) if ri(0, 1) else None
    if ctx_f is not None:
        blocks = min(blocks, 3)
    act = ACTS[ri(0, len(ACTS) - 1)]
    torch.manual_seed(seed * 100000 + c)

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=hidden, context_features=ctx_f, num_blocks=blocks, activation=act)

    layers = []
    for i in range(ri(1, 3)):
        mask = (torch.rand(d, generator=g) < 0.5).int()
        if mask.sum() == 0 or mask.sum() == d:
            mask[0] = 1 - mask[0]
        which = ri(0, 3)
        if which == 0:
            layers.append(T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=8, tails="linear",
                                                                        tail_bound=float(ri(2, 4))))
        elif which == 1:
            layers.append(T.AffineCouplingTransform(mask, net))
        elif which == 2 and d <= 24:
            layers.append(T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                d, hidden, context_features=ctx_f, num_blocks=min(blocks, 3), num_bins=8, tails="linear",
                tail_bound=3.0, activation=act if not isinstance(act, torch.nn.Module) else F.relu))
        else:
            layers.append(T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=ri(2, 12), tails="linear",
                                                                        tail_bound=3.0))
        layers.append(T.RandomPermutation(d))
    flow = flows.Flow(T.CompositeTransform(layers), distributions.StandardNormal([d])).eval()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(float(torch.rand(1, generator=g)) + 0.8)
    x = torch.randn(n, d, generator=g) * 1.2
    ctx = torch.randn(n, ctx_f, generator=g) if ctx_f else None
    with torch.no_grad():
        ref = O.flow_log_prob(flow, x, ctx)
        z_ref, _ = O.transform_apply(flow._transform, x.clone(), ctx)
        b_ref, _ = O.transform_apply(flow._transform, z_ref, ctx, inverse=True)
        flow = flow.to(dev)
        cd = None if ctx is None else ctx.to(dev)
        got = flow.log_prob(x.to(dev), cd) if cd is not None else flow.log_prob(x.to(dev))
        options._values["ar_incremental"] = "force" if ri(0, 1) else "auto"
        z, _ = flow._transform(x.to(dev), cd)
        back, _ = flow._transform.inverse(z, cd)
    e = md(got, ref)
    bound = 5e-5 * max(1.0, float(ref.abs().max()))
    assert got.shape == (n,) and e <= bound, (c, d, n, hidden, blocks, ctx_f, act, e, bound)
    rt, rt_ref = md(back, x), md(b_ref, x)
    assert rt <= 5e-4 * max(1.0, float(x.abs().max())) + 8 * rt_ref, (c, "round trip", d, n, hidden, blocks, ctx_f, rt, rt_ref)
    worst = max(worst, e / bound)
print("fuzz ok: seed %d, %d flows; worst log_prob error / bound %.2f" % (seed, cases, worst))
