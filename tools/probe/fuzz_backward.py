"""Random transforms in TRAINING mode: gradients of L = sum(gy * y) + sum(gl * logabsdet) with respect to the inputs and every
parameter, HIP path (autograd nodes over the backward kernels) against torch.autograd walking the CPU oracle in float64;
float32 autograd on the oracle is the noise floor.  Covers the fused RQ coupling layer (conditioner + spline backward kernels),
affine / additive / linear / quadratic / cubic coupling, sum-of-sigmoids, planar, Sylvester, Householder, LU, MAF.
Not part of the test suite; run on the GPU box:  python tools/probe/fuzz_backward.py [seed] [cases]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from flowconductor_amd import transforms as T, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
g = torch.Generator().manual_seed(seed)
dev = "cuda"

KINDS = ["rq_fused", "rq_fused", "rq_box", "affine", "additive", "linear", "quadratic", "cubic", "sos", "planar", "sylvester",
         "householder", "lu", "maf", "maf_rq"]


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def md(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max()) if a.numel() else 0.0


def grads(t, x, gy, gl, dtype, device):
    t = copy.deepcopy(t).to(dtype).to(device).train()
    xi = x.to(dtype).to(device).clone().requires_grad_(True)
    if device == "cpu":
        with O.differentiable_parameters():
            y, lad = O.transform_apply(t, xi)
    else:
        y, lad = t(xi)
    ((y * gy.to(dtype).to(device)).sum() + (lad * gl.to(dtype).to(device)).sum()).backward()
    return xi.grad, {n: p.grad for n, p in t.named_parameters()}, y.detach(), lad.detach()


worst = {}
for c in range(cases):
    kind = KINDS[ri(0, len(KINDS) - 1)]
    n = ri(1, 1500)
    d = ri(2, 48)
    k = ri(2, 12)
    hidden = [16, 32, 64][ri(0, 2)]
    blocks = ri(1, 2)
    if kind == "rq_fused":          # the shapes the fused training path takes: K = 4..16, hidden <= 64, <= 2 ReLU blocks
        k, d = ri(4, 10), 2 * ri(2, 40)
    mask_kind = ri(0, 2)
    if mask_kind == 0:
        mask = utils.create_alternating_binary_mask(d, even=bool(ri(0, 1)))
    elif mask_kind == 1:
        mask = utils.create_mid_split_binary_mask(d)
    else:
        mask = (torch.rand(d, generator=g) < 0.5).int()
        if mask.sum() == 0 or mask.sum() == d:
            mask[0] = 1 - mask[0]

    def net(i, o):
        return nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=blocks)

    torch.manual_seed(seed * 100000 + c)
    unit = kind in ("rq_box", "linear", "quadratic", "cubic")
    if kind == "rq_fused":
        t = T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=k, tails="linear", tail_bound=float(ri(2, 4)))
    elif kind == "rq_box":
        t = T.PiecewiseRationalQuadraticCouplingTransform(mask, net, num_bins=k)
    elif kind == "linear":
        t = T.PiecewiseLinearCouplingTransform(mask, net, num_bins=k)
    elif kind == "quadratic":
        t = T.PiecewiseQuadraticCouplingTransform(mask, net, num_bins=k)
    elif kind == "cubic":
        t = T.PiecewiseCubicCouplingTransform(mask, net, num_bins=k)
    elif kind == "affine":
        t = T.AffineCouplingTransform(mask, net)
    elif kind == "additive":
        t = T.AdditiveCouplingTransform(mask, net)
    elif kind == "sos":
        d = ri(1, 12)
        t = T.SumOfSigmoids(features=d, n_sigmoids=ri(2, 30))
    elif kind == "planar":
        t = T.PlanarTransform(features=d)
    elif kind == "sylvester":
        t = T.SylvesterTransform(features=d, num_householder=ri(1, min(d, 8)), device="cpu")
    elif kind == "householder":
        # (more than 2 d - 1 reflections make the reference's initialisation index past the features or produce zero vectors)
        t = T.HouseholderSequence(features=d, num_transforms=ri(1, min(8, 2 * d - 1)))
    elif kind == "lu":
        t = T.LULinear(d, identity_init=False)
    elif kind == "maf":
        t = T.MaskedAffineAutoregressiveTransform(features=d, hidden_features=hidden, num_blocks=blocks)
    else:
        t = T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features=d, hidden_features=hidden, num_blocks=blocks,
                                                                      num_bins=k, tails="linear", tail_bound=3.0)
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(float(torch.rand(1, generator=g)) * 1.5 + 0.5)
    x = torch.rand(n, d, generator=g) * 0.98 + 0.01 if unit else torch.randn(n, d, generator=g) * 1.3
    gy, gl = torch.randn(n, d, generator=g), torch.randn(n, generator=g)

    gx64, gp64, y64, l64 = grads(t, x, gy, gl, torch.float64, "cpu")
    gx32, gp32, _, _ = grads(t, x, gy, gl, torch.float32, "cpu")
    gx, gp, y, lad = grads(t, x, gy, gl, torch.float32, dev)
    tag = (kind, "n", n, "d", d, "k", k, "hidden", hidden, "blocks", blocks, "mask", mask_kind)
    assert md(y, y64) <= 2e-5 * max(1.0, float(y64.abs().max())) + 8 * 1e-6, tag + ("forward",)
    sx = max(1e-6, float(gx64.abs().max()))
    ex, fx = md(gx, gx64), md(gx32, gx64)
    assert ex <= 2e-4 * sx + 8 * fx, tag + ("grad x", ex, fx, sx)
    ratio = ex / (2e-4 * sx + 8 * fx)
    assert set(gp) == set(gp64)
    for name, ref in gp64.items():
        if ref is None:
            assert gp[name] is None or float(gp[name].abs().max()) == 0.0, tag + (name, "unexpected gradient")
            continue
        assert gp[name] is not None, tag + (name, "missing gradient")
        sp = max(1e-6, float(ref.abs().max()))
        ep, fp = md(gp[name], ref), md(gp32[name], ref)
        # a parameter gradient is a sum of n per-sample terms of O(|gy|) = O(1), each good to ~1e-7 relative in float32: where
        # they cancel (sum-of-sigmoids shifts: |sum| ~ 0.05 from 1 333 terms) the absolute error of the sum does not shrink with it
        bound = 3e-4 * sp + 8 * fp + 1e-7 * n
        assert ep <= bound, tag + (name, ep, fp, sp)
        ratio = max(ratio, ep / bound)
    worst[kind] = max(worst.get(kind, 0.0), ratio)
print("fuzz ok: seed %d, %d cases; worst gradient error / bound per kind %s" % (seed, cases, {a: "%.2f" % b for a, b in worst.items()}))
