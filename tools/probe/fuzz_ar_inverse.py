"""Random MADE-conditioned autoregressive layers through fc_made_inverse (the D passes in one kernel, each pass on the
hidden-unit prefix it reads) against (a) the oracle's D full passes in float64 and (b) this package's host loop; the forward
of the result must give the inputs back.  Affine and RQ forms, D 2..64, hidden 1..64, 0..3 blocks, K 1..16, both tail modes,
any batch.  Not part of the test suite; run on the GPU box:  python tools/probe/fuzz_ar_inverse.py [seed] [cases]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from flowconductor_amd import ops, options, transforms as T  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator().manual_seed(seed)
dev = "cuda"


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def md(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max()) if a.numel() else 0.0


worst = 0.0
ran = failed = left_out = 0
for c in range(cases):
    d = ri(2, 64)
    n = ri(1, 3000) if ri(0, 3) else ri(60000, 140000)        # (large batches: two blocks per wave, odd group counts)
    hidden = ri(1, 64)
    blocks = ri(0, 3)
    torch.manual_seed(seed * 100000 + c)
    kind = ri(0, 2)
    if kind == 0:
        t = T.MaskedAffineAutoregressiveTransform(d, hidden, num_blocks=blocks)
        box = False
    else:
        k = ri(1, 16)
        box = kind == 2
        t = T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            d, hidden, num_blocks=blocks, num_bins=k, tails=None if box else "linear", tail_bound=float(ri(2, 4)))
    t = t.eval()
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.0 + 1.5 * float(torch.rand(1, generator=g)))
    x = torch.rand(n, d, generator=g) if box else 2.0 * torch.randn(n, d, generator=g)
    t_cpu = t
    t = copy.deepcopy(t_cpu).to(dev)
    xd = x.to(dev)
    with torch.no_grad():
        if not t._device_loop_ok(xd, None):
            print("case %d: d=%d hidden=%d blocks=%d kind=%d -> host loop (not counted)" % (c, d, hidden, blocks, kind))
            continue
        with ops.KernelTimer("fc_made_inverse") as timer:
            y, lad = t.inverse(xd)
        assert len(timer.pairs) == 1
        with options.override(ar_device_loop=False):
            y_host, lad_host = t.inverse(xd)
        z, lad_fwd = t.forward(y)
    # the oracle on the rows where the two device paths disagree most (an ill-conditioned row amplifies the rounding of
    # every earlier column: the float32-against-float64 oracle difference on the SAME rows sets the tolerance) + 512 others
    if n > 1024:
        score = (y - y_host).abs().amax(dim=1) + (lad - lad_host).abs() + (z - xd).abs().amax(dim=1)
        rows = torch.cat((torch.topk(score, 256).indices.cpu(), torch.randperm(n, generator=g)[:512])).unique()
    else:
        rows = torch.arange(n)
    x_ref = x[rows]
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(t_cpu, x_ref.clone(), inverse=True)
        ref_y64, ref_lad64 = O.transform_apply(copy.deepcopy(t_cpu).double(), x_ref.double(), inverse=True)
    rows = rows.to(dev)
    y, lad, y_host, lad_host, z, lad_fwd, xd = y[rows], lad[rows], y_host[rows], lad_host[rows], z[rows], lad_fwd[rows], xd[rows]
    tol_y = 1e-4 * max(1.0, float(ref_y.abs().max())) + 4 * md(ref_y, ref_y64)
    tol_l = 1e-3 * max(1.0, float(ref_lad.abs().max()) / 10) + 4 * md(ref_lad, ref_lad64)
    # The device loop is held to the oracle on every row.  The host loop and the forward pass hand the conditioner whole
    # rows of y: the hidden-layer kernel's row scaling (fc_split.h) keeps an entry's absolute error at 2^-38 of the row
    # maximum, so rows where an (ill-conditioned) inverse has produced |y| > 1e3 beside O(1) columns are left out of THOSE
    # two comparisons (the reference's masked f32 GEMM multiplies the large column by an exact zero instead).
    tame = (ref_y64.abs().amax(dim=1) <= 1e3).to(dev)
    left_out += int((~tame).sum())
    tc = tame.cpu()
    if bool(tc.any()):
        ty = 1e-4 * max(1.0, float(ref_y[tc].abs().max())) + 4 * md(ref_y[tc], ref_y64[tc])
        tl = 1e-3 * max(1.0, float(ref_lad[tc].abs().max()) / 10) + 4 * md(ref_lad[tc], ref_lad64[tc])
    else:
        ty, tl = tol_y, tol_l
    # (two float32 paths against each other, and a round trip through the forward map whose slope reaches 1 / 1e-3: four
    #  times the allowance of one path against float64)
    e = (md(y, ref_y64) / tol_y, md(lad, ref_lad64) / tol_l, md(y[tame], y_host[tame]) / (4 * ty),
         md(lad[tame], lad_host[tame]) / (4 * tl), md(z[tame], xd[tame]) / (4 * ty),
         md((lad + lad_fwd)[tame], torch.zeros_like(lad[tame])) / (4 * tl))
    worst = max(worst, max(e))
    ran += 1
    flag = "" if max(e) <= 1.0 else "   <-- FAIL"
    print("case %d: d=%d n=%d hidden=%d blocks=%d kind=%d  err/tol %s%s" % (c, d, n, hidden, blocks, kind,
                                                                           " ".join("%.2f" % v for v in e), flag))
    if flag:
        failed += 1
        r = int(((y - y_host).abs().amax(dim=1) * tame).argmax())
        print("   worst row %d of the checked ones: y %s\n   host %s\n   f64  %s\n   f32  %s" % (
            r, y[r, :6].tolist(), y_host[r, :6].tolist(), ref_y64[r, :6].tolist(), ref_y[r, :6].tolist()))
print("fuzz_ar_inverse seed %d: %d cases on the device loop, %d failed, worst err/tol %.2f (%d rows with |y| > 1e3 left out of "
      "the host-loop / forward comparisons)" % (seed, ran, failed, worst, left_out))
if failed:
    sys.exit(1)
print("fuzz ok: seed %d, %d device-loop layers" % (seed, ran))
