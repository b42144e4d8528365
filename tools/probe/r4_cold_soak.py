"""First launches of a fresh process through the kernels added or re-cut in round 4 -- fc_made_inverse (affine and RQ forms,
prefix passes), fc_sylvester_mm, fc_planar (sub-wave rows), fc_elementwise (16-byte rows) -- each against float64 on the CPU.
The round-2 backward fault only showed on cold launches; run this in many fresh processes:  bash tools/probe/r4_cold_soak.sh"""
import copy
import os
import sys

import torch

sys.path[:0] = [os.getcwd()]
from flowconductor_amd import ops, transforms as T  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

dev = "cuda"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(seed)
bad = []


def md(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


# 1. the D passes in one kernel, RQ form (the very first launch of the process), then the affine form
for name, make, d, n in (("rq_ar", lambda: T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                              8, 64, num_blocks=2, num_bins=8, tails="linear", tail_bound=3.0), 8, 70000),
                         ("maf", lambda: T.MaskedAffineAutoregressiveTransform(40, 48, num_blocks=2), 40, 4099)):
    t = make().eval()
    x = torch.randn(n, d)
    with torch.no_grad():
        sub = x[:512]
        ref_y, ref_lad = O.transform_apply(copy.deepcopy(t).double(), sub.double(), inverse=True)
        with ops.KernelTimer("fc_made_inverse") as tm:
            y, lad = t.to(dev).inverse(x.to(dev))
        z, lad_f = t(y)
    if len(tm.pairs) != 1:
        bad.append(name + ": device loop not taken")
    e = (md(y[:512], ref_y), md(lad[:512], ref_lad), md(z, x), md(lad + lad_f, torch.zeros_like(lad)))
    if not (e[0] <= 2e-4 and e[1] <= 2e-3 and e[2] <= 5e-4 and e[3] <= 5e-3):
        bad.append("%s: %s" % (name, " ".join("%.2e" % v for v in e)))

# 2. shared-parameter Sylvester on the matrix cores
t = T.SylvesterTransform(features=128, num_householder=32, device=None).eval()
x = torch.randn(4096 + 16, 128)
with torch.no_grad():
    t.Q_orth.q_vectors.copy_(torch.randn(32, 128))
    ref_y, ref_lad = O.transform_apply(copy.deepcopy(t).double(), x.double())
    with ops.KernelTimer("fc_sylvester_mm") as tm:
        y, lad = t.to(dev)(x.to(dev))
if len(tm.pairs) != 1 or md(y, ref_y) > 2e-5 * max(1.0, float(ref_y.abs().max())) or md(lad, ref_lad) > 2e-4:
    bad.append("sylvester_mm: %d launches, %.2e %.2e" % (len(tm.pairs), md(y, ref_y), md(lad, ref_lad)))

# 3. planar (sub-wave rows) and Tanh (16-byte rows)
x = torch.randn(5001, 64)
w, u, b = torch.randn(1, 64) * 0.3, torch.randn(1, 64) * 0.3, torch.randn(1) * 0.2
y, lad = ops.planar(x.to(dev), w.to(dev), u.to(dev), b.to(dev))
a = (x.double() * w.double()).sum(-1) + b.double()
th = torch.tanh(a)
ref_y = x.double() + u.double() * th.unsqueeze(-1)
ref_lad = torch.log(1e-7 + (1 + (1 - th ** 2) * (u.double() * w.double()).sum(-1)).abs())
a32 = (x * w).sum(-1) + b                 # the reference's float32 sequence: the noise floor where |1 + s| is small
th32 = torch.tanh(a32)
floor = md(torch.log(1e-7 + (1 + (1 - th32 ** 2) * (u * w).sum(-1)).abs()), ref_lad)
if md(y, ref_y) > 4e-6 or md(lad, ref_lad) > 2e-5 + 4 * floor:
    bad.append("planar: %.2e %.2e (float32 reference %.2e)" % (md(y, ref_y), md(lad, ref_lad), floor))
with torch.no_grad():
    y, lad = T.Tanh().to(dev)(x.to(dev))
x64 = x.double()
ref_lad = (2 * (torch.log(torch.tensor(2.0, dtype=torch.float64)) - x64.abs() - torch.log1p(torch.exp(-2 * x64.abs())))).sum(-1)
if md(y, torch.tanh(x64)) > 4e-7 or md(lad, ref_lad) > 1e-4:
    bad.append("tanh: %.2e %.2e" % (md(y, torch.tanh(x64)), md(lad, ref_lad)))

print("seed %d: %s" % (seed, "clean" if not bad else "BAD " + "; ".join(bad)))
