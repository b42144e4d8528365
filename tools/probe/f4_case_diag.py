"""Diagnostic for one case of tests/test_gpu_fused4.py: resident / streamed kernels and the float32 oracle against float64."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_fused4 as T  # noqa: E402
from flowconductor_amd import ops  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

k, n, d, d_t, inverse, tails = 7, 32, 128, 32, True, None
dev = torch.device("cuda:0")
x, h, w, b, cols = T._case(n, d, d_t, k, seed=n + d + d_t + k, h_decades=not inverse, tails=tails)
p = w.shape[0] // d_t
rows64 = (h.double() @ w.double().T + b.double()).view(n, d_t, p)
xs = x[:, cols.long()]
o64, l64 = O.rq_from_rows(xs.double(), rows64.clone(), k, tails, 3.0, inverse, wh_divisor=8.0)
o32, l32 = O.rq_from_rows(xs, rows64.float().clone(), k, tails, 3.0, inverse, wh_divisor=8.0)
packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
kw = dict(num_bins=k, tails=tails, tail_bound=3.0, wh_divisor=8.0, inverse=inverse)
with torch.no_grad():
    y, lad = ops.rq_spline_fused_general(x.to(dev), h.to(dev), *packed, cols.to(dev), **kw)
    ys, lads = ops.rq_spline_fused_general(x.to(dev), h.to(dev), *packed, cols.to(dev), streamed_weights=True, **kw)
ref = l64.sum(dim=1)
for name, v in (("resident", lad.cpu().double()), ("streamed", lads.cpu().double()), ("oracle f32", l32.double().sum(dim=1))):
    dd = (v - ref).abs()
    print(name, "max |dlogabsdet| vs float64 %.3e at row %d" % (float(dd.max()), int(dd.argmax())))
row = int((lad.cpu().double() - ref).abs().argmax())
ye = (y.cpu()[row, cols.long()].double() - o64[row]).abs()
print("row", row, "worst element", int(ye.argmax()), "dy resident %.3e  dy oracle f32 %.3e" % (float(ye.max()), float((o32[row].double() - o64[row]).abs().max())))
print("element logabsdet float64 (min over the row) %.3f" % float(l64[row].min()))
