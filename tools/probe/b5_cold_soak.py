"""First launch of a fresh process through the ONE-LAUNCH fused backward (fc_rq_fused_backward512.h): all four gradients of
fc_rq_fused_linear_backward at N = 4096 (two sweeps, 16 tiles per workgroup... one tile per workgroup at this size) against
float64 autograd on the oracle.  The round-2 fault of the two-launch kernel showed on 3-20 % of such cold launches
(DESIGN.md section 4d); run it in many fresh processes:  bash tools/probe/b5_cold_soak.sh"""
import os
import sys

import torch

sys.path[:0] = [os.getcwd()]
from flowconductor_amd import ops  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

dev = "cuda"
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 8
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 4096
tails, d, d_t, hidden = "linear", 64, 32, 64
torch.manual_seed(0)
p = 3 * k - 1
x = torch.randn(n, d) * 1.5
h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
b = torch.randn(d_t * p) * 0.3
cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
gy, gl = torch.randn(n, d), torch.randn(n)
kw = dict(wh_divisor=float(hidden) ** 0.5)
# the FIRST kernel launches of this process
packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
packed_t = ops.pack_final_layer_transposed(w.to(dev), k, tails)
got = ops.rq_fused_linear_backward(x.to(dev), h.to(dev), gy.to(dev), gl.to(dev), packed, packed_t, cols.to(dev), num_bins=k,
                                   tails=tails, tail_bound=3.0, **kw)
got = [g.cpu().double() for g in got]
x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
rows = (h64 @ w64.T + b64).view(n, d_t, p)
out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
y64 = x64.clone().index_copy(1, cols.long(), out)
loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
refs = torch.autograd.grad(loss, (x64, h64, w64, b64))
bad = []
for name, g, r in zip(("gx", "gh", "gW", "gb"), got, refs):
    rel = float((g - r).abs().max() / r.abs().max())
    if not rel <= 1e-4:
        rows_bad = torch.nonzero((g - r).abs().reshape(g.shape[0], -1).amax(dim=1) > 1e-3 * float(r.abs().max())).flatten().tolist()
        bad.append((name, rel, rows_bad[:8]))
print("first launch K=%d N=%d: %s" % (k, n, "clean" if not bad else "BAD %s" % bad))
