"""Localises the round-2 gW fault on the dump build of the old kernel (tools/probe/build/libfc_oldbwd_dump.so: role 1 writes, per
lane, gp[0][0], gp[1][0] (the width-0 gradient of its two elements, straight after the spline backward), acc[1][0][0] and the
unscale factor into the gh buffer).  For every (tile, wave) whose gW share is wrong: are the lane's gradients already wrong
(upstream: recompute / spline) or still right (downstream: scaling / strip / product / accumulators)?
    python tools/probe/gw_dump_check.py --lib tools/probe/build/libfc_oldbwd_dump.so"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from flowconductor_amd import _hip, ops  # noqa: E402

_hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
from oracle import torch_oracle as O  # noqa: E402

dev = "cuda"
k, tails, d, d_t, n, hidden = 8, "linear", 64, 32, 4096, 64
p = 3 * k - 1
torch.manual_seed(0)
x = torch.randn(n, d) * 1.5
h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
b = torch.randn(d_t * p) * 0.3
cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
gy, gl = torch.randn(n, d), torch.randn(n)
kw = dict(wh_divisor=float(hidden) ** 0.5)
packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
packed_t = ops.pack_final_layer_transposed(w.to(dev), k, tails)
x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
rows = (h64 @ w64.T + b64).view(n, d_t, p)
out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
y64 = x64.clone().index_copy(1, cols.long(), out)
loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
(G,) = torch.autograd.grad(loss, rows, retain_graph=True)              # [n, d_t, p]
gw_ref = torch.autograd.grad(loss, w64)[0]
o = ops.rq_fused_linear_backward(x.to(dev), h.to(dev), gy.to(dev), gl.to(dev), packed, packed_t, cols.to(dev), num_bins=k,
                                 tails=tails, tail_bound=3.0, merged=False, **kw)
torch.cuda.synchronize()
gw = o[2].cpu().double()
scale = float(gw_ref.abs().max())
bad_rows = torch.nonzero((gw - gw_ref).abs().amax(dim=1) > 1e-3 * scale).flatten().tolist()
print("bad rows", sorted({(r // p) for r in bad_rows}))
dump = o[1].cpu().double().view(n // 32, 8, 64, 4)                     # [tile, wave, lane, slot]
part = o[0].cpu().double().view(n // 32, 8, 4, 64)                     # FC_BWD_PARTIALS: share of tile in gW[(4 wave + 3), r, :]
Gt = G.detach().view(n // 32, 32, d_t, p)
refpart = torch.einsum("tsw,tsh->twh", Gt[:, :, 3::4, :4].reshape(n // 32, 32, 32), h.double().view(n // 32, 32, 64)).view(n // 32, 8, 4, 64)
bad = torch.nonzero((part - refpart).abs().amax(dim=(2, 3)) > 1e-4 * scale)
lanes = torch.arange(64)
g_of, s_of = lanes // 16, lanes % 16
for t_, w_ in bad[:8].tolist():
    exp0 = Gt[t_, s_of, 4 * w_ + g_of, 0]                 # gp[0][0] of lane (g, s16): sample s16, dim 4w + g, param 0
    exp1 = Gt[t_, 16 + s_of, 4 * w_ + g_of, 0]
    got0, got1 = dump[t_, w_, :, 0], dump[t_, w_, :, 1]
    e0 = ((got0 - exp0).abs() / (exp0.abs() + 1e-6)).view(4, 16).amax(dim=1)
    e1 = ((got1 - exp1).abs() / (exp1.abs() + 1e-6)).view(4, 16).amax(dim=1)
    print("tile %d wave %d: rel. error of gp[0][0] per lane group %s   gp[1][0] per lane group %s"
          % (t_, w_, ["%.1e" % v for v in e0.tolist()], ["%.1e" % v for v in e1.tolist()]))
    # which r of the partial is wrong, and is it explained by the lanes' dumped gradients?
    bad_r = torch.nonzero((part[t_, w_] - refpart[t_, w_]).abs().amax(dim=1) > 1e-4 * scale).flatten().tolist()
    print("      wrong share rows r =", bad_r, " acc[1][0][0] lanes 48..51:", dump[t_, w_, 48:52, 2].tolist())
print("clean" if not len(bad) else "%d bad (tile, wave) pairs" % len(bad))
