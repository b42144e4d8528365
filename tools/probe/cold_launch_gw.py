"""First launch of a fresh process: gW of fc_rq_fused_linear_backward (roles 0 + 1) at N = 4096 against float64 autograd, with the
per-tile shares of a probe build when given (FC_BWD_PARTIALS).  Round 2 history: with the fragment addresses held as two dozen
spilled 64-bit VGPR pointers, about every second cold launch produced a wrong [dim 4 w + 3, widths] slice for one tile
(second wave of a SIMD, lane group 3); with scalar bases (no address spills) 0 of 44.  Run it several times, each in a new
process:  for i in 1 2 3 4 5 6 7 8; do python tools/probe/cold_launch_gw.py [--k 10] [--lib tools/probe/build/libfc_ablN.so]; done"""
import os, sys, torch
sys.path[:0] = [os.getcwd()]
from flowconductor_amd import ops, _hip
if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
from oracle import torch_oracle as O
dev = "cuda"
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 8
tails, d, d_t, n, seed = "linear", 64, 32, 4096, 0
torch.manual_seed(seed)
hidden = 64
p = 3 * k - 1
x = torch.randn(n, d) * 1.5
h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
b = torch.randn(d_t * p) * 0.3
cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
gy, gl = torch.randn(n, d), torch.randn(n)
kw = dict(wh_divisor=float(hidden) ** 0.5)
packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
packed_t = ops.pack_final_layer_transposed(w.to(dev), k, tails)

def ref(sl):
    x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x[sl], h[sl], w, b))
    m = x64.shape[0]
    rows = (h64 @ w64.T + b64).view(m, d_t, p)
    out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
    y64 = x64.clone().index_copy(1, cols.long(), out)
    loss = (y64 * gy[sl].double()).sum() + (lad_e.sum(dim=1) * gl[sl].double()).sum()
    # per-sample G for the feature under suspicion
    (grows,) = torch.autograd.grad(loss, rows, retain_graph=True)
    return torch.autograd.grad(loss, w64)[0], grows

def gpu(sl, merged=False):
    return ops.rq_fused_linear_backward(x[sl].to(dev), h[sl].to(dev), gy[sl].to(dev), gl[sl].to(dev), packed, packed_t, cols.to(dev),
                                        num_bins=k, tails=tails, tail_bound=3.0, merged=merged, **kw)[2].cpu().double()

full_ref, grows = ref(slice(0, n))
scale = float(full_ref.abs().max())
out = ops.rq_fused_linear_backward(x.to(dev), h.to(dev), gy.to(dev), gl.to(dev), packed, packed_t, cols.to(dev),
                                   num_bins=k, tails=tails, tail_bound=3.0, merged=False, **kw)
full, dbg = out[2].cpu().double(), out[0].cpu().view(n // 32, 2048)
e = (full - full_ref)
rows_bad = torch.nonzero(e.abs().amax(dim=1) > 1e-3 * scale).flatten().tolist()
print("first launch: bad rows", [(r // p, r % p) for r in rows_bad])
part = dbg.view(n // 32, 8, 4, 64).double()            # [tile, wave, r, hidden]: share of tile in gW[(4 wave + 3), r, :]
hd = h.double()
G = grows.detach()                                        # [n, d_t, p]
refpart = torch.einsum("tsw,tsh->twh", G.view(n // 32, 32, d_t, p)[:, :, 3::4, :4].reshape(n // 32, 32, 32), hd.view(n // 32, 32, 64)).view(n // 32, 8, 4, 64)
err = (part - refpart).abs().amax(dim=3)                  # [tile, wave, r]
bad = torch.nonzero(err > 1e-4 * scale)
print("bad (tile, wave, r):", bad.tolist()[:40])
for t_, w_, r_ in bad[:6].tolist():
    pr, rr = part[t_, w_, r_], refpart[t_, w_, r_]
    print("  tile %d wave %d r %d: got %s\n      ref %s" % (t_, w_, r_, ["%.3e" % v for v in pr[:8].tolist()], ["%.3e" % v for v in rr[:8].tolist()]))
    # per-sample decomposition: which samples are missing?  solve err = sum_s c_s G_s h_s restricted to the tile (32 unknowns, 64 equations)
    Gs = G[32 * t_:32 * t_ + 32, 4 * w_ + 3, r_]
    A = (Gs.unsqueeze(1) * hd[32 * t_:32 * t_ + 32]).T       # [64, 32]
    sol = torch.linalg.lstsq(A, (pr - rr).unsqueeze(1)).solution.flatten()
    print("      per-sample factor of the error (0 = sample right, -1 = sample missing):", ["%.2f" % v for v in sol.tolist()])
print("done")
