"""Operator-level repro: gW of fc_rq_fused_linear_backward at larger N against float64 autograd."""
import os, sys, torch
sys.path[:0] = [os.getcwd()]
from flowconductor_amd import ops
from oracle import torch_oracle as O
dev = "cuda"

def run(k, tails, d, d_t, n, seed, hscale=True, wscale=None):
    torch.manual_seed(seed)
    hidden = 64
    p = 3 * k - 1 if tails == "linear" else 3 * k + 1
    x = torch.randn(n, d) * 1.5
    h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
    if hscale:
        h *= torch.logspace(-2, 1, n).unsqueeze(1)
    w = torch.randn(d_t * p, hidden) * (wscale or 1.0 / hidden ** 0.5)
    b = torch.randn(d_t * p) * 0.3
    cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
    gy, gl = torch.randn(n, d), torch.randn(n)
    kw = dict(wh_divisor=float(hidden) ** 0.5)
    x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
    rows = (h64 @ w64.T + b64).view(n, d_t, p)
    out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
    y64 = x64.clone().index_copy(1, cols.long(), out)
    loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
    gx_ref, gh_ref, gw_ref, gb_ref = torch.autograd.grad(loss, (x64, h64, w64, b64))
    packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
    packed_t = ops.pack_final_layer_transposed(w.to(dev), k, tails)
    res = {}
    for merged in (False, True):
        gx, gh, gw, gb = ops.rq_fused_linear_backward(x.to(dev), h.to(dev), gy.to(dev), gl.to(dev), packed, packed_t, cols.to(dev),
                                                      num_bins=k, tails=tails, tail_bound=3.0, merged=merged, **kw)
        res[merged] = gw.cpu().double()
        e = (res[merged] - gw_ref).abs()
        idx = torch.nonzero(e > 1e-3 * gw_ref.abs().max())
        print("k %d d %d d_t %d n %d seed %d merged %s: gW relerr %.2e  gb relerr %.2e  gh %.2e gx %.2e bad entries %d %s" % (
            k, d, d_t, n, seed, merged, e.max() / gw_ref.abs().max(), (gb.cpu().double() - gb_ref).abs().max() / gb_ref.abs().max(),
            (gh.cpu().double() - gh_ref).abs().max() / gh_ref.abs().max(), (gx.cpu().double() - gx_ref).abs().max() / gx_ref.abs().max(),
            idx.shape[0], [(int(i) // p, int(i) % p, int(j)) for i, j in idx[:6]]))

for seed in range(4):
    run(8, "linear", 64, 32, 4096, seed, hscale=False)
run(8, "linear", 64, 32, 4096, 0, hscale=True)
run(8, "linear", 64, 32, 1024, 0, hscale=False)
run(10, "linear", 64, 32, 4096, 0, hscale=False)
