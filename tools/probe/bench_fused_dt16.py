import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops
n, d, d_t, k = 1 << 20, 32, 16, 8
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(n, d, device=dev) * 1.5
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
h = torch.randn(n, 64, device=dev)
w = torch.randn(d_t * 23, 64, device=dev) * 0.2
b = torch.randn(d_t * 23, device=dev) * 0.1
wp, bp = ops.pack_final_layer(w, b)
kw = dict(num_bins=k, tail_bound=3.0, wh_divisor=8.0)
with torch.no_grad():
    for _ in range(3): ops.rq_spline_fused_linear(x, h, wp, bp, cols, **kw)
    with ops.KernelTimer("fc_rq_spline_fused_linear") as t:
        for _ in range(20): ops.rq_spline_fused_linear(x, h, wp, bp, cols, **kw)
    torch.cuda.synchronize()
    ms = sorted(t.durations_ms()); print("fused d_t=16 D=32 N=2^20: median %.3f ms" % ms[10])
    params = torch.randn(n, d_t * 23, device=dev)
    lin = torch.nn.Linear(64, d_t * 23).to(dev)
    for _ in range(3): p2 = lin(h); ops.rq_spline(x, p2, cols, tails="linear", **kw)
    torch.cuda.synchronize()
    import time; t0 = time.time()
    for _ in range(20): p2 = lin(h); ops.rq_spline(x, p2, cols, tails="linear", **kw)
    torch.cuda.synchronize(); print("unfused Linear + fc_rq_spline: %.3f ms" % ((time.time() - t0) / 20 * 1e3))
