import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops
n, d, d_t, k = 1 << 20, 64, 32, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(n, d, device=dev) * 1.5
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
params = torch.randn(n, d_t * 23, device=dev)
gy, gl = torch.randn(n, d, device=dev), torch.randn(n, device=dev)
kw = dict(num_bins=k, tails="linear", tail_bound=3.0, wh_divisor=8.0)
for _ in range(3): gx, gp = ops.rq_spline_backward(x, params, cols, gy, gl, **kw)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): gx, gp = ops.rq_spline_backward(x, params, cols, gy, gl, **kw)
torch.cuda.synchronize(); t1 = time.time()
print("wall per call %.3f ms (includes the grad clone)" % ((t1 - t0) / 20 * 1e3))
# rows at the end of the batch against a small launch of the same rows
sl = slice(n - 4096, n)
gx2, gp2 = ops.rq_spline_backward(x[sl].contiguous(), params[sl].contiguous(), cols, gy[sl].contiguous(), gl[sl].contiguous(), **kw)
print("tail rows equal:", torch.equal(gx[sl], gx2), torch.equal(gp[sl], gp2), float(gp.abs().mean()))
