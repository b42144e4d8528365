"""fc_rq_spline_fused_linear launch time against the row count (D = 64, 32 transformed dims): the intercept is the
per-launch set-up (weight rows -> scaled f16 pieces in registers, LDS tables)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
w = torch.randn(32 * 23, 64, device=dev) * 0.2
b = torch.randn(32 * 23, device=dev) * 0.1
wp, bp = ops.pack_final_layer(w, b)
cols = torch.arange(1, 64, 2, dtype=torch.int32, device=dev)
for log2n in (10, 14, 16, 18, 20):
    n = 1 << log2n
    x = torch.randn(n, 64, device=dev)
    h = torch.randn(n, 64, device=dev)
    with torch.no_grad():
        for _ in range(5):
            ops.rq_spline_fused_linear(x, h, wp, bp, cols, num_bins=8, tail_bound=3.0, wh_divisor=8.0)
        with ops.KernelTimer("fc_rq_spline_fused_linear") as t:
            for _ in range(40):
                ops.rq_spline_fused_linear(x, h, wp, bp, cols, num_bins=8, tail_bound=3.0, wh_divisor=8.0)
    torch.cuda.synchronize()
    ms = sorted(t.durations_ms())
    print(f"N=2^{log2n}: median {ms[len(ms) // 2] * 1e3:.1f} us  min {ms[0] * 1e3:.1f} us")
