"""Per-sample [N, D, D] matrices applied to rows (ConditionalRotation / ConditionalLU, conditional.py:275-401): purely
HBM-bound, every matrix element is read once.  python tools/probe/bench_per_sample_linear.py [D] [log2 rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 17)
dev = "cuda"
x = torch.randn(n, d, device=dev)
m = torch.randn(n, d, d, device=dev) / d ** 0.5
for mode, name in ((ops.PER_SAMPLE_DENSE, "M x"), (ops.PER_SAMPLE_DENSE_T, "M^T x"), (ops.PER_SAMPLE_LU_FORWARD, "L (U x)"),
                   (ops.PER_SAMPLE_LU_INVERSE, "U^-1 L^-1 x")):
    with torch.no_grad():
        for _ in range(2):
            ops.linear_per_sample(x, m, mode=mode, offdiag_scale=0.1, eps=1e-3, want_logabsdet=True)
        with ops.KernelTimer("fc_linear_per_sample") as t:
            for _ in range(5):
                ops.linear_per_sample(x, m, mode=mode, offdiag_scale=0.1, eps=1e-3, want_logabsdet=True)
        torch.cuda.synchronize()
    ms = sorted(t.durations_ms())[2]
    byts = n * (4 * d * d + 8 * d)
    print("%-12s N=%d D=%d: %.3f ms = %.2f TB/s (%.2f of 8 TB/s)" % (name, n, d, ms, byts / ms / 1e9, byts / ms / 1e9 / 8))
