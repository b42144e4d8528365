"""Per-sample Householder sequences q [N, K, D] (ConditionalOrthogonal / ConditionalSVD, conditional.py:404-603):
HBM-bound by the q rows.  python tools/probe/bench_per_sample_householder.py [D] [K] [log2 rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 18)
dev = "cuda"
x = torch.randn(n, d, device=dev)
q = torch.randn(n, k, d, device=dev)
with torch.no_grad():
    for _ in range(2):
        ops.householder(x, q)
    with ops.KernelTimer("fc_householder") as t:
        for _ in range(5):
            ops.householder(x, q)
    torch.cuda.synchronize()
ms = sorted(t.durations_ms())[2]
byts = n * (4 * k * d + 8 * d)
print("householder per-sample N=%d K=%d D=%d: %.3f ms = %.2f TB/s (%.2f of 8 TB/s)" % (n, k, d, ms, byts / ms / 1e9, byts / ms / 1e9 / 8))
