"""Launch time of fc_sylvester_mm (shared-parameter Sylvester, D = 128) against the batch: the intercept is the per-workgroup
prologue (scaling, splitting and laying out both matrices).  python tools/probe/bench_sylvester_mm.py [--lib probe.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import _hip, ops  # noqa: E402

if "--lib" in sys.argv:          # a probe build (e.g. -DFC_SYL_BPW=2)
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])

dev = torch.device("cuda:0")
torch.manual_seed(0)
d, m = 128, 32
q = torch.randn(m, d, device=dev)
r1 = torch.triu(torch.randn(d, d, device=dev) * 0.1)
r2 = torch.triu(torch.randn(d, d, device=dev) * 0.1)
w1, w2, rdiag = ops.pack_sylvester(q, r1, r2)
bias = torch.randn(d, device=dev) * 0.1
for log2n in (4, 12, 15, 18, 20):
    n = 1 << log2n
    x = torch.randn(n, d, device=dev)
    for _ in range(5):
        ops.sylvester_mm(x, w1, w2, bias, rdiag)
    best = 1e9
    for _ in range(20):
        with ops.KernelTimer("fc_sylvester_mm") as t:
            ops.sylvester_mm(x, w1, w2, bias, rdiag)
        torch.cuda.synchronize()
        best = min(best, min(t.durations_ms()))
    y_chk, lad_chk = ops.sylvester_mm(x, w1, w2, bias, rdiag)
    chk = (float(y_chk.double().sum()), float(lad_chk.double().sum()))
    alg = n * (2 * d + 1) * 4
    print("N=2^%d: %.4f ms  (%.0f GB/s algorithmic)  checksums %.6f %.6f" % (log2n, best, alg / best / 1e6, chk[0], chk[1]))
