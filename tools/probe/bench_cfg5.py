"""BASELINE.json configs[4]: Sylvester flow, D = 128, M = 32 Householder vectors, batch 2^18 (shared weights)
-- time of the fused Sylvester kernel and of the Householder-only kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

n, d, m = 1 << 18, 128, 32
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(n, d, device=dev)
q = torch.randn(m, d, device=dev)
r1 = torch.triu(torch.randn(d, d, device=dev) / d ** 0.5)
r2 = torch.triu(torch.randn(d, d, device=dev) / d ** 0.5)
r1.diagonal().copy_(torch.tanh(r1.diagonal()))
r2.diagonal().copy_(torch.tanh(r2.diagonal()))
bias = torch.randn(d, device=dev) * 0.1
for name, fn in (("fc_sylvester", lambda: ops.sylvester(x, q, r1, r2, bias)),
                 ("fc_householder", lambda: ops.householder(x, q))):
    with torch.no_grad():
        for _ in range(3):
            fn()
        with ops.KernelTimer(name) as t:
            for _ in range(10):
                fn()
        torch.cuda.synchronize()
    ms = sorted(t.durations_ms())
    print("%s N=2^18 D=128 M=32: median %.3f ms" % (name, ms[len(ms) // 2]))

from flowconductor_amd.transforms import SylvesterTransform  # noqa: E402

t = SylvesterTransform(features=d, num_householder=m).to(dev).eval()
with torch.no_grad():
    t.Q_orth.q_vectors.copy_(q)
    for _ in range(3):
        t(x)
    with ops.KernelTimer("fc_sylvester_mm") as tm:
        for _ in range(10):
            t(x)
    torch.cuda.synchronize()
ms = sorted(tm.durations_ms())
print("fc_sylvester_mm (SylvesterTransform.forward) N=2^18 D=128 M=32: median %.3f ms" % ms[len(ms) // 2])

# conditional variant: per-sample q [N, 32, 128], R1 / R2 [N, 128, 128], bias [N, 128] -- 148 480 B per
# sample, HBM-bound by construction
for log2n in (16, 18):
    nn_ = 1 << log2n
    xs = torch.randn(nn_, d, device=dev)
    qs = torch.randn(nn_, m, d, device=dev)
    r1s = torch.triu(torch.randn(nn_, d, d, device=dev) / d ** 0.5)
    r2s = torch.triu(torch.randn(nn_, d, d, device=dev) / d ** 0.5)
    r1s.diagonal(dim1=1, dim2=2).tanh_()
    r2s.diagonal(dim1=1, dim2=2).tanh_()
    bs = torch.randn(nn_, d, device=dev) * 0.1
    with torch.no_grad():
        for _ in range(2):
            ops.sylvester(xs, qs, r1s, r2s, bs)
        with ops.KernelTimer("fc_sylvester") as t:
            for _ in range(5):
                ops.sylvester(xs, qs, r1s, r2s, bs)
        torch.cuda.synchronize()
    ms = sorted(t.durations_ms())
    med = ms[len(ms) // 2]
    byts = nn_ * (4 * (2 * d * d + 2 * d + m * d) + 4 * d + 4)
    print("fc_sylvester per-sample parameters N=2^%d D=128 M=32: median %.3f ms = %.2f TB/s of %d B/sample (%.2f of 8 TB/s)"
          % (log2n, med, byts / med / 1e9, byts // nn_, byts / med / 1e9 / 8.0))
    del xs, qs, r1s, r2s, bs
    torch.cuda.empty_cache()
