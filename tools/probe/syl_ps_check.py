"""Per-sample Sylvester kernel against a float64 torch restatement (planar.py:144-166 with per-sample parameters);
python tools/probe/syl_ps_check.py [--lib path.so]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
torch.manual_seed(0)
for d, m, n in ((70, 3, 65), (128, 32, 64), (16, 1, 8), (16, 9, 8)):
    x = torch.randn(n, d, dtype=torch.float64)
    q = torch.randn(n, max(m, 1), d, dtype=torch.float64)[:, :m]
    r1 = torch.randn(n, d, d, dtype=torch.float64) / d ** 0.5
    r2 = torch.randn(n, d, d, dtype=torch.float64) / d ** 0.5
    b = torch.randn(n, d, dtype=torch.float64)

    def house(v, rev):
        order = range(m - 1, -1, -1) if rev else range(m)
        for k in order:
            qk = q[:, k]
            v = v - 2 * (v * qk).sum(1, keepdim=True) / (qk * qk).sum(1, keepdim=True) * qk
        return v
    t = house(x, True)
    a = torch.tanh(torch.einsum("nij,nj->ni", torch.triu(r1), t) + b)
    ref = x + house(torch.einsum("nij,nj->ni", torch.triu(r2), a), False)
    dev = "cuda"
    f = lambda v: v.float().to(dev).contiguous()
    y, lad = ops.sylvester(f(x), f(q), f(r1), f(r2), f(b))
    print(d, m, n, "max err", float((y.cpu().double() - ref).abs().max()))
