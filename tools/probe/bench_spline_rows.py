"""Launch time of the sibling spline coupling bijectors (fc_piecewise_spline tile kernel) at the shape of `configs.kernels`:
python tools/probe/bench_spline_rows.py [--lib probe.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import _hip, ops, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
torch.manual_seed(0)
d, n = 64, 1 << 18
x = torch.randn(n, d, device=dev)
mask = utils.create_alternating_binary_mask(d, even=True)
out = []
for name, cls in (("linear", transforms.PiecewiseLinearCouplingTransform), ("quadratic", transforms.PiecewiseQuadraticCouplingTransform),
                  ("cubic", transforms.PiecewiseCubicCouplingTransform)):
    t = cls(mask, lambda i, o: nets.ResidualNet(i, o, hidden_features=64, num_blocks=2), num_bins=8, tails="linear",
            tail_bound=3.0).to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            t(x)
        best = 1e9
        for _ in range(10):
            with ops.KernelTimer("fc_piecewise_spline") as tm:
                t(x)
            torch.cuda.synchronize()
            best = min(best, min(tm.durations_ms()))
    out.append("%s %.3f ms" % (name, best))
print("; ".join(out))
