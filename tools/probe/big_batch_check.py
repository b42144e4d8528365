"""Index arithmetic beyond 2^31 elements: one cfg-3 RQ coupling layer on N = 2^22 rows (the [N, 736] parameter tensor has
3.1e9 elements, 12.4 GB), stand-alone spline kernel and fused path; the last rows must equal the same rows evaluated
alone.  python tools/probe/big_batch_check.py [log2 rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import transforms, utils  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
dev = torch.device("cuda:0")
torch.manual_seed(0)
t = transforms.PiecewiseRationalQuadraticCouplingTransform(
    utils.create_alternating_binary_mask(64), lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2),
    num_bins=8, tails="linear", tail_bound=3.0).to(dev).eval()
with torch.no_grad():
    for p in t.parameters():
        p.mul_(1.5)
x = torch.randn(n, 64, device=dev)
tail = slice(n - 4096, n)
for mode in ("1", "0"):
    options._values["fused_final_layer"] = mode == "1"
    options._values["fused_hidden"] = mode == "1"
    with torch.no_grad():
        y, lad = t(x)
        y_t, lad_t = t(x[tail].contiguous())
        xb, ladb = t.inverse(y)
    torch.cuda.synchronize()
    print("FC_FUSED=%s N=2^%d: last 4096 rows vs alone: max|dy| %.2e max|dlad| %.2e; round trip max %.2e; peak %.1f GiB"
          % (mode, n.bit_length() - 1, float((y[tail] - y_t).abs().max()), float((lad[tail] - lad_t).abs().max()),
             float((xb - x).abs().max()), torch.cuda.max_memory_allocated() / 2**30))
    del y, lad, xb, ladb
    torch.cuda.empty_cache()
