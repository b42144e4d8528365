"""Launch time of one affine coupling layer (fc_affine_coupling_resnet, D = 32, ResidualNet hidden 64 x 2 blocks) against the
batch: the intercept is the per-workgroup prologue (weight image into LDS) + launch.  python tools/probe/bench_affine_layer.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
d = 32
t = transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(d, even=True),
                                       lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2)).to(dev).eval()
with torch.no_grad():
    for log2n in (8, 12, 14, 16, 18, 20):
        x = torch.randn(1 << log2n, d, device=dev)
        for _ in range(5):
            t(x)
        best = 1e9
        for _ in range(20):
            with ops.KernelTimer("fc_affine_coupling_resnet") as tm:
                t(x)
            torch.cuda.synchronize()
            best = min(best, min(tm.durations_ms()))
        print("N=2^%d: %.4f ms" % (log2n, best))
