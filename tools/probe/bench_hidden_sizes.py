"""fc_resnet_hidden launch time against the row count: the intercept is the per-launch weight set-up."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nets.ResidualNet(32, 8, hidden_features=64, num_blocks=2).to(dev).eval()
ids = torch.arange(0, 64, 2, dtype=torch.int32, device=dev)
for log2n in (10, 14, 16, 18, 20):
    x = torch.randn(1 << log2n, 64, device=dev)
    with torch.no_grad():
        for _ in range(5):
            net.hidden_hip(x, ids)
        with ops.KernelTimer("fc_resnet_hidden") as t:
            for _ in range(40):
                net.hidden_hip(x, ids)
    torch.cuda.synchronize()
    ms = sorted(t.durations_ms())
    print(f"N=2^{log2n}: median {ms[len(ms) // 2] * 1e3:.1f} us  min {ms[0] * 1e3:.1f} us")
