"""BASELINE.json configs[1]: 8-layer affine-coupling flow, D = 32, batch 2^18 -- log_prob throughput."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms, utils  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

torch.manual_seed(0)
layers = [transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(32, even=(i % 2 == 0)),
                                             lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2))
          for i in range(8)]
flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([32])).to("cuda").eval()
x = torch.randn(1 << 18, 32, device="cuda")
with torch.no_grad():
    for _ in range(3):
        lp = flow.log_prob(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        lp = flow.log_prob(x)
    torch.cuda.synchronize()
dt = (time.time() - t0) / 20
print("cfg2: %.3f ms per log_prob of 2^18 samples -> %.1f M samples/s (FC_FUSED_HIDDEN=%s)"
      % (dt * 1e3, (1 << 18) / dt / 1e6, options.get("fused_hidden")))
