"""Host time of the README flow (BASELINE.json configs[0]) per eager log_prob call, with a cProfile of where it goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = []
for _ in range(2):
    layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
    layers.append(transforms.RandomPermutation(features=2))
flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).eval().to(dev)
x = torch.randn(4096, 2, device=dev)
with torch.no_grad():
    for _ in range(50):
        flow.log_prob(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        flow.log_prob(x)
    torch.cuda.synchronize()
    print("per log_prob call: %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(500):
        flow.log_prob(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
