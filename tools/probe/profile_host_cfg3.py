"""Host (Python) time per `log_prob` of the cfg-3 flow at a small batch, where the kernels are negligible: what the eager
path costs per coupling layer, and where (cProfile).  python tools/probe/profile_host_cfg3.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
    utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
    lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2), num_bins=8, tails="linear", tail_bound=3.0)
    for i in range(32)]
flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([64])).eval().to(dev)
x = torch.randn(4096, 64, device=dev)
with torch.no_grad():
    for _ in range(10):
        flow.log_prob(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        flow.log_prob(x)
    torch.cuda.synchronize()
    print("per log_prob call: %.1f us (32 layers, 64 launches)" % ((time.perf_counter() - t0) / 100 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        flow.log_prob(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
