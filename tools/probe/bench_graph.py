"""Launch-bound small batches: eager Flow.log_prob vs its HIP-graph replay (utils/graphs.GraphedCall)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402
from flowconductor_amd.utils.graphs import GraphedCall  # noqa: E402


def timeit(fn, x, reps=50):
    with torch.no_grad():
        for _ in range(5):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn(x)
        torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


torch.manual_seed(0)
# BASELINE.json configs[0]: README flow, D = 2, N = 4096
layers = []
for _ in range(2):
    layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
    layers.append(transforms.RandomPermutation(features=2))
toy = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).to("cuda").eval()
x = torch.randn(4096, 2, device="cuda")
e, g = timeit(toy.log_prob, x), timeit(GraphedCall(toy.log_prob, x, clone=False), x)
print("cfg 1 (README MAF flow, N=4096): eager %.3f ms, HIP graph %.3f ms  (x%.1f)" % (e, g, e / g))
# cfg 3 flow at a small batch
layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
    utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
    lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2), num_bins=8, tails="linear", tail_bound=3.0)
    for i in range(32)]
nsf = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([64])).to("cuda").eval()
for n in (4096, 32768):
    x = torch.randn(n, 64, device="cuda")
    e, g = timeit(nsf.log_prob, x), timeit(GraphedCall(nsf.log_prob, x, clone=False), x)
    print("cfg 3 flow at N=%d: eager %.3f ms, HIP graph %.3f ms  (x%.1f)" % (n, e, g, e / g))
