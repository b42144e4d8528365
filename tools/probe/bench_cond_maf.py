"""Conditional MAF in the size range of simulation-based inference (5 x [MAF(D = 8, hidden 50, 2 blocks, context 10),
reverse permutation]): log_prob(x | c) with the MADE hidden stacks in fc_resnet_hidden_context (additive mode) vs on
PyTorch-ROCm kernels.  python tools/probe/bench_cond_maf.py [log2 rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.utils.graphs import GraphedCall  # noqa: E402


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 12)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = []
    for _ in range(5):
        layers.append(transforms.MaskedAffineAutoregressiveTransform(8, 50, context_features=10, num_blocks=2))
        layers.append(transforms.ReversePermutation(8))
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([8])).to(dev).eval()
    x = torch.randn(n, 8, device=dev)
    c = torch.randn(n, 10, device=dev)
    res = {}
    with torch.no_grad():
        for _ in range(2):
            for mode in ("1", "0"):
                options._values["fused_hidden"] = mode == "1"
                res[mode] = min(res.get(mode, 1e9), timed(lambda: flow.log_prob(x, c)))
        options._values["fused_hidden"] = True
        graphed = GraphedCall(flow.log_prob, x, c)
        res["graph"] = timed(lambda: graphed(x, c))
    print(f"conditional MAF N={n}: hidden kernel {res['1']:.3f} ms, PyTorch hidden layers {res['0']:.3f} ms "
          f"(x{res['0'] / res['1']:.1f}), hidden kernel as one HIP graph {res['graph']:.3f} ms")


if __name__ == "__main__":
    main()
