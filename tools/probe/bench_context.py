"""Conditional cfg-3 flow (32 x RQ coupling, D = 64, K = 8, ResidualNet(64, 2 blocks) with a 16-feature context):
log_prob(x | c) with the hidden layers in fc_resnet_hidden_context vs on PyTorch-ROCm kernels (FC_FUSED_HIDDEN=0).
python tools/probe/bench_context.py [log2 rows] [context features]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, ops, transforms, utils  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
    ctx_f = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=64, context_features=ctx_f, num_blocks=2),
        num_bins=8, tails="linear", tail_bound=3.0) for i in range(32)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([64])).to(dev).eval()
    x = torch.randn(n, 64, device=dev)
    c = torch.randn(n, ctx_f, device=dev)

    def timed(reps=3):
        with torch.no_grad():
            for _ in range(2):
                lp = flow.log_prob(x, c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                lp = flow.log_prob(x, c)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, lp

    best = {}
    for _ in range(2):
        for mode in ("1", "0"):
            options._values["fused_hidden"] = mode == "1"
            ms, lp = timed()
            if mode not in best or ms < best[mode][0]:
                best[mode] = (ms, lp)
    with torch.no_grad(), ops.KernelTimer("fc_resnet_hidden_context") as t:
        options._values["fused_hidden"] = True
        flow.log_prob(x, c)
    k = t.durations_ms()
    d = float((best["1"][1] - best["0"][1]).abs().max())
    print(f"conditional cfg 3, N={n}, context {ctx_f}: hidden kernel {best['1'][0]:.2f} ms/step "
          f"({n / best['1'][0] / 1e3:.1f} M samples/s), PyTorch hidden layers {best['0'][0]:.2f} ms/step "
          f"({n / best['0'][0] / 1e3:.1f} M samples/s); fc_resnet_hidden_context {sum(k) / len(k):.3f} ms/launch; "
          f"max |d log_prob| {d:.2e}")


if __name__ == "__main__":
    main()
