"""RQ coupling flow (K = 8, linear tails, ResidualNet(64, 2 blocks)) of any width: fused path vs FC_FUSED=0
(final Linear as a library GEMM + fc_rq_spline).  python tools/probe/bench_coupling_widths.py [D] [layers] [log2 rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, transforms, utils  # noqa: E402
from flowconductor_amd import options  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 19)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(d, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2),
        num_bins=8, tails="linear", tail_bound=3.0) for i in range(nl)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([d])).to(dev).eval()
    x = torch.randn(n, d, device=dev)

    def timed(reps=5):
        with torch.no_grad():
            for _ in range(2):
                lp = flow.log_prob(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                lp = flow.log_prob(x)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, lp

    best = {}
    for _ in range(2):
        for mode in ("1", "0"):
            options._values["fused_final_layer"] = mode == "1"
            ms, lp = timed()
            if mode not in best or ms < best[mode][0]:
                best[mode] = (ms, lp)
    print(f"D={d} layers={nl} N={n}: fused {best['1'][0]:.2f} ms ({n / best['1'][0] / 1e3:.1f} M samples/s), "
          f"unfused {best['0'][0]:.2f} ms  x{best['0'][0] / best['1'][0]:.1f}  "
          f"max |d log_prob| {float((best['1'][1] - best['0'][1]).abs().max()):.2e}")


if __name__ == "__main__":
    main()
