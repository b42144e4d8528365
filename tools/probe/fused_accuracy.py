"""Error of fc_rq_spline_fused_linear against the float64 oracle, beside the oracle's own float32 evaluation (= the
reference's float32 path) on the same inputs: max / 99.99th percentile / rms of |y - y64| and |lad - lad64|, forward and
inverse.  Usage: python tools/probe/fused_accuracy.py [--lib path.so] [--n 65536]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def stats(name, a, b):
    e = (a.double().cpu() - b.double().cpu()).abs().flatten()
    q = torch.quantile(e[torch.randperm(e.numel())[:1 << 20]], 0.9999) if e.numel() > 1 else e.max()
    print("  %-22s max %.3e  p99.99 %.3e  rms %.3e" % (name, float(e.max()), float(q), float((e ** 2).mean().sqrt())))


def main():
    argv = sys.argv[1:]
    n = 1 << 16
    if "--lib" in argv:
        i = argv.index("--lib")
        _hip.use_library(argv[i + 1])
        del argv[i:i + 2]
    if "--n" in argv:
        i = argv.index("--n")
        n = int(argv[i + 1])
    dev = torch.device("cuda:0")
    d, d_t, k, hidden = 64, 32, 8, 64
    for seed, wscale in ((0, 0.2), (1, 0.6)):
        torch.manual_seed(seed)
        x = torch.randn(n, d) * 1.5
        h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
        w = torch.randn(d_t * (3 * k - 1), hidden) * wscale
        b = torch.randn(d_t * (3 * k - 1)) * 0.1
        cols = torch.arange(0, d, 2)
        rows64 = (h.double() @ w.double().T + b.double()).view(n, d_t, 3 * k - 1)
        rows32 = rows64.float().clone()
        wp, bp = ops.pack_final_layer(w.to(dev), b.to(dev))
        for inverse in (False, True):
            y64, l64 = O.rq_from_rows(x[:, cols].double(), rows64.clone(), k, "linear", 3.0, inverse, wh_divisor=float(hidden) ** 0.5)
            y32, l32 = O.rq_from_rows(x[:, cols], rows32.clone(), k, "linear", 3.0, inverse, wh_divisor=float(hidden) ** 0.5)
            with torch.no_grad():
                y, lad = ops.rq_spline_fused_linear(x.to(dev), h.to(dev), wp, bp, cols.to(dev).int(), num_bins=k, tail_bound=3.0,
                                                    wh_divisor=float(hidden) ** 0.5, inverse=inverse)
            print("weights x%.1f %s" % (wscale, "inverse" if inverse else "forward"))
            stats("oracle f32: y", y32, y64)
            stats("kernel:     y", y[:, cols.to(dev)], y64)
            stats("oracle f32: logabsdet", l32.sum(1), l64.sum(1))
            stats("kernel:     logabsdet", lad, l64.sum(1))


if __name__ == "__main__":
    main()
