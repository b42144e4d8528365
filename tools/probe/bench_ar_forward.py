"""Density direction of an RQ-spline autoregressive layer (MADE hidden 64, 2 blocks, K = 8, linear tails): hidden stack
+ fused masked final Linear + spline vs masked final Linear as a library GEMM + fc_rq_spline.
python tools/probe/bench_ar_forward.py [features] [log2 rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import transforms as T  # noqa: E402
from flowconductor_amd import options  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    features = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    t = T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features, 64, num_blocks=2, num_bins=8, tails="linear", tail_bound=3.0).to(dev).eval()
    x = torch.randn(n, features, device=dev)
    with torch.no_grad():
        fused = unfused = float("inf")
        for _ in range(3):
            options._values["fused_final_layer"] = True
            fused = min(fused, timed(lambda: t(x)))
            y1, l1 = t(x)
            options._values["fused_final_layer"] = False
            unfused = min(unfused, timed(lambda: t(x)))
            y0, l0 = t(x)
    print(f"rq_ar forward D={features} N={n}: fused {fused:.3f} ms ({n / fused / 1e3:.0f} M samples/s), "
          f"unfused {unfused:.3f} ms  x{unfused / fused:.1f}  max|dy| {float((y1 - y0).abs().max()):.2e}  "
          f"max|dlad| {float((l1 - l0).abs().max()):.2e}")


if __name__ == "__main__":
    main()
