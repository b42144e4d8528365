"""Autoregressive inverse (sampling direction): the D passes inside one kernel (fc_made_inverse, round 4) vs the host loop of
column-at-a-time passes vs the reference's D full passes (autoregressive.py:44-53).
python tools/probe/bench_ar_inverse.py [features] [log2 rows]"""
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
    features = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cases = {
        "maf": T.MaskedAffineAutoregressiveTransform(features, 64, num_blocks=2),
        "rq_ar_k8": T.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            features, 64, num_blocks=2, num_bins=8, tails="linear", tail_bound=3.0),
    }
    z = torch.randn(n, features, device=dev)
    for name, t in cases.items():
        t = t.to(dev).eval()
        with torch.no_grad():
            dev_loop = inc = full = float("inf")
            for _ in range(3):          # alternate: the first measurements of a process run on a cold device
                options._values["ar_device_loop"] = True
                dev_loop = min(dev_loop, timed(lambda: t.inverse(z)))
                y2, l2 = t.inverse(z)
                options._values["ar_device_loop"] = False
                options._values["ar_incremental"] = "force"
                inc = min(inc, timed(lambda: t.inverse(z)))
                y1, l1 = t.inverse(z)
                options._values["ar_incremental"] = "off"
                full = min(full, timed(lambda: t.inverse(z)))
                y0, l0 = t.inverse(z)
            options._values["ar_device_loop"] = True
            options._values["ar_incremental"] = "auto"
        print(f"{name}: D={features} N={n}  device loop {dev_loop:.3f} ms  column-at-a-time {inc:.2f} ms  full passes {full:.2f} ms"
              f"  x{full / dev_loop:.1f} / x{inc / dev_loop:.1f}  max|dy| {float((y2 - y0).abs().max()):.2e}"
              f"  max|dlad| {float((l2 - l0).abs().max()):.2e}")


if __name__ == "__main__":
    main()
