"""Error distribution of the HIP RQ-spline kernel vs float64 truth, next to the float32 CPU
oracle's own error vs the same truth (the reference's float32 noise floor).

    python tools/noise_floor.py            # needs a GPU; prints a table
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def stats(name, err):
    e = err.double().abs().flatten().numpy()
    print("  %-34s median %.2e  p99 %.2e  p99.99 %.2e  max %.2e" % (
        name, np.median(e), np.percentile(e, 99), np.percentile(e, 99.99), e.max()))


def main():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda:0")
    n, d, k, tb = 1 << 14, 32, 8, 3.0
    gen = torch.Generator().manual_seed(0)
    rows = torch.randn(n, d, 3 * k - 1, generator=gen)
    x = torch.randn(n, d, generator=gen) * 1.5
    for inverse in (False, True):
        print("inverse" if inverse else "forward", "N=%d d=%d K=%d params~N(0,1)" % (n, d, k))
        y64, l64 = O.rq_from_rows(x.double(), rows.double().clone(), k, "linear", tb, inverse)
        y32, l32 = O.rq_from_rows(x.clone(), rows.clone(), k, "linear", tb, inverse)
        with torch.no_grad():
            yg, lg = ops.rq_spline(x.to(dev), rows.reshape(n, -1).to(dev), None, num_bins=k, tails="linear",
                                   tail_bound=tb, inverse=inverse)
        yg, lg = yg.cpu(), lg.cpu()
        stats("oracle-f32 outputs vs f64", y32 - y64)
        stats("HIP        outputs vs f64", yg - y64)
        stats("HIP        outputs vs oracle-f32", yg - y32)
        stats("oracle-f32 elem lad sum vs f64", l32.sum(1) - l64.sum(1))
        stats("HIP        lad [N] vs f64", lg - l64.sum(1))
        stats("HIP        lad [N] vs oracle-f32", lg - l32.sum(1))


if __name__ == "__main__":
    main()
