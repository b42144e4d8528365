"""cfg-3 log_prob step with and without the per-launch HIP-event pairs bench.py records for `roofline`, and replayed
as one HIP graph: how much of the step is launch gaps.  python tools/probe/bench_step_overheads.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from flowconductor_amd import ops  # noqa: E402
from flowconductor_amd.utils.graphs import GraphedCall  # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    flow = bench.build_flow().to(dev)
    x = torch.randn(1 << 20, bench.FEATURES, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))

    def step():
        with torch.no_grad():
            return flow.log_prob(x)

    def step_timers():
        with ops.KernelTimer("fc_rq_spline_fused_linear"), ops.KernelTimer("fc_resnet_hidden"):
            return step()

    graphed = GraphedCall(flow.log_prob, x)
    res = {}
    for _ in range(2):
        for name, fn in (("plain", step), ("with event pairs", step_timers), ("HIP graph replay", lambda: graphed(x))):
            res[name] = min(res.get(name, 1e9), timed(fn))
    print("  ".join(f"{k}: {v:.3f} ms" for k, v in res.items()))


if __name__ == "__main__":
    main()
