"""Does splitting the 2^20-row batch over two HIP streams (each half through all 32 layers) fill the kernel-boundary gaps
and drain tails of the headline flow?  python tools/probe/bench_two_streams.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

flow = bench.build_flow().to("cuda").eval()
n = 1 << 20
x = torch.randn(n, 64, device="cuda")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one():
    return flow.log_prob(x)


def two(parts=2):
    outs = []
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams[:parts]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(flow.log_prob(x[i * n // parts:(i + 1) * n // parts]))
    for s in streams[:parts]:
        cur.wait_stream(s)
    return torch.cat(outs)


with torch.no_grad():
    for fn in (one, two, one, two):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = fn()
        torch.cuda.synchronize()
        print(fn.__name__, "%.3f ms per 2^20 rows" % ((time.perf_counter() - t0) / 10 * 1e3))
    print("max |difference|", float((one() - two()).abs().max()))
