"""Times the round-2 HIP backward kernels of the non-RQ bijectors (forward + backward of one op through autograd,
HIP events around the backward kernel launches via ops.KernelTimer).  python tools/probe/bench_backward_kernels.py [log2 rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(name, timer_name, make, algorithmic_bytes):
    args = make()
    for _ in range(3):
        for a in args:
            a.grad = None
        y, lad = fn[name](*args)
        (y.sum() + lad.sum()).backward()
    torch.cuda.synchronize()
    timer = ops.KernelTimer(timer_name)
    with timer:
        for _ in range(10):
            for a in args:
                a.grad = None
            y, lad = fn[name](*args)
            (y.sum() + lad.sum()).backward()
    torch.cuda.synchronize()
    ms = sorted(timer.durations_ms())
    per_call = len(ms) // 10
    med = sorted(sum(ms[i::1][:0]) for i in range(1)) if False else None
    total = sum(timer.durations_ms()) / 10
    print("%-34s N=2^%d: %.3f ms per backward (%d launch(es)), %.0f GB/s of algorithmic bytes"
          % (name, n.bit_length() - 1, total, per_call, algorithmic_bytes / (total * 1e-3) / 1e9))


def leaf(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).requires_grad_(True)


d, s_ = 8, 30
fn = {
    "sum_of_sigmoids (D=8, S=30)": lambda x, p: ops.sum_of_sigmoids_autograd(x, p, s_),
    "planar (D=128)": lambda x, w, u, b: ops.planar_autograd(x, w, u, b),
    "householder (D=128, K=32)": lambda x, q: ops.householder_autograd(x, q),
    "sylvester (D=128, M=32)": lambda x, q, r1, r2, b: ops.sylvester_autograd(x, q, r1, r2, b),
    "quadratic spline (d_t=16, K=8)": lambda x, p: ops.piecewise_spline_autograd(
        x, p, None, kind=ops.SPLINE_QUADRATIC, num_bins=8, tails="linear", tail_bound=3.0),
    "cubic spline (d_t=16, K=8)": lambda x, p: ops.piecewise_spline_autograd(
        x, p, None, kind=ops.SPLINE_CUBIC, num_bins=8, tails="linear", tail_bound=3.0),
}
run("sum_of_sigmoids (D=8, S=30)", "fc_sum_of_sigmoids_backward", lambda: (leaf(n, d, scale=3.0), leaf(n, d * (3 * s_ + 1))),
    n * d * (8 * (3 * s_ + 1) + 12))
run("planar (D=128)", "fc_planar_backward", lambda: (leaf(n, 128), leaf(1, 128, scale=0.1), leaf(1, 128, scale=0.1), leaf(1)),
    n * 128 * 12)
run("householder (D=128, K=32)", "fc_householder_backward", lambda: (leaf(n, 128), leaf(32, 128)), n * 128 * 12)
r = lambda: torch.triu(torch.randn(128, 128, device=dev) / 12).requires_grad_(True)  # noqa: E731
run("sylvester (D=128, M=32)", "fc_householder_backward", lambda: (leaf(n, 128), leaf(32, 128), r(), r(), leaf(128, scale=0.1)),
    n * 128 * 12 * 2)
run("quadratic spline (d_t=16, K=8)", "fc_piecewise_spline_backward", lambda: (leaf(n, 16), leaf(n, 16 * 15)),
    n * 16 * (8 * 15 + 12))
run("cubic spline (d_t=16, K=8)", "fc_piecewise_spline_backward", lambda: (leaf(n, 16), leaf(n, 16 * 18)),
    n * 16 * (8 * 18 + 12))
