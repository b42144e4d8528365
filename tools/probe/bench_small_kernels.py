"""Streaming kernels at 2^20 x 64: achieved GB/s (read + write of the [N, D] tensor) -- looks for scalar-load leftovers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops  # noqa: E402

n, d = 1 << 20, 64
dev = "cuda"
x = torch.randn(n, d, device=dev)
perm = torch.randperm(d, device=dev)
scale, shift = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
params = torch.randn(n, d, device=dev)
cases = {
    "fc_permute": (lambda: ops.permute(x, perm), 2 * x.numel() * 4),
    "fc_pointwise_affine": (lambda: ops.pointwise_affine(x, scale, shift), 2 * x.numel() * 4),
    "fc_elementwise": (lambda: ops.elementwise(x, ops.EW_TANH), 2 * x.numel() * 4),
    "fc_affine": (lambda: ops.affine_coupling(x, params, cols), (2 * x.numel() + params.numel()) * 4),
    "fc_householder": (lambda: ops.householder(x, torch.randn(8, d, device=dev)), 2 * x.numel() * 4),
}
for name, (fn, nbytes) in cases.items():
    try:
        with torch.no_grad():
            for _ in range(3):
                fn()
            with ops.KernelTimer(name) as t:
                for _ in range(10):
                    fn()
            torch.cuda.synchronize()
        ms = sorted(t.durations_ms())
        med = ms[len(ms) // 2]
        print("%-22s %.3f ms  %.0f GB/s" % (name, med, nbytes / med / 1e6))
    except Exception as e:  # noqa: BLE001
        print(name, "skipped:", type(e).__name__, e)
