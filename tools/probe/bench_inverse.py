"""cfg-3 flow, sampling direction: Flow.sample-style inverse pass of 2^20 noise rows through the 32 layers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

flow = bench.build_flow().to("cuda")
z = torch.randn(1 << 20, 64, device="cuda")
with torch.no_grad():
    for _ in range(2):
        x, lad = flow._transform.inverse(z)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        x, lad = flow._transform.inverse(z)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    z2, lad2 = flow._transform(x)
print("inverse pass: %.2f ms per 2^20 samples -> %.1f M samples/s; round trip max |z - f(f^-1(z))| = %.2e, max |lad + lad_inv| = %.2e"
      % (dt * 1e3, (1 << 20) / dt / 1e6, float((z2 - z).abs().max()), float((lad + lad2).abs().max())))
