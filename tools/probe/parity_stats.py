"""rms / p99 / p99.99 / max of the cfg-3 flow's error against float64, GPU vs the float32 CPU oracle, default and
trained-like weights (the statistics tests/test_gpu_parity_gate.py asserts).  python tools/probe/parity_stats.py [rows]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import bench  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
flow_cpu = bench.build_flow()
for name, f in (("default_init", flow_cpu), ("trained_like", bench.trained_like(flow_cpu))):
    x = torch.randn(rows, 64, generator=torch.Generator().manual_seed(7))
    st = f._transform
    with torch.no_grad():
        z32, l32 = O.transform_apply(st, x.clone())
        z64, l64 = O.transform_apply(copy.deepcopy(st).double(), x.double())
        z, lad = copy.deepcopy(st).to(dev).eval()(x.to(dev))

    def rel(a, b):
        return ((a.double().cpu() - b.double()).abs() / b.double().abs().clamp_min(1.0)).flatten()

    for what, g, r in (("samples", rel(z, z64), rel(z32, z64)), ("logabsdet", rel(lad, l64), rel(l32, l64))):
        def stats(e):
            return [float(e.pow(2).mean().sqrt()), float(e.quantile(0.99)) if e.numel() < 16e6 else 0.0,
                    float(e.kthvalue(max(1, int(e.numel() * 0.9999))).values), float(e.max())]
        sg, sr = stats(g), stats(r)
        print("%s %s rows %d: GPU rms %.3g p99 %.3g p99.99 %.3g max %.3g | ref32 rms %.3g p99 %.3g p99.99 %.3g max %.3g | ratios %s"
              % (name, what, rows, *sg, *sr, ["%.2f" % (a / b) for a, b in zip(sg, sr)]))
