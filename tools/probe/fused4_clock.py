"""In-kernel clock of the K = 10 resident-weight fused kernel: needs a probe build with FC_F4_STAMP
(tools/probe/build_f4_variants.sh stamp "-DFC_F4_STAMP" X=1; python tools/probe/fused4_clock.py --lib tools/probe/build/libf4_stamp.so)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])

n, d, d_t, k = 1 << 20, 64, 32, 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(n, d, device=dev) * 1.5
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
h = torch.randn(n, 64, device=dev)
w = torch.randn(d_t * (3 * k - 1), 64, device=dev) * 0.125
b = torch.randn(d_t * (3 * k - 1), device=dev) * 0.1
packed = ops.pack_final_layer_general(w, b, k, "linear", 64)
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < 2.5:   # >= 2 s of back-to-back launches before reading the stamps
        for _ in range(50):
            y, lad = ops.rq_spline_fused_general(x, h, *packed, cols, num_bins=k, tails="linear", tail_bound=3.0, wh_divisor=8.0)
        torch.cuda.synchronize()
cus = torch.cuda.get_device_properties(0).multi_processor_count
R = 32        # rows per tile
tiles = n / R / cus
yy = y.view(-1)[:cus * R * d].view(cus, R * d)[:, :52].double().cpu()
ph = (yy[:, 4:52].median(dim=0).values / tiles).view(8, 6)
print("phase cycles per %d-row tile [loop overhead, fetch + steps 0..NB-2, park, barrier, write-out of the previous tile, last step] per wave:" % R)
print("")
for wv in range(8):
    print("  wave", wv, [int(v) for v in ph[wv]], "sum", int(ph[wv].sum()))
cyc, rt = yy[:, 0], yy[:, 1]
ghz = cyc / rt * 0.1
print("workgroups %d  tiles/wg %.0f  cycles median %.0f  realtime median %.1f us  clock median %.3f GHz  cycles per tile %.0f"
      % (cus, tiles, cyc.median(), rt.median() / 100.0, ghz.median(), cyc.median() / tiles))
