"""In-kernel clock of the fused kernel: needs a probe build with FC_ABL & 16
(tools/probe/build_fused_variants.sh 16; python tools/probe/fused_clock.py --lib tools/probe/build/libfc_abl16.so)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])

n, d, d_t, k = 1 << 20, 64, 32, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(n, d, device=dev) * 1.5
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
h = torch.randn(n, 64, device=dev)
w = torch.randn(d_t * 23, 64, device=dev) * 0.2
b = torch.randn(d_t * 23, device=dev) * 0.1
wp, bp = ops.pack_final_layer(w, b)
total = torch.zeros(n, device=dev) if "--accumulate" in sys.argv else None   # the flow's running logabsdet total
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < 2.5:   # >= 2 s of back-to-back launches before reading the stamps
        for _ in range(50):
            y, lad = ops.rq_spline_fused_linear(x, h, wp, bp, cols, num_bins=k, tail_bound=3.0, wh_divisor=8.0,
                                                logabsdet_accum=total)
        torch.cuda.synchronize()
cus = torch.cuda.get_device_properties(0).multi_processor_count
yy = y.view(-1, 64 * d)[:cus, :60].double().cpu()
ph = (yy[:, 12:60].median(dim=0).values / (n // 64 / cus)).view(8, 6)
print('phase cycles per 64-row tile [loop overhead + fetch, steps 0-2, park, barrier, last step, write-out of the previous tile] per wave:')
for w in range(8):
    print('  wave', w, [int(v) for v in ph[w]])
print('barrier wait per wave (cycles per tile, median over workgroups):', [round(float(v), 0) for v in (yy[:, 4:12].median(dim=0).values / (n // 64 / cus))])
pro, entry = yy[:, 2], yy[:, 3]
print("prologue median %.1f us max %.1f us; entry-time spread over workgroups %.1f us; loop end spread %.1f us"
      % (pro.median() / 100, pro.max() / 100, (entry.max() - entry.min()) / 100,
         ((entry + pro + yy[:, 1]).max() - (entry + pro + yy[:, 1]).min()) / 100))
fin = entry + pro + yy[:, 1]
print("workgroup finish times (10 ns ticks after the first entry): mean %.0f  median %.0f  max %.0f -> the last workgroup "
      "ends %.1f us after the average one (%.1f %% of the kernel)"
      % (fin.mean() - entry.min(), fin.median() - entry.min(), fin.max() - entry.min(),
         (fin.max() - fin.mean()) / 100, 100 * (fin.max() - fin.mean()) / (fin.max() - entry.min())))
cyc, rt = yy[:, 0], yy[:, 1]
ghz = (cyc / rt * 0.1)
tiles = n // 64 / cus
print("workgroups %d  tiles/wg %.0f  cycles median %.0f  realtime median %.1f us  clock median %.3f GHz (min %.3f max %.3f)  cycles per 64-row tile %.0f (%.0f per 32 rows)"
      % (cus, tiles, cyc.median(), rt.median() / 100.0, ghz.median(), ghz.min(), ghz.max(), cyc.median() / tiles, cyc.median() / tiles / 2))
