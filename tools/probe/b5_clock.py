"""In-kernel clock of the one-launch backward (fc_rq_fused_backward512.h): needs a probe build with -DFC_B5_STAMP
(tools/probe/build_b5_variants.sh stamp "-DFC_B5_STAMP"; python tools/probe/b5_clock.py --lib tools/probe/build/libb5_stamp.so)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
log2n = int(sys.argv[sys.argv.index("--log2n") + 1]) if "--log2n" in sys.argv else 19
n, d, d_t, k = 1 << log2n, 64, 32, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(n, d, device=dev) * 1.5
cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
p = 3 * k - 1
h = torch.randn(n, 64, device=dev)
w = torch.randn(d_t * p, 64, device=dev) * 0.125
b = torch.randn(d_t * p, device=dev) * 0.1
packed = ops.pack_final_layer_general(w, b, k, "linear", 64)
packed_t = ops.pack_final_layer_transposed(w, k, "linear")
gy, gl = torch.randn(n, d, device=dev), torch.randn(n, device=dev)
for _ in range(5):
    gx, gh, gw, gb = ops.rq_fused_linear_backward(x, h, gy, gl, packed, packed_t, cols, num_bins=k, tails="linear",
                                                  tail_bound=3.0, wh_divisor=8.0)
torch.cuda.synchronize()
cus = torch.cuda.get_device_properties(0).multi_processor_count
st = gh[:cus].view(cus, 4, 16).double().cpu()
tiles = n // 32 / cus * 2      # tile-sweeps per workgroup
names = ["loop", "recompute", "fetch+wt issue", "spline b0", "spline b1", "gh", "gW pass 1+2", "gW pass 3", "wring+park",
         "barrier 1", "write-out", "barrier 2"]
med = st.median(dim=0).values
print("cycles per tile-sweep and wave (median over %d workgroups, %.0f tile-sweeps each):" % (cus, tiles))
for i, nm in enumerate(names):
    print("  %-16s" % nm, [int(med[wv, i] / tiles) for wv in range(4)])
print("  %-16s" % "sum", [int(med[wv, :12].sum() / tiles) for wv in range(4)])
arr = st[:, :, 14]
print("arrival at barrier 1 of the 11th tile, cycles after kernel entry, waves 0-3 of the first workgroups:")
for wg in range(4):
    print("   wg", wg, [int(v) for v in arr[wg]], " spread", int(arr[wg].max() - arr[wg].min()))
print("   median spread over workgroups %.0f" % (arr.max(dim=1).values - arr.min(dim=1).values).median())
tot, rt = st[:, 0, 12], st[:, 0, 13]
print("kernel: cycles median %.0f, %.1f us, clock %.3f GHz; per tile-sweep %.0f cycles"
      % (tot.median(), rt.median() / 100, (tot / rt * 0.1).median(), tot.median() / tiles))
