"""In-kernel clock of fc_resnet_hidden: needs a probe build with -DFC_HID_STAMP
(FILE=fc_resnet_hidden EXTRA=-DFC_HID_STAMP tools/probe/build_fused_variants.sh 0;
 python tools/probe/hidden_clock.py --lib tools/probe/build/libfc_abl0.so [--log2n 20])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops, _hip  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
log2n = int(sys.argv[sys.argv.index("--log2n") + 1]) if "--log2n" in sys.argv else 20
n, d, hid, blocks = 1 << log2n, 64, 64, 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(n, d, device=dev)
ids = torch.arange(1, d, 2, dtype=torch.int32, device=dev)
from flowconductor_amd.nn import nets  # noqa: E402
net = nets.ResidualNet(32, 8, hidden_features=hid, num_blocks=blocks).eval().to(dev)
with torch.no_grad():
    for _ in range(200):
        h = net.hidden_hip(x, ids) if "--unpacked" not in sys.argv else ops.resnet_hidden(
            x, ids, ops.pack_resnet_hidden(net), 32, blocks)
torch.cuda.synchronize()
hh = h.view(-1, 16 * hid)[:, :4].cpu()            # first four floats of every 16-row block
ok = (hh[:, 1] > 0) & (hh[:, 1] < 1e6) & (hh[:, 1] == hh[:, 1].round()) & (hh[:, 0] > 1000)
cand = hh[ok]
per_block = cand[:, 0] / cand[:, 1]
print("prologue median %.0f cycles (max %.0f); loop median %.1f us -> clock %.3f GHz"
      % (cand[:, 2].median(), cand[:, 2].max(), cand[:, 3].median() / 100, float((cand[:, 0] / cand[:, 3]).median()) * 0.1))
print("stamped waves %d: cycles per 16-sample block per wave: median %.0f  min %.0f  max %.0f; blocks per wave median %.0f"
      % (cand.shape[0], per_block.median(), per_block.min(), per_block.max(), cand[:, 1].median()))
