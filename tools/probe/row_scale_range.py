"""What the row scale of the split-f16 products (fc_split.h, kSplitTopExp) buys a MASKED conditioner: a MAF layer whose
last column is 2^r times larger than the others, error of the first D - 1 outputs against float64.
python tools/probe/row_scale_range.py [--lib probe.so built with -DFC_SPLIT_TOP_EXP=10]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import _hip, transforms as T  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
torch.manual_seed(11)
d = 12
t = T.MaskedAffineAutoregressiveTransform(d, 64, num_blocks=2).eval()
with torch.no_grad():
    for p in t.parameters():
        p.mul_(1.5)
tg = copy.deepcopy(t).to("cuda")
base = torch.randn(2048, d)
for r in (0, 8, 12, 14, 16, 18, 20, 22, 24):
    x = base.clone()
    x[:, d - 1] *= 2.0 ** r
    with torch.no_grad():
        ref_y, _ = O.transform_apply(copy.deepcopy(t).double(), x.double())
        f32_y, _ = O.transform_apply(t, x)
        y, _ = tg(x.cuda())
    head = slice(0, d - 1)
    e = float((y[:, head].cpu().double() - ref_y[:, head]).abs().max())
    f = float((f32_y[:, head].double() - ref_y[:, head]).abs().max())
    print("last column x 2^%-2d: kernel %.2e   float32 reference %.2e" % (r, e, f))
