"""Tanh forward (nonlinearities.py:40-43): float32 reference sequence and fc_elementwise against float64, inputs 2 N(0, 1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import transforms as T
torch.manual_seed(0)
x = torch.randn(4096, 64) * 2.0
t = T.Tanh()
y32 = torch.tanh(x); lad32 = torch.log(1 - y32 ** 2).sum(-1)
x64 = x.double(); y64 = torch.tanh(x64); lad64 = (2 * (torch.log(torch.tensor(2.0, dtype=torch.float64)) - x64.abs() - torch.log1p(torch.exp(-2 * x64.abs())))).sum(-1)
yg, ladg = t.to("cuda")(x.cuda())
ladg = ladg.cpu()
ok = torch.isfinite(lad32) & torch.isfinite(ladg)
print("rows with -inf: reference %d, kernel %d (y rounded to 1 in float32)" % (int((~torch.isfinite(lad32)).sum()), int((~torch.isfinite(ladg)).sum())))
print("reference f32 vs f64: y %.2e lad %.2e" % (float((y32.double() - y64).abs().max()), float((lad32.double() - lad64)[ok].abs().max())))
print("kernel      vs f64: y %.2e lad %.2e" % (float((yg.cpu().double() - y64).abs().max()), float((ladg.double() - lad64)[ok].abs().max())))
