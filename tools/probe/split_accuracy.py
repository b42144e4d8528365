"""CPU simulation of the split-precision products used by the fused kernel (fc_rq_fused3.hip): error of
W h against float64 for an f32 GEMM, the 3-piece bf16 split (6 terms) and the scaled 2-piece f16 split
(3 terms).  Pure torch-CPU arithmetic; products of 16-bit pieces are exact in f32/f64."""
import torch

torch.manual_seed(0)
n = 8192
h = torch.relu(torch.randn(n, 64)) * 2 + torch.randn(n, 64) * 0.3
h[::7] *= 30
h[::11] *= 1e-3
w = torch.randn(768, 64) * 0.2
w[::5] *= 1e-3


def d(a, b):
    return a.double() @ b.double().T


def split(x, dtype, pieces):
    out, r = [], x
    for _ in range(pieces):
        p = r.to(dtype).float()
        out.append(p)
        r = r - p
    return out


ex = d(h, w)
sab = d(h.abs(), w.abs())
rows = [("f32 GEMM (torch CPU)", (h @ w.T).double())]
hh, hm, hl = split(h, torch.bfloat16, 3)
wh, wm, wl = split(w, torch.bfloat16, 3)
rows.append(("bf16 x3, 6 terms", d(hh, wl) + d(hl, wh) + d(hm, wm) + d(hh, wm) + d(hm, wh) + d(hh, wh)))
rows.append(("bf16 x3, 3 terms", d(hh, wm) + d(hm, wh) + d(hh, wh)))
S = torch.floor(10 - torch.log2(w.abs().max()))
T = torch.floor(10 - torch.log2(h.abs().amax(dim=1, keepdim=True)))
hh, hl = split(h * 2 ** T, torch.float16, 2)
wh, wl = split(w * 2 ** S, torch.float16, 2)
un = (2.0 ** (-S)) * (2.0 ** (-T)).double()
rows.append(("f16 x2 scaled, 3 terms", (d(hh, wl) + d(hl, wh) + d(hh, wh)) * un))
rows.append(("f16 x2 scaled, 3 terms, each term rounded to f32",
             ((d(hh, wl).float() + d(hl, wh).float()) + d(hh, wh).float()).double() * un))
for name, v in rows:
    e = (v - ex).abs() / sab
    print("%-52s max %.2e  rms %.2e   (relative to sum |W||h|)" % (name, e.max(), (e ** 2).mean().sqrt()))
