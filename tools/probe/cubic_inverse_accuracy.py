"""CPU probe behind fc_splines.hip's cubic inverse: the kernel's stable evaluation (larger cube + p q = -delta_1, stable
quadratic fallback, two guarded Newton steps) restated in torch float32, against the reference's op sequence (the oracle) in
float32, both measured against the reference's op sequence in float64.  Runs anywhere:
    python tools/probe/cubic_inverse_accuracy.py"""
import sys, torch, math
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
from oracle import torch_oracle as O
import torch.nn.functional as F
torch.manual_seed(0)
def coeffs(uw, uh, dl, dr, y, K, thresh=1e-3):
    def normalise(u, floor):
        v = F.softmax(u, dim=-1); v = floor + (1 - floor * K) * v
        cum = torch.cumsum(v, dim=-1); cum[..., -1] = 1
        return v, F.pad(cum, pad=(1, 0), value=0.0)
    w, cw = normalise(uw, 1e-3); h, ch = normalise(uh, 1e-3)
    s = h / w
    m1 = torch.min(s[..., :-1].abs(), s[..., 1:].abs())
    m2 = 0.5 * (w[..., 1:] * s[..., :-1] + w[..., :-1] * s[..., 1:]) / (w[..., :-1] + w[..., 1:])
    dL = torch.sigmoid(dl) * 3 * s[..., 0][..., None]; dR = torch.sigmoid(dr) * 3 * s[..., -1][..., None]
    de = torch.cat([dL, torch.min(m1, m2) * (torch.sign(s[..., :-1]) + torch.sign(s[..., 1:])), dR], -1)
    a = (de[..., :-1] + de[..., 1:] - 2 * s) / w.pow(2); b = (3 * s - 2 * de[..., :-1] - de[..., 1:]) / w; c = de[..., :-1]; d = ch[..., :-1]
    k = O.searchsorted(ch.clone(), y)[..., None]
    g = lambda t: t.gather(-1, k)[..., 0]
    return g(a), g(b), g(c), g(d), g(cw), cw.gather(-1, k + 1)[..., 0]
def stable_inverse(ca, cb, cc, dco, left, right, y, thresh=1e-3, eps=1e-5, newton=2):
    b_ = (cb / ca) / 3.0; c_ = (cc / ca) / 3.0; d_ = (dco - y) / ca
    d1 = -b_ * b_ + c_; d2 = -c_ * b_ + d_; d3 = b_ * d_ - c_ * c_
    disc = 4.0 * d1 * d3 - d2 * d2; dep1 = -2.0 * b_ * d1 + d2
    # one root: the larger cube by the non-cancelling sum, the other from p q = -delta_1
    sq = torch.sqrt((-disc).clamp_min(0))
    A = -dep1
    big = (A + torch.where(A >= 0, sq, -sq)) / 2.0
    p = torch.sign(big) * torch.exp(torch.log(big.abs()) / 3.0)
    q = torch.where(p != 0, -d1 / p, torch.zeros_like(p))
    one = (p + q) - b_ + left
    th = torch.atan2(torch.sqrt(disc.clamp_min(0)), -dep1) / 3.0
    c1, s1 = torch.cos(th), torch.sin(th); k3 = 0.5 * math.sqrt(3)
    sc = 2 * torch.sqrt((-d1).clamp_min(0)); sh = -b_ + left
    r1 = c1 * sc + sh; r2 = (-0.5 * c1 - k3 * s1) * sc + sh; r3 = (-0.5 * c1 + k3 * s1) * sc + sh
    m = lambda r: ((left - eps) < r) & (r < (right + eps))
    three = torch.where(m(r1), r1, torch.where(m(r2), r2, torch.where(m(r3), r3, r1)))
    out = torch.where(disc >= 0, three, one)
    for _ in range(newton):
        t = out - left
        fv = ((ca * t + cb) * t + cc) * t + (dco - y)
        fp = (3 * ca * t + 2 * cb) * t + cc
        new = out - fv / fp
        ok = (fp > 0) & (new > left - eps) & (new < right + eps) & torch.isfinite(new)
        out = torch.where(ok, new, out)
    quad = ca.abs() < thresh
    a2, b2, c2 = cb, cc, dco - y
    alpha = 2 * c2 / (-b2 - torch.sqrt(b2 * b2 - 4 * a2 * c2))
    out = torch.where(quad, alpha + left, out)
    return out
def reference_inverse(uw, uh, dl, dr, y):
    return O.cubic_spline(y, uw, uh, dl, dr, inverse=True)[0]
for scale in (0.5, 1.5, 3.0, 6.0):
  for K in (2, 4, 8, 12):
    n = 200000
    uw = torch.randn(n, K) * scale; uh = torch.randn(n, K) * scale; dl = torch.randn(n, 1) * scale; dr = torch.randn(n, 1) * scale
    y = torch.rand(n) * 0.96 + 0.02
    ref32 = reference_inverse(uw, uh, dl, dr, y)
    ref64 = reference_inverse(uw.double(), uh.double(), dl.double(), dr.double(), y.double())
    ca, cb, cc, dco, left, right = coeffs(uw, uh, dl, dr, y, K)
    new32 = stable_inverse(ca, cb, cc, dco, left, right, y)
    ok = torch.isfinite(ref64)
    e_ref = (ref32.double() - ref64).abs()[ok]; e_new = (new32.double() - ref64).abs()[ok]
    e_ref = torch.nan_to_num(e_ref, nan=1.0); e_new = torch.nan_to_num(e_new, nan=1.0)
    # forward consistency in f64
    print("scale %.1f K %2d  ref32: max %.2e p99.9 %.2e  | stable32: max %.2e p99.9 %.2e | nan ref32 %d new %d ref64 %d | worst new/ref excess %.2e"
          % (scale, K, e_ref.max(), e_ref.quantile(0.999), e_new.max(), e_new.quantile(0.999), int(torch.isnan(ref32).sum()), int(torch.isnan(new32).sum()), int((~ok).sum()),
             (e_new - 4 * e_ref).max()))
