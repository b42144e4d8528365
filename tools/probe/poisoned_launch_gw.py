"""Is the gW fault of the old fc_rq_fused_linear_backward a read of INHERITED state?  Before the role-1 launch a probe kernel
(tools/probe/poison_state.hip) leaves a bit pattern in VGPRs v[lo, lo + len) of every wave slot, in the whole LDS and in the
scratch memory; a kernel that reads any of it before writing it then fails on EVERY launch, not on every second cold one.

    python tools/probe/poisoned_launch_gw.py [--lib tools/probe/build/libfc_oldbwd_v0.so] [--k 8] [--scan]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from flowconductor_amd import _hip, ops  # noqa: E402

if "--lib" in sys.argv:
    _hip.use_library(sys.argv[sys.argv.index("--lib") + 1])
from oracle import torch_oracle as O  # noqa: E402

poison = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "build", "libpoison.so"))
poison.fc_probe_poison.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 8
tails, d, d_t, n, hidden = "linear", 64, 32, 4096, 64
p = 3 * k - 1
torch.manual_seed(0)
x = torch.randn(n, d) * 1.5
h = torch.relu(torch.randn(n, hidden)) * 1.5 + torch.randn(n, hidden) * 0.2
w = torch.randn(d_t * p, hidden) * (1.0 / hidden ** 0.5)
b = torch.randn(d_t * p) * 0.3
cols = torch.arange(0, 2 * d_t, 2, dtype=torch.int32)[:d_t]
gy, gl = torch.randn(n, d), torch.randn(n)
kw = dict(wh_divisor=float(hidden) ** 0.5)
packed = ops.pack_final_layer_general(w.to(dev), b.to(dev), k, tails, 64)
packed_t = ops.pack_final_layer_transposed(w.to(dev), k, tails)
x64, h64, w64, b64 = (t.double().requires_grad_(True) for t in (x, h, w, b))
rows = (h64 @ w64.T + b64).view(n, d_t, p)
out, lad_e = O.rq_from_rows(x64[:, cols.long()], rows.clone(), k, tails, 3.0, False, **kw)
y64 = x64.clone().index_copy(1, cols.long(), out)
loss = (y64 * gy.double()).sum() + (lad_e.sum(dim=1) * gl.double()).sum()
gw_ref, gh_ref, gx_ref = torch.autograd.grad(loss, [w64, h64, x64])
scale = float(gw_ref.abs().max())
sink = torch.zeros(4, dtype=torch.int32, device=dev)
args_dev = [t.to(dev) for t in (x, h, gy, gl)]
state = {"cfg": None, "role": 1}
real_call = ops._call


def patched(name, fn, device, *a):
    if name == "fc_rq_fused_linear_backward" and a[0] == state["role"] and state["cfg"] is not None:
        pat, lo, ln, lds, scr = state["cfg"]
        rc = poison.fc_probe_poison(pat, lo, ln, lds, scr, ctypes.c_void_p(sink.data_ptr()), _hip.stream_ptr(device))
        assert rc == 0, rc
    return real_call(name, fn, device, *a)


ops._call = patched


def launch(cfg):
    state["cfg"] = cfg
    o = ops.rq_fused_linear_backward(*args_dev, packed, packed_t, cols.to(dev), num_bins=k, tails=tails, tail_bound=3.0,
                                     merged=False, **kw)
    torch.cuda.synchronize()
    gw = o[2].cpu().double()
    e = (gw - gw_ref).abs().amax(dim=1)
    bad = torch.nonzero(~(e <= 1e-3 * scale)).flatten().tolist()      # NaN counts as bad
    egh = float((o[1].cpu().double() - gh_ref).abs().max()) / float(gh_ref.abs().max())
    egx = float((o[0].cpu().double() - gx_ref).abs().max()) / float(gx_ref.abs().max())
    return bad, bool(torch.isnan(gw).any()), egh, egx


def show(tag, cfg, reps=3):
    res = [launch(cfg) for _ in range(reps)]
    print("%-44s %s" % (tag, " | ".join("%d bad rows%s dims %s gh %.1e gx %.1e" % (
        len(bad), " (NaN)" if nan else "", sorted({r // p for r in bad})[:6], egh, egx) for bad, nan, egh, egx in res)))
    return any(bad for bad, _, _, _ in res)


NAN, ONE = 0x7fc00000, 0x3f800000
for role in ((0, 1) if "--both-roles" in sys.argv else (1,)):
    state["role"] = role
    print("== state poisoned before the role-%d launch" % role)
    show("no poison (warm state)", None)
    for pat, pname in ((NAN, "NaN"), (ONE, "1.0f")):
        any_v = show("VGPRs v0..v255 <- %s" % pname, (pat, 0, 256, 0, 0))
        show("LDS <- %s" % pname, (pat, 0, 0, 1, 0))
        show("scratch <- %s" % pname, (pat, 0, 0, 0, 1))
        show("VGPRs + LDS + scratch <- %s" % pname, (pat, 0, 256, 1, 1))
        if any_v and "--scan" in sys.argv:
            hits = [r for r in range(256) if launch((pat, r, 1, 0, 0))[0]]
            print("   single registers whose inherited value reaches the result (%s): %s" % (pname, hits))
print("done")
