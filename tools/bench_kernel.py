"""Times one bijector kernel in isolation on the BASELINE.json cfg-3 layer shape and prints achieved
algorithmic GB/s (HIP events on the launch stream).  Usage: python tools/bench_kernel.py [--lib path.so] [--log2n 20] [rq|rq_inv|rq_bwd|affine|fused|fused_inv|hidden|general|general_k<K>[_box][_streamed|_inv]|general_h256|fused_bwd|fused_bwd_k10|hidden_bwd|hidden_wide|hidden_wide128]"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowconductor_amd import ops  # noqa: E402


def main():
    argv = sys.argv[1:]
    log2n = 20
    if "--lib" in argv:      # an ablation build of the same ABI (tools/probe/build_fused_variants.sh)
        i = argv.index("--lib")
        from flowconductor_amd import _hip
        _hip.use_library(argv[i + 1])
        del argv[i:i + 2]
    if "--log2n" in argv:
        i = argv.index("--log2n")
        log2n = int(argv[i + 1])
        del argv[i:i + 2]
    which = argv[0] if argv else "rq"
    n = 1 << log2n
    d, d_t, k = 64, 32, 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(n, d, device=dev) * 1.5
    cols = torch.arange(0, d, 2, dtype=torch.int32, device=dev)
    if which in ("rq", "rq_inv"):
        p = 3 * k - 1
        params = torch.randn(n, d_t * p, device=dev)
        fn = lambda: ops.rq_spline(x, params, cols, num_bins=k, tails="linear", tail_bound=3.0,  # noqa: E731
                                   wh_divisor=8.0, inverse=which == "rq_inv")
        name = "fc_rq_spline"
    elif which in ("fused", "fused_inv"):
        p = 3 * k - 1
        h = torch.randn(n, 64, device=dev)
        w = torch.randn(d_t * p, 64, device=dev) * 0.2
        b = torch.randn(d_t * p, device=dev) * 0.1
        frag, bpad = ops.pack_final_layer(w, b)
        fn = lambda: ops.rq_spline_fused_linear(x, h, frag, bpad, cols, num_bins=k, tail_bound=3.0,  # noqa: E731
                                                wh_divisor=8.0, inverse=which == "fused_inv")
        name = "fc_rq_spline_fused_linear"
    elif which == "rq_bwd":
        p = 3 * k - 1
        params = torch.randn(n, d_t * p, device=dev)
        gy, gl = torch.randn(n, d, device=dev), torch.randn(n, device=dev)
        kw = dict(num_bins=k, tails="linear", tail_bound=3.0, wh_divisor=8.0)
        fn = lambda: ops.rq_spline_backward(x, params, cols, gy, gl, **kw)  # noqa: E731
        name = "fc_rq_spline_backward"
    elif which in ("general", "general_h256") or re.fullmatch(r"general_k\d+(_box)?(_streamed|_inv)?", which):
        # the general fused final-layer kernel: K = 8 / hidden 64 (the headline shape on the general structure),
        # K = 10 / hidden 64, K = 10 / hidden 256
        # (K = 10 / hidden 64 runs on the resident-weight instance, fc_rq_fused4; `_streamed`: the streamed-weight kernel)
        kk, hid = (8, 64) if which == "general" else (10, 256) if which == "general_h256" else (int(re.findall(r"\d+", which)[0]), 64)
        gtails = None if "_box" in which else "linear"
        p = 3 * kk + 1 if gtails is None else 3 * kk - 1
        h = torch.randn(n, hid, device=dev)
        w = torch.randn(d_t * p, hid, device=dev) * (1.0 / hid ** 0.5)
        b = torch.randn(d_t * p, device=dev) * 0.1
        if gtails is None:
            x = torch.rand(n, d, device=dev)
        packed = ops.pack_final_layer_general(w, b, kk, gtails, hid)
        fn = lambda: ops.rq_spline_fused_general(x, h, *packed, cols, num_bins=kk, tails=gtails, tail_bound=3.0,  # noqa: E731
                                                 wh_divisor=float(hid) ** 0.5, inverse=which.endswith("_inv"),
                                                 streamed_weights=which.endswith("_streamed"))
        name = "fc_rq_spline_fused_general"
    elif which in ("fused_bwd", "fused_bwd_k10", "fused_bwd_wide", "fused_bwd_wide_k10"):      # (the _wide names: round-4 probe logs)
        kk = 10 if which.endswith("_k10") else 8
        p = 3 * kk - 1
        h = torch.randn(n, 64, device=dev)
        w = torch.randn(d_t * p, 64, device=dev) * 0.125
        b = torch.randn(d_t * p, device=dev) * 0.1
        packed = ops.pack_final_layer_general(w, b, kk, "linear", 64)
        packed_t = ops.pack_final_layer_transposed(w, kk, "linear")
        gy, gl = torch.randn(n, d, device=dev), torch.randn(n, device=dev)
        fn = lambda: ops.rq_fused_linear_backward(x, h, gy, gl, packed, packed_t, cols, num_bins=kk, tails="linear",  # noqa: E731
                                                  tail_bound=3.0, wh_divisor=8.0)
        name = "fc_rq_fused_linear_backward"
    elif which == "hidden_bwd":
        from flowconductor_amd.nn import nets
        p = 0
        net = nets.ResidualNet(32, 8, hidden_features=64, num_blocks=2).to(dev)
        ids = torch.arange(1, d, 2, device=dev)
        gh = torch.randn(n, 64, device=dev)
        packed = net.hidden_backward_packed()
        fn = lambda: ops.resnet_hidden_backward(x, gh, ids, packed, 32, 2)  # noqa: E731
        name = "fc_resnet_hidden_backward"
    elif which in ("hidden_wide", "hidden_wide128"):
        from flowconductor_amd.nn import nets
        p = 0
        net = nets.ResidualNet(32, 8, hidden_features=256 if which == "hidden_wide" else 128, num_blocks=2).eval().to(dev)
        ids = torch.arange(1, d, 2, device=dev)
        fn = lambda: net.hidden_hip_wide(x, ids)  # noqa: E731
        name = "fc_resnet_hidden_wide"
    elif which in ("sos_inv", "sos_inv_bisect", "sos_fwd"):
        # sum-of-sigmoids (S = 30: 91 raw values per dim), D = 8 all transformed: numerical inverse / forward
        d, d_t, ns = 8, 8, 30
        p = 3 * ns + 1
        x = torch.randn(n, d, device=dev) * 2.0
        params = torch.randn(n, d * p, device=dev)
        iters = -50 if which == "sos_inv_bisect" else 50
        fn = lambda: ops.sum_of_sigmoids(x, params, ns, inverse=which != "sos_fwd", iterations=iters)  # noqa: E731
        name = "fc_sum_of_sigmoids"
    elif which == "hidden":
        from flowconductor_amd.nn import nets
        p = 0
        net = nets.ResidualNet(32, 8, hidden_features=64, num_blocks=2).eval().to(dev)
        ids = torch.arange(1, d, 2, device=dev)
        fn = lambda: net.hidden_hip(x, ids)  # noqa: E731
        name = "fc_resnet_hidden"
    else:
        p = 2
        params = torch.randn(n, d_t * p, device=dev)
        fn = lambda: ops.affine_coupling(x, params, cols)  # noqa: E731
        name = "fc_affine"
    alg = (4 * d_t * (p + 2) + 8) * n
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        timer = ops.KernelTimer(name)
        with timer:
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
    ms = sorted(timer.durations_ms())
    med = ms[len(ms) // 2]
    print("%s N=2^%d: median %.4f ms  min %.4f  max %.4f  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), "
          "actual bytes/alg = %.3f" % (which, n.bit_length() - 1, med, ms[0], ms[-1], alg / med / 1e6,
                                       alg / med / 1e6 / 80.0, (4 * (d_t * p + 2 * d + 1)) / (4 * d_t * (p + 2) + 8)))


if __name__ == "__main__":
    main()
