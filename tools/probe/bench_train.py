"""Training step of the cfg-3 flow (32 x RQ coupling, D = 64, K = 8): -log_prob.mean().backward() + Adam step.

Two paths, same flow, same data, interleaved in one process:
  fused    forward in fc_resnet_hidden + fc_rq_spline_fused_general, backward in fc_rq_fused_linear_backward (+ the hidden
           stack's backward): no [N, 736] parameter / gradient tensor in either direction
  unfused  conditioners on PyTorch autograd (library GEMMs), bijector forward / backward in fc_rq_spline(_backward)
python tools/probe/bench_train.py [log2 rows] [--json] [--fused-only]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import _hip, distributions, flows, ops, options, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

KERNELS = ("fc_rq_spline", "fc_rq_spline_backward", "fc_rq_spline_fused_general", "fc_rq_fused_linear_backward",
           "fc_resnet_hidden", "fc_resnet_hidden_backward")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = 1 << (int(args[0]) if args else 17)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2),
        num_bins=8, tails="linear", tail_bound=3.0) for i in range(32)]
    flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([64])).to(dev).train()
    opt = torch.optim.Adam(flow.parameters(), lr=1e-4)
    x = torch.randn(n, 64, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()
        return loss

    res = {}
    modes = ("fused", "fused") if "--fused-only" in sys.argv else ("fused", "unfused", "fused", "unfused")
    for mode in modes:
        with options.override(fused_training=(mode == "fused")):
            torch.cuda.reset_peak_memory_stats()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            timers = [ops.KernelTimer(k) for k in KERNELS]
            for t in timers:
                t.__enter__()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                loss = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            for t in reversed(timers):
                t.__exit__(None, None, None)
        rec = {"ms_per_step": ms, "samples_per_s": n / ms * 1e3, "loss": float(loss),
               "peak_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
               "kernels_ms_per_step": {t.name: [round(sum(t.durations_ms()) / reps, 3), len(t.pairs) // reps]
                                       for t in timers if t.pairs}}
        if mode not in res or ms < res[mode]["ms_per_step"]:
            res[mode] = rec
    res["rows"] = n
    if "unfused" not in res:
        res["library"] = _hip.library_info()
        print(json.dumps(res))
        return
    res["speedup"] = res["unfused"]["ms_per_step"] / res["fused"]["ms_per_step"]
    if "--json" in sys.argv:
        res["library"] = _hip.library_info()
        print(json.dumps(res))
        return
    for mode in ("fused", "unfused"):
        r = res[mode]
        print(f"train step N={n} [{mode}]: {r['ms_per_step']:.1f} ms ({r['samples_per_s'] / 1e6:.2f} M samples/s), loss "
              f"{r['loss']:.3f}, peak {r['peak_GiB']:.1f} GiB; kernels (ms/step, launches): {r['kernels_ms_per_step']}")
    print(f"fused / unfused: x{res['speedup']:.2f}")


if __name__ == "__main__":
    main()
