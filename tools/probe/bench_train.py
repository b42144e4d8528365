"""Training step of the cfg-3 flow (32 x RQ coupling, D = 64, K = 8): -log_prob.mean().backward() + Adam step through
the HIP forward / backward bijector kernels with the conditioners on PyTorch autograd.
python tools/probe/bench_train.py [log2 rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import distributions, flows, ops, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 17)
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

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    timers = [ops.KernelTimer(k) for k in ("fc_rq_spline", "fc_rq_spline_backward")]
    t0 = time.perf_counter()
    reps = 3
    with timers[0], timers[1]:
        for _ in range(reps):
            loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    f = timers[0].durations_ms()
    b = timers[1].durations_ms()
    print(f"train step N={n}: {ms:.1f} ms ({n / ms / 1e3:.2f} M samples/s), loss {float(loss):.3f}; "
          f"fc_rq_spline {sum(f) / reps:.1f} ms/step in {len(f) // reps} launches, "
          f"fc_rq_spline_backward {sum(b) / reps:.1f} ms/step in {len(b) // reps} launches; "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
