"""BASELINE.md section 3 cross-check, build container only (needs /root/reference): the CPU restatement that
bench.py times as `cpu_baseline` (oracle/torch_oracle.py) beside the ACTUAL reference on the same cfg-3 flow, same
weights, same inputs, same thread count.  Prints both throughputs, their ratio and the largest output difference.
    python tools/time_oracle_vs_reference.py [log2 rows per chunk] [chunks]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import make_golden  # noqa: E402  (import_reference: the reference + the UMNN placeholder of SURVEY Appendix B)
from oracle import torch_oracle as O  # noqa: E402


def build(L):
    torch.manual_seed(0)
    layers = [L.transforms.PiecewiseRationalQuadraticCouplingTransform(
        L.utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
        lambda a, b: L.nets.ResidualNet(a, b, hidden_features=64, num_blocks=2),
        num_bins=8, tails="linear", tail_bound=3.0) for i in range(32)]
    return L.flows.Flow(L.transforms.CompositeTransform(layers), L.distributions.StandardNormal([64])).eval()


def median_time(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    rows = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 14)
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    ref_lib = make_golden.import_reference()
    ref_flow = build(ref_lib)
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets

    class Mine:
        pass

    Mine.transforms, Mine.nets, Mine.utils, Mine.flows, Mine.distributions = transforms, nets, utils, flows, distributions
    my_flow = build(Mine)
    my_flow.load_state_dict(ref_flow.state_dict())
    x = torch.randn(rows * chunks, 64, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        def run_ref():
            return torch.cat([ref_flow.log_prob(x[i * rows:(i + 1) * rows]) for i in range(chunks)])

        def run_port():
            return torch.cat([O.flow_log_prob(my_flow, x[i * rows:(i + 1) * rows]) for i in range(chunks)])

        t_ref = median_time(run_ref)
        t_port = median_time(run_port)
        d = float((run_ref() - run_port()).abs().max())
    n = rows * chunks
    print(f"cfg 3 log_prob on {threads} threads, {chunks} chunks of {rows} rows: reference {n / t_ref:.0f} samples/s, "
          f"oracle port {n / t_port:.0f} samples/s, port / reference = {t_ref / t_port:.2f}, "
          f"max |d log_prob| = {d:.2e}")


if __name__ == "__main__":
    main()
