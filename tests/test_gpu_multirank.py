"""The N > 1 code path of bench.py on a one-GPU box, in fresh CHILD processes (never a re-exec of a process that has
touched the GPU): (1) one RCCL rank under torch.distributed.run -- process group bound to the device, the C-ABI reducer
(fc_comm_init_rank_on_device on its helper thread), every barrier, fc_allreduce_loglik per step, the MAX all-reduce of the
time, the gather of the per-rank records; (2) two ranks sharing device 0 over gloo, which drives the reducer's whole
set-up protocol with real peers (agreement, id broadcast, bootstrap under the deadline, agreement): RCCL either builds a
two-rank communicator on the shared device or refuses it ("duplicate GPU") on both ranks, and then both ranks must fall back
to torch.distributed together and still print a valid line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(nproc, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--batch-log2", "16", "--no-cpu-baseline",
           "--no-configs"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, "bench.py failed:\n%s\n%s" % (res.stdout[-2000:], res.stderr[-4000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "expected ONE JSON line, got %d:\n%s" % (len(lines), res.stdout[-2000:])
    return json.loads(lines[0]), res.stderr


def _single_process_mean(per_rank, device):
    """mean log_prob of the union of the ranks' shards, evaluated in THIS process: same flow (bench.build_flow under
    manual_seed(0)), same inputs (each rank's generator seed and row count), float64 sum of the per-sample values."""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    flow = bench.build_flow().to(device)
    total, count = 0.0, 0
    with torch.no_grad():
        for r in per_rank:
            n = r["rows"][1] - r["rows"][0]
            gen = torch.Generator(device=device).manual_seed(r["seed"])
            x = torch.randn(n, bench.FEATURES, device=device, generator=gen)
            total += float(flow.log_prob(x).double().sum())
            count += n
    return total / count


def _check_reduced_mean(out, device):
    """whichever reducer ran (ABI / torch.distributed), the all-reduced {sum, count} must give the single-process mean"""
    want = _single_process_mean(out["per_rank"], device)
    got = out["config"]["mean_log_prob"]
    assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (got, want, out["reducer"])


def test_one_rccl_rank_through_the_abi_reducer(device):
    out, _ = _run_bench(1, ["--rehearse-dist"])
    assert out["rccl_ranks"] == 1 and out["reducer"] == "abi" and out["degraded"] is False
    _check_reduced_mean(out, device)
    assert out["loglik_allreduce"].startswith("fc_allreduce_loglik (RCCL ncclAllReduce through the C ABI, 1 ranks)")
    assert out["n_gpus"] == 1 and len(out["per_rank"]) == 1 and out["value"] > 0
    assert out["rank_skew"]["max_over_min"] == 1.0
    assert out["config"]["samples_per_gpu"] == 1 << 16


def test_two_ranks_sharing_the_device_agree_on_the_reducer(device):
    out, err = _run_bench(2, ["--dist-backend", "gloo", "--share-device0", "--comm-timeout", "60"])
    assert out["n_gpus"] == 2 and len(out["per_rank"]) == 2
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert [r["rows"] for r in out["per_rank"]] == [[0, 1 << 16], [1 << 16, 2 << 16]]
    assert out["per_rank"][0]["seed"] != out["per_rank"][1]["seed"]
    note = out["loglik_allreduce"]
    assert (note.startswith("fc_allreduce_loglik (RCCL ncclAllReduce through the C ABI, 2 ranks)")
            or "failed to initialise on some rank" in note), note
    # machine-readable: either the ABI reducer on every rank, or the agreed fall-back flagged as degraded
    assert (out["reducer"], out["degraded"]) in (("abi", False), ("torch", True)), (out["reducer"], out["degraded"])
    assert out["value"] > 0 and out["config"]["global_batch"] == 2 << 16
    assert out["rank_skew"]["max_over_min"] >= 1.0
    _check_reduced_mean(out, device)


def test_strong_scaling_plan_two_ranks(device):
    out, _ = _run_bench(2, ["--dist-backend", "gloo", "--share-device0", "--loglik-allreduce", "torch",
                            "--scaling", "strong", "--total-log2", "17"])
    assert out["scaling"] == "strong" and out["reducer"] == "torch" and out["degraded"] is False
    assert [r["rows"] for r in out["per_rank"]] == [[0, 1 << 16], [1 << 16, 1 << 17]]
    assert out["config"]["global_batch"] == 1 << 17


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_four_ranks_sharing_the_device_whole_protocol(scaling, device):
    """Four peers through the reducer's agreement protocol and both sharding plans on one GPU (BASELINE.json configs[3] is
    the 8-rank form; the GPU boxes allow at most 6 processes on the card at once and this test's parent is one of them --
    the 8-rank plans run on CPU in tests/test_parallel_gloo.py)."""
    extra = ["--dist-backend", "gloo", "--share-device0", "--comm-timeout", "60", "--batch-log2", "12", "--scaling", scaling]
    if scaling == "strong":
        extra += ["--total-log2", "14"]
    out, _ = _run_bench(4, extra)
    assert out["n_gpus"] == 4 and [r["rank"] for r in out["per_rank"]] == [0, 1, 2, 3]
    assert (out["reducer"], out["degraded"]) in (("abi", False), ("torch", True))
    assert [r["rows"] for r in out["per_rank"]] == [[i << 12, (i + 1) << 12] for i in range(4)]
    assert out["config"]["global_batch"] == 4 << 12
    _check_reduced_mean(out, device)
