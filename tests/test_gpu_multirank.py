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


def test_one_rccl_rank_through_the_abi_reducer(device):
    out, _ = _run_bench(1, ["--rehearse-dist"])
    assert out["rccl_ranks"] == 1
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
    assert out["value"] > 0 and out["config"]["global_batch"] == 2 << 16
    assert out["rank_skew"]["max_over_min"] >= 1.0


def test_strong_scaling_plan_two_ranks(device):
    out, _ = _run_bench(2, ["--dist-backend", "gloo", "--share-device0", "--loglik-allreduce", "torch",
                            "--scaling", "strong", "--total-log2", "17"])
    assert out["scaling"] == "strong"
    assert [r["rows"] for r in out["per_rank"]] == [[0, 1 << 16], [1 << 16, 1 << 17]]
    assert out["config"]["global_batch"] == 1 << 17
