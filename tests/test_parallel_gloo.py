"""N > 1 path on CPU: world_size-2 gloo processes run the batch-sharded log_prob reduction of
flowconductor_amd.parallel (the HIP kernels need a GPU, so each rank's local log_prob here is the
CPU oracle evaluating the same flow -- what is under test is sharding + the {sum, count} all-reduce)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flow():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    from _util import Lib

    torch.manual_seed(0)
    d = 6
    layers = [Lib.transforms.PiecewiseRationalQuadraticCouplingTransform(
        Lib.utils.create_alternating_binary_mask(d, even=(l % 2 == 0)),
        lambda i, o: Lib.nets.ResidualNet(i, o, hidden_features=8), num_bins=4, tails="linear", tail_bound=3.0)
        for l in range(2)]
    return Lib.flows.Flow(Lib.transforms.CompositeTransform(layers), Lib.distributions.StandardNormal([d])).eval()


def _worker(rank, world, port, n, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowconductor_amd import parallel
        from oracle import torch_oracle as O

        flow = _flow()
        x = torch.randn(n, 6, generator=torch.Generator().manual_seed(7))
        lo, hi = parallel.shard_bounds(n, rank, world)
        with torch.no_grad():
            mean = parallel.sharded_log_prob_mean(lambda v: O.flow_log_prob(flow, v), x[lo:hi], chunk=5,
                                                  group=dist.group.WORLD)
        if rank == 0:
            torch.save({"mean": mean, "bounds": [parallel.shard_bounds(n, r, world) for r in range(world)]}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 64])
def test_sharded_log_prob_mean_world2(tmp_path, n):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["bounds"][0][0] == 0 and res["bounds"][-1][1] == n
    assert res["bounds"][0][1] == res["bounds"][1][0]
    from oracle import torch_oracle as O

    flow = _flow()
    x = torch.randn(n, 6, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        expect = float(O.flow_log_prob(flow, x).double().mean())
    assert abs(res["mean"] - expect) <= 1e-5 * max(1.0, abs(expect))


def test_shard_bounds_cover_everything():
    from flowconductor_amd import parallel

    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, n, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowconductor_amd import parallel
        from oracle import torch_oracle as O

        flow = _flow()
        x = torch.randn(n, 6, generator=torch.Generator().manual_seed(11))
        lo, hi = parallel.shard_bounds(n, rank, world)
        params = list(flow.parameters())
        # tiny buckets: the flat-bucket split / scatter-back is what is under test
        orig = parallel.allreduce_gradients
        parallel.allreduce_gradients = lambda p, g: orig(p, g, bucket_bytes=256)
        nll = parallel.sharded_nll_backward(lambda v: O.flow_log_prob(flow, v), x[lo:hi], params,
                                            group=dist.group.WORLD)
        if rank == 1:
            torch.save({"nll": nll, "grads": [p.grad.clone() for p in params]}, out_path)
    finally:
        dist.destroy_process_group()


def test_sharded_nll_backward_world2_matches_single_process(tmp_path):
    """Data-parallel gradient all-reduce (uneven shards: 37 rows over 2 ranks) = the gradient of
    the global mean NLL computed in one process."""
    n = 37
    out = str(tmp_path / "grads.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    res = torch.load(out)
    from oracle import torch_oracle as O

    flow = _flow()
    x = torch.randn(n, 6, generator=torch.Generator().manual_seed(11))
    loss = -O.flow_log_prob(flow, x).mean()
    loss.backward()
    expect = float(loss.detach())
    assert abs(res["nll"] - expect) <= 1e-5 * max(1.0, abs(expect))
    for got, p in zip(res["grads"], flow.parameters()):
        assert torch.allclose(got, p.grad, rtol=1e-4, atol=1e-6)


def _plan_worker(rank, world, port, scaling, total, out_path):
    """What bench.py does per rank, on CPU: rank_plan -> its rows of ONE global batch -> sharded mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowconductor_amd import parallel
        from oracle import torch_oracle as O

        plan = parallel.rank_plan(rank, world, rank, scaling, rows_per_gpu=5, total_rows=total)
        flow = _flow()
        rows = world * 5 if scaling == "weak" else total
        x = torch.randn(rows, 6, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            mean = parallel.sharded_log_prob_mean(lambda v: O.flow_log_prob(flow, v), x[plan["row_lo"]:plan["row_hi"]],
                                                  group=dist.group.WORLD)
        plans = [None] * world
        dist.all_gather_object(plans, plan)
        if rank == world - 1:
            torch.save({"mean": mean, "plans": plans, "rows": rows}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scaling,total", [("weak", None), ("strong", 43), ("strong", 5)])
def test_rank_plan_world8_sharded_mean(tmp_path, scaling, total):
    """World size 8 (the driver's largest run): weak plan (equal shards), strong plans with uneven shards (43 rows over
    8 ranks) and with EMPTY shards (5 rows over 8 ranks): the global mean equals the single-process one."""
    out = str(tmp_path / "res.pt")
    mp.spawn(_plan_worker, args=(8, _free_port(), scaling, total, out), nprocs=8, join=True)
    res = torch.load(out)
    plans = res["plans"]
    assert [p["rank"] for p in plans] == list(range(8))
    assert [p["device_index"] for p in plans] == list(range(8))
    assert len({p["seed"] for p in plans}) == 8
    assert plans[0]["row_lo"] == 0 and plans[-1]["row_hi"] == res["rows"]
    assert all(a["row_hi"] == b["row_lo"] for a, b in zip(plans, plans[1:]))
    from oracle import torch_oracle as O

    flow = _flow()
    x = torch.randn(res["rows"], 6, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        expect = float(O.flow_log_prob(flow, x).double().mean())
    assert abs(res["mean"] - expect) <= 1e-5 * max(1.0, abs(expect))
