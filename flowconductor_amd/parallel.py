"""Batch-sharded evaluation across the GPUs of one node (one process per GPU).

Every bijector on the path is row-wise over the batch, so ``log_prob`` shards on the batch axis
with no data-path collective: rank r evaluates its own rows; the only exchange is one
all-reduce of ``{sum of log_prob (f64), row count (f64)}`` = 16 bytes per evaluation, over RCCL
(``backend="nccl"`` on ROCm) -- latency-bound, so it rides the tree/one-shot path, not the
ring.  The reference has no distributed code at all; there is nothing to
translate.  Outputs (noise, per-sample log_prob) stay sharded.
"""
import ctypes

import torch


def rank_plan(rank, world, local_rank, scaling="weak", rows_per_gpu=1 << 20, total_rows=None, base_seed=1234):
    """What rank ``rank`` of ``world`` does in a batch-sharded evaluation, as plain data:
    ``device_index`` (one process per GPU: its LOCAL_RANK), ``seed`` (inputs are generated per rank on the device,
    ``base_seed + rank``), the global row range ``[row_lo, row_hi)`` of its contiguous shard and ``n_local``.
    ``scaling="weak"``: every rank owns ``rows_per_gpu`` rows (BASELINE.json configs[2] per GPU; at world = 8 the job
    is configs[3]).  ``scaling="strong"``: ``total_rows`` (configs[3]: 2^23) are split into near-equal shards."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    if scaling == "weak":
        lo, hi = rank * rows_per_gpu, (rank + 1) * rows_per_gpu
    elif scaling == "strong":
        if total_rows is None:
            raise ValueError("strong scaling needs total_rows")
        lo, hi = shard_bounds(total_rows, rank, world)
    else:
        raise ValueError("scaling must be 'weak' or 'strong'")
    return {"rank": rank, "world": world, "device_index": local_rank, "seed": base_seed + rank,
            "row_lo": lo, "row_hi": hi, "n_local": hi - lo, "scaling": scaling}


class LoglikAllReduce:
    """The path's one collective through the C ABI: ``fc_allreduce_loglik`` (RCCL ``ncclAllReduce`` of two float64 on
    the compute stream).  Built collectively by all ranks; the 128-byte RCCL unique id travels over the existing
    ``torch.distributed`` process group (any backend -- it is host data).

        reducer = LoglikAllReduce(device, group)          # every rank
        total, count = reducer(log_prob)                   # per evaluation; returns Python floats
    """

    def __init__(self, device, group=None):
        import torch.distributed as dist

        from flowconductor_amd import _hip

        self._lib = _hip.load()
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if group is not None or dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _hip.check(self._lib.fc_comm_unique_id(ident), "fc_comm_unique_id")
        payload = [ident.raw]
        if self.world > 1:
            dist.broadcast_object_list(payload, src=0, group=group)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):       # ncclCommInitRank binds the current device
            _hip.check(self._lib.fc_comm_init_rank(ctypes.byref(comm), self.world, payload[0], self.rank),
                       "fc_comm_init_rank")
        self._comm = comm
        self._stats = torch.zeros(2, dtype=torch.float64, device=self.device)

    def __call__(self, log_prob):
        from flowconductor_amd import _hip

        stats = self._stats
        stats[0] = log_prob.double().sum()
        stats[1] = float(log_prob.numel())
        with torch.cuda.device(self.device):
            _hip.check(self._lib.fc_allreduce_loglik(ctypes.c_void_p(stats.data_ptr()), self._comm,
                                                     _hip.stream_ptr(self.device)), "fc_allreduce_loglik")
        total, count = stats.tolist()
        return total, count

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.fc_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n, rank, world):
    """Contiguous near-equal shards: rows [lo, hi) of an n-row batch for ``rank``."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_log_prob(log_prob_fn, inputs, context=None, chunk=None):
    """Per-sample log_prob of this rank's rows, optionally in row chunks (bounds the
    ``[chunk, d_t*(3K-1)]`` conditioner-output buffer that each layer materialises)."""
    n = inputs.shape[0]
    if chunk is None or chunk >= n:
        return log_prob_fn(inputs) if context is None else log_prob_fn(inputs, context)
    parts = []
    for lo in range(0, n, chunk):
        hi = min(lo + chunk, n)
        if context is None:
            parts.append(log_prob_fn(inputs[lo:hi]))
        else:
            parts.append(log_prob_fn(inputs[lo:hi], context[lo:hi]))
    return torch.cat(parts)


def allreduce_sum_count(log_prob, group=None):
    """All-reduce {sum, count} in f64; returns (global sum, global count) as Python floats."""
    stats = torch.stack((log_prob.double().sum(), torch.tensor(float(log_prob.numel()), dtype=torch.float64,
                                                               device=log_prob.device)))
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    total, count = stats.tolist()
    return total, count


def sharded_log_prob_mean(log_prob_fn, local_inputs, context=None, chunk=None, group=None, reducer=None):
    """Mean log-likelihood over all ranks' rows; each rank passes only its own shard.  ``reducer``: a
    ``LoglikAllReduce`` (RCCL through the C ABI); otherwise the reduction goes through ``torch.distributed`` on
    ``group`` (gloo on CPU in the tests, nccl = RCCL on GPUs)."""
    lp = local_log_prob(log_prob_fn, local_inputs, context, chunk)
    total, count = reducer(lp) if reducer is not None else allreduce_sum_count(lp, group)
    return total / count if count else float("nan")


def allreduce_gradients(parameters, group=None, bucket_bytes=64 << 20):
    """Sum the ``.grad`` of ``parameters`` over the ranks, in place, in flat buckets.

    The whole cfg-3 flow has 8.5 MB of gradients: one bucket = one ring all-reduce
    (2 (R-1)/R x 8.5 MB per rank over ~153 GB/s xGMI links, ~0.1 ms at R = 8), so the default
    bucket holds everything; larger models split at ``bucket_bytes`` so that the first buckets
    can be reduced while later ones are still being flattened.  Parameters without a gradient
    on this rank contribute zeros (every rank must reduce the same element count)."""
    params = [p for p in parameters if p.requires_grad]
    if group is None or not params:
        return
    import torch.distributed as dist

    bucket, size = [], 0
    buckets = []
    for p in params:
        nbytes = p.numel() * p.element_size()
        if bucket and (size + nbytes > bucket_bytes or p.dtype != bucket[0].dtype):
            buckets.append(bucket)
            bucket, size = [], 0
        bucket.append(p)
        size += nbytes
    buckets.append(bucket)
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        offset = 0
        for p in bucket:
            piece = flat[offset:offset + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = piece.clone()
            else:
                p.grad.copy_(piece)
            offset += p.numel()


def sharded_nll_backward(log_prob_fn, local_inputs, parameters, context=None, group=None):
    """One data-parallel training step's loss and gradients: the global mean
    negative log-likelihood, with ``.grad`` of ``parameters`` = its exact gradient on every rank.

    Each rank differentiates ``-sum(local log_prob) / N_global`` (shards may be uneven, so the
    global row count is reduced first: 8 bytes), then the gradients are summed over the ranks.
    Returns the global mean NLL as a Python float; the caller steps its optimizer."""
    parameters = list(parameters)
    count = torch.tensor(float(local_inputs.shape[0]), dtype=torch.float64, device=local_inputs.device)
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
    lp = log_prob_fn(local_inputs) if context is None else log_prob_fn(local_inputs, context)
    loss = -(lp.sum() / count.to(lp.dtype))
    loss.backward()
    allreduce_gradients(parameters, group)
    total = loss.detach().double().reshape(1)
    if group is not None:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return float(total)
