"""Batch-sharded evaluation across the GPUs of one node (one process per GPU).

Every bijector on the path is row-wise over the batch, so ``log_prob`` shards on the batch axis
with no data-path collective: rank r evaluates its own rows; the only exchange is one
all-reduce of ``{sum of log_prob (f64), row count (f64)}`` = 16 bytes per evaluation, over RCCL
(``backend="nccl"`` on ROCm) -- latency-bound, so it rides the tree/one-shot path, not the
ring.  The reference has no distributed code at all; there is nothing to
translate.  Outputs (noise, per-sample log_prob) stay sharded.
"""
import torch


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


def sharded_log_prob_mean(log_prob_fn, local_inputs, context=None, chunk=None, group=None):
    """Mean log-likelihood over all ranks' rows; each rank passes only its own shard."""
    lp = local_log_prob(log_prob_fn, local_inputs, context, chunk)
    total, count = allreduce_sum_count(lp, group)
    return total / count if count else float("nan")
