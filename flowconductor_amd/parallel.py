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


class CollectiveSetupFailed(RuntimeError):
    """``LoglikAllReduce`` could not be built on SOME rank; raised on EVERY rank of the group (they agree first), so the
    callers all take the same fall-back.  ``stuck_helper`` is true on a rank whose ``fc_comm_init_rank`` never returned:
    that process still has a thread parked inside RCCL and should leave through ``os._exit`` when it is done."""

    def __init__(self, message, stuck_helper=False):
        super().__init__(message)
        self.stuck_helper = stuck_helper


class LoglikAllReduce:
    """The path's one collective through the C ABI: ``fc_allreduce_loglik`` (RCCL ``ncclAllReduce`` of two float64 on
    the compute stream).  Built collectively by all ranks of ``group``:

        reducer = LoglikAllReduce(device, group)          # every rank; raises CollectiveSetupFailed on every rank or none
        total, count = reducer(log_prob)                   # per evaluation; returns Python floats (one host sync)
        stats = reducer.reduce_async(log_prob)             # device tensor {sum, count}, no host sync (training loops)

    Set-up protocol (every ``torch.distributed`` call is issued by the CALLING thread inside ``torch.cuda.device(device)``:
    the current device is thread-local, and a NCCL group bound with ``device_id`` refuses tensors of another device):
      1. each rank loads the library, rank 0 draws the 128-byte RCCL id; MIN all-reduce of "ok so far" -- a rank that
         cannot even load RCCL stops everybody HERE, before anyone enters the blocking collective of step 3;
      2. the id travels as a uint8 tensor broadcast (on ``device`` for nccl groups, on the CPU for gloo);
      3. ``fc_comm_init_rank_on_device`` (``ncclCommInitRank``, blocking, cannot be interrupted) runs on a helper thread
         under a deadline; the ABI entry sets the device itself, the helper touches nothing of ``torch.distributed``;
      4. MIN all-reduce of the outcome; on failure the ranks that did get a communicator abort it.
    """

    def __init__(self, device, group=None, init_timeout_s=120.0):
        import threading

        import torch.distributed as dist

        from flowconductor_amd import _hip

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("LoglikAllReduce needs a HIP device, got %s" % (self.device,))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._comm = None
        distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if distributed else 1
        self.rank = dist.get_rank(group) if distributed else 0
        on_device = distributed and dist.get_backend(group) == "nccl"
        flag_device = self.device if on_device else torch.device("cpu")

        def agree(ok):
            """MIN over the ranks of a local success flag (main thread, device context)."""
            if not distributed or self.world == 1:
                return bool(ok)
            flag = torch.tensor([1.0 if ok else 0.0], device=flag_device)
            with torch.cuda.device(self.device):
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return float(flag.item()) == 1.0

        # 1. local, no collective
        error = None
        ident = ctypes.create_string_buffer(128)
        try:
            self._lib = _hip.load()
            if self.rank == 0:
                _hip.check(self._lib.fc_comm_unique_id(ident), "fc_comm_unique_id")
        except Exception as e:      # noqa: BLE001 -- reported through the agreement below
            error = e
        if not agree(error is None):
            raise CollectiveSetupFailed("fc_comm_unique_id / library load failed on some rank (this rank: %r)" % (error,))
        # 2. the id, as a tensor broadcast on the calling thread
        if distributed and self.world > 1:
            wire = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).to(flag_device)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            with torch.cuda.device(self.device):
                dist.broadcast(wire, src=src, group=group)
            ident = ctypes.create_string_buffer(bytes(wire.cpu().tolist()), 128)
        # 3. the blocking RCCL bootstrap, under a deadline
        box = {}
        lib, world, rank, index = self._lib, self.world, self.rank, self.device.index

        lock = threading.Lock()

        def bootstrap():
            comm = ctypes.c_void_p()
            try:
                _hip.check(lib.fc_comm_init_rank_on_device(ctypes.byref(comm), world, ident, rank, index),
                           "fc_comm_init_rank_on_device")
            except Exception as e:  # noqa: BLE001
                with lock:
                    box.setdefault("error", e)
                return
            with lock:
                if box.get("cancelled"):
                    # the caller gave up on this bootstrap (deadline): nobody will ever use or destroy the communicator, and
                    # its peers must not take it for live
                    lib.fc_comm_abort(comm)
                else:
                    box["comm"] = comm

        helper = threading.Thread(target=bootstrap, daemon=True, name="fc_comm_init_rank")
        helper.start()
        helper.join(timeout=init_timeout_s)
        with lock:
            stuck = "comm" not in box and "error" not in box
            if stuck:
                box["cancelled"] = True
                box["error"] = TimeoutError("fc_comm_init_rank_on_device did not return within %g s" % init_timeout_s)
        # 4. agree on the outcome
        if not agree("comm" in box and "error" not in box):
            if "comm" in box and not stuck:
                self._lib.fc_comm_abort(box["comm"])
            raise CollectiveSetupFailed("fc_comm_init_rank failed on some rank (this rank: %r)" % (box.get("error"),),
                                        stuck_helper=stuck)
        self._comm = box["comm"]
        self._stats = torch.zeros(2, dtype=torch.float64, device=self.device)

    def reduce_async(self, log_prob):
        """{sum log_prob, count} over all ranks as a float64 DEVICE tensor of two elements, enqueued on the current
        stream of the device; no host synchronisation.  The tensor is this object's own buffer: read it (or copy it)
        before the next call."""
        from flowconductor_amd import _hip

        stats = self._stats
        stats[0] = log_prob.double().sum()
        stats[1] = float(log_prob.numel())
        with torch.cuda.device(self.device):
            _hip.check(self._lib.fc_allreduce_loglik(ctypes.c_void_p(stats.data_ptr()), self._comm,
                                                     _hip.stream_ptr(self.device)), "fc_allreduce_loglik")
        return stats

    def __call__(self, log_prob):
        total, count = self.reduce_async(log_prob).tolist()
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
    if n == 0:      # an empty shard (fewer rows than ranks): contributes {0, 0}; the reference cannot evaluate 0 rows either
        return inputs.new_zeros(0)
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
