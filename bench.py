#!/usr/bin/env python
"""Headline benchmark: log_prob samples/sec of a 32-layer RQ-NSF coupling flow (D=64, K=8)
on synthetic Gaussian batches, one process per GPU, batch-sharded (BASELINE.json cfg 3/4).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                       # weak scaling: 2^20 samples per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --scaling strong --total-log2 23           # BASELINE.json configs[3]: 2^23 rows over the ranks

One "step" = one ``Flow.log_prob`` pass of every rank's 2^20-sample shard through all 32
layers (conditioners on PyTorch-ROCm, bijectors in the HIP kernels) + the RCCL all-reduce of
{sum log_prob, count}.  Inputs are resident in HBM before the timed region.  Prints ONE JSON
line on rank 0 (contract in the task statement), carrying ``roofline`` (the dominant kernel of the timed
region: the fused final-Linear + spline kernel, against its algorithmic HBM bytes), ``roofline_hidden``
(the conditioner's hidden-layer kernel), ``roofline_unfused_rq_spline`` (the HBM-bound stand-alone
bijector kernel, timed in an extra pass) and ``cpu_baseline`` (the CPU oracle timed on the host cores,
rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import flowconductor_amd  # noqa: E402,F401
from flowconductor_amd import distributions, flows, ops, options, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

FEATURES, LAYERS, BINS, HIDDEN, BLOCKS, TAIL_BOUND = 64, 32, 8, 64, 2, 3.0
HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak, same guide (2:1-sparsity figures excluded)


def build_flow():
    """BASELINE.json cfg 3: 32 x PiecewiseRationalQuadraticCouplingTransform,
    alternating masks, ResidualNet(hidden 64, 2 blocks) conditioners, linear tails at 3."""
    torch.manual_seed(0)
    layers = []
    for l in range(LAYERS):
        layers.append(transforms.PiecewiseRationalQuadraticCouplingTransform(
            utils.create_alternating_binary_mask(FEATURES, even=(l % 2 == 0)),
            lambda i, o: nets.ResidualNet(i, o, hidden_features=HIDDEN, num_blocks=BLOCKS),
            num_bins=BINS, tails="linear", tail_bound=TAIL_BOUND))
    return flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([FEATURES])).eval()


TRAFFIC_PROFILE = "profiles/r04_hbm_traffic.json"
FUSED_COUNTERS = "profiles/r04_fused_sq_counters.txt"
HIDDEN_COUNTERS = "profiles/r04_hidden_sq_counters.txt"
# Set by main(): True = a profile taken from ANOTHER build of the library ends the run (tools/profile_bench.sh verifies its own
# output this way); False (default) = its numbers are left out of the line (traffic / issue_bound null) and the mismatch is
# reported in `profile_mismatch` -- the driver's end-of-round run must not die of a stale profile.
STRICT_PROFILES = False
PROFILE_MISMATCH = []


def _profile_sha_ok(name, recorded):
    """True when the profile file `name` was taken from the library loaded now (or predates the sha256 bookkeeping)."""
    from flowconductor_amd import _hip
    if not recorded:
        PROFILE_MISMATCH.append({"profile": name, "reason": "no library.sha256 recorded"})
        if STRICT_PROFILES:
            raise SystemExit("bench.py: %s carries no library.sha256 -- re-run tools/profile_bench.sh" % name)
        return False
    loaded = _hip.library_info()["sha256"]
    if recorded != loaded:
        PROFILE_MISMATCH.append({"profile": name, "recorded_sha256": recorded, "loaded_sha256": loaded})
        if STRICT_PROFILES:
            raise SystemExit("bench.py: %s was taken from library %s, the loaded library is %s -- re-run tools/profile_bench.sh"
                             % (name, recorded[:16], loaded[:16]))
        print("[bench] %s was taken from another library build (%s..., loaded %s...): its numbers are left out"
              % (name, recorded[:12], loaded[:12]), file=sys.stderr)
        return False
    return True
# kernel symbol (substring) each C-ABI entry launches in this flow: the committed PMC profile must have counted THAT
# kernel, or its number does not belong in this line
EXPECTED_KERNELS = {"fc_rq_spline_fused_linear": "rq_fused_linear_kernel3", "fc_resnet_hidden": "resnet_hidden_kernel",
                    "fc_rq_spline": "rq_wave_kernel"}


def measured_traffic_per_launch(entry, rows_per_launch):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command
    (tools/profile_bench.sh: FETCH_SIZE x2 + WRITE_SIZE, separate --pmc runs).  PMC counters cannot be read from
    inside this process, so the number comes from profiles/ and the line names the profile (`traffic_profile`).
    None if the profile is absent or this run's launch shape differs from the profiled one; a profile that counted a
    DIFFERENT kernel than the one this flow launches is an error, not a stale number."""
    try:
        rec = json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))
    except (OSError, ValueError):
        return None
    if entry not in rec or rows_per_launch != (1 << 20):
        return None
    if not _profile_sha_ok(TRAFFIC_PROFILE, rec.get("library", {}).get("sha256")):
        return None
    if rec[entry].get("kernel") != EXPECTED_KERNELS[entry]:
        raise SystemExit("bench.py: %s counted kernel %r for %s, this flow launches %r -- re-run tools/profile_bench.sh"
                         % (TRAFFIC_PROFILE, rec[entry].get("kernel"), entry, EXPECTED_KERNELS[entry]))
    return rec[entry]["traffic_bytes_per_launch"]


def issue_bound(counters_file, launch_ms, rows_per_launch, valu_cycles):
    """How much of the SIMDs' instruction-issue time a compute-bound kernel uses: wave-level VALU and MFMA instruction
    counts per launch from the committed SQ-counter passes (tools/probe/pmc_kernel.sh) priced at `valu_cycles` per VALU
    instruction (measured, tools/probe/valu_costs.hip: ~3.6 at the fused kernel's two waves per SIMD, ~2.6 at the hidden
    kernel's four; transcendentals twice that) and 8 per MFMA (MI355X_MICROARCH.md: a 16x16x32 MFMA holds the SIMD's vector
    issue for 8 of its 16 cycles), against 256 CUs x 4 SIMDs x the 2.4 GHz peak clock over this run's launch duration
    (the kernels sustain 2.1 / 1.8 GHz, so 1.0 is not reachable).  None if the profile is absent."""
    try:
        if rows_per_launch != (1 << 20):
            return None
        vals, sha = {}, None
        for line in open(os.path.join(ROOT, counters_file)):
            parts = line.split()
            if len(parts) >= 2 and parts[0] == "library.sha256":
                sha = parts[1]
            if len(parts) >= 3 and parts[0] in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32"):
                vals[parts[0]] = float(parts[2])
        if not _profile_sha_ok(counters_file, sha):
            return None
        cycles = valu_cycles * (vals["SQ_INSTS_VALU"] + vals.get("SQ_INSTS_VALU_TRANS_F32", 0.0)) + 8.0 * vals["SQ_INSTS_MFMA"]
        avail = 256 * 4 * 2.4e9 * launch_ms * 1e-3
        return {"valu_wave_instructions": vals["SQ_INSTS_VALU"], "mfma_wave_instructions": vals["SQ_INSTS_MFMA"],
                "cycles_per_valu_instruction": valu_cycles, "issue_cycles": cycles, "simd_cycles_at_2.4GHz": avail,
                "frac": cycles / avail, "source": counters_file}
    except (OSError, ValueError, KeyError):
        return None


def algorithmic_bytes_per_sample_layer():
    """B = 4*d_t*(P + 2) + 8: params + x_t + y_t + logabsdet r/w."""
    d_t = FEATURES // 2
    p = 3 * BINS - 1
    return 4 * d_t * (p + 2) + 8


def host_cores():
    """CPU threads this process may actually use (affinity and cgroup quota), not the host's count."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(flow_cpu, chunk, repeats=3, budget_s=30.0):
    """Time the CPU oracle (torch-CPU restatement of the reference's op sequence) on host cores, as BASELINE.md section 3
    plans it: chunks of 2^16 rows, one warm-up, 3 timed repeats, median (throughput falls with the chunk size on the CPU:
    mask-gather temporaries).  Bounded: stops repeating once `budget_s` seconds of timed work are spent."""
    from oracle import torch_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(repeats * chunk, FEATURES, generator=gen)
    times = []
    with torch.no_grad():
        O.flow_log_prob(flow_cpu, x[:2048].clone())  # warm-up
        for r in range(repeats):
            t0 = time.perf_counter()
            O.flow_log_prob(flow_cpu, x[r * chunk:(r + 1) * chunk].clone())
            times.append(time.perf_counter() - t0)
            if sum(times) > budget_s:
                break
    med = sorted(times)[len(times) // 2]
    return {"value": chunk / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d chunk(s) of %d samples through the 32-layer flow, median of %s s"
                      % (len(times), chunk, "/".join("%.1f" % t for t in times))}


def parity(flow, flow_cpu, device, rows=2048):
    """GPU vs CPU oracle on the same weights / inputs: max |delta logabsdet| and max relative sample error after
    all 32 layers, next to the float64 truth -- the 1e-5 target sits at the reference's own float32 noise floor
    (SURVEY section 7), so the honest gate is "GPU error against float64 <= reference-f32 error against float64
    and ~all elements within 1e-5 of the f32 reference"."""
    import copy

    from oracle import torch_oracle as O

    gen = torch.Generator().manual_seed(7)
    x = torch.randn(rows, FEATURES, generator=gen)
    with torch.no_grad():
        z_ref, lad_ref = O.transform_apply(flow_cpu._transform, x.clone())
        z64, lad64 = O.transform_apply(copy.deepcopy(flow_cpu._transform).double(), x.double())
        z, lad = flow._transform(x.to(device))
    z, lad = z.cpu(), lad.cpu()

    def rel(a, b):
        return (a.double() - b.double()).abs() / b.double().abs().clamp_min(1.0)

    dz = rel(z, z_ref)
    return {"max_abs_dlogabsdet": float((lad - lad_ref).abs().max()),
            "max_rel_dlogabsdet": float(rel(lad, lad_ref).max()),
            "max_rel_dsamples": float(dz.max()),
            "frac_samples_within_1e-5": float((dz <= 1e-5).double().mean()),
            "vs_float64": {"gpu_max_rel_dsamples": float(rel(z, z64).max()),
                           "cpu_f32_oracle_max_rel_dsamples": float(rel(z_ref, z64).max()),
                           "gpu_max_rel_dlogabsdet": float(rel(lad, lad64).max()),
                           "cpu_f32_oracle_max_rel_dlogabsdet": float(rel(lad_ref, lad64).max())},
            "rows": rows}


def trained_like(flow_cpu, rows=2048):
    """BASELINE.md section 3's second variant: conditioner outputs ~ N(0, 1) so that every spline bin is exercised
    (default-initialised conditioners give near-identity splines).  Re-draws each layer's final Linear, layer by
    layer on the data the previous layers produce, so that its outputs have unit scale."""
    import copy

    from oracle import torch_oracle as O

    flow2 = copy.deepcopy(flow_cpu)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(rows, FEATURES, generator=gen)
    with torch.no_grad():
        for t in flow2._transform._transforms:
            net = t.transform_net
            h = net.hidden(x[:, t.identity_features])
            lin = net.final_layer
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=gen) / (HIDDEN ** 0.5 * float(h.std())))
            lin.bias.copy_(torch.randn(lin.bias.shape, generator=gen) * 0.1)
            x, _ = O.transform_apply(t, x)
    return flow2


LEGACY_ENV_SWITCHES = ("FLOWCON_HIP_LIB", "FC_FUSED", "FC_FUSED_HIDDEN", "FC_SYLVESTER_MM", "FC_RQ_PATH",
                       "FC_AR_INCREMENTAL")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch-log2 samples on every GPU (configs[2] per GPU; 8 GPUs = configs[3]); "
                         "strong: --total-log2 samples split over the GPUs (configs[3] at every N)")
    ap.add_argument("--batch-log2", type=int, default=20, help="weak scaling: log2 samples per GPU")
    ap.add_argument("--total-log2", type=int, default=23, help="strong scaling: log2 samples of the whole job")
    ap.add_argument("--chunk-log2", type=int, default=0, help="log2 rows per pass through the stack (0 = whole shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary `configs` block (cfg 1, 2, 5, K=10)")
    ap.add_argument("--strict-profiles", action="store_true",
                    help="end the run (SystemExit) when a committed profile (HBM traffic, SQ counters) was taken from another "
                         "build of the library than the one loaded; default: leave its numbers out and say so")
    ap.add_argument("--cpu-sample-log2", type=int, default=16, help="rows per CPU-baseline chunk (BASELINE.md: 2^16)")
    ap.add_argument("--loglik-allreduce", default="abi", choices=["abi", "torch"],
                    help="abi: fc_allreduce_loglik (RCCL through the C ABI); torch: torch.distributed.all_reduce")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device0 rehearses the N>1 code path on a one-GPU box")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="rehearsal only: set up the process group, the C-ABI reducer and every collective of the N>1 "
                         "path although WORLD_SIZE is 1 (one-GPU box: RCCL with a single rank)")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="deadline in seconds for the RCCL bootstrap of the C-ABI reducer (then: torch.distributed path)")
    ap.add_argument("--share-device0", action="store_true",
                    help="rehearsal only: every rank on the single GPU of the box (with --dist-backend gloo)")
    args = ap.parse_args()
    global STRICT_PROFILES
    STRICT_PROFILES = args.strict_profiles

    stray = [k for k in LEGACY_ENV_SWITCHES if k in os.environ]
    if stray:
        raise SystemExit("bench.py: %s set in the environment; kernel selection is not driven by environment variables "
                         "(flowconductor_amd.options) -- unset them" % ", ".join(stray))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    from flowconductor_amd import _hip, parallel

    plan = parallel.rank_plan(rank, world, 0 if args.share_device0 else local_rank, args.scaling,
                              rows_per_gpu=1 << args.batch_log2, total_rows=1 << args.total_log2)
    device = torch.device("cuda", plan["device_index"])
    torch.cuda.set_device(device)
    dist = None
    collective = world > 1 or args.rehearse_dist      # the N>1 code path (process group, reducer, barriers)
    if collective:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    flow = build_flow().to(device)      # weights replicated: every rank builds the same flow under manual_seed(0)
    n_local = plan["n_local"]
    gen = torch.Generator(device=device).manual_seed(plan["seed"])
    x = torch.randn(n_local, FEATURES, device=device, generator=gen)
    chunk = (1 << args.chunk_log2) if args.chunk_log2 else None

    # the path's one collective: {sum log_prob, count}, 16 bytes, through the C ABI (RCCL) when the job runs on RCCL
    # `reducer_kind` / `degraded` are the machine-readable form of the note: "abi" = fc_allreduce_loglik (RCCL through the C ABI),
    # "torch" = torch.distributed.all_reduce, "none" = one rank without the N > 1 code path; degraded = the ABI reducer was
    # asked for (the default on RCCL) and the job fell back to torch.distributed
    reducer, reducer_note, stuck_helper, reducer_kind, degraded = None, "none (1 rank)", False, "none", False
    if collective:
        reducer_kind = "torch"
        reducer_note = "torch.distributed.all_reduce (%s)" % args.dist_backend
        if args.loglik_allreduce == "abi" and (args.dist_backend == "nccl" or args.share_device0):
            # Collective set-up on THIS thread (it owns the device); only the blocking RCCL bootstrap inside runs under a
            # deadline.  Either every rank gets a reducer or every rank gets CollectiveSetupFailed (they agree inside).
            try:
                reducer = parallel.LoglikAllReduce(device, dist.group.WORLD, init_timeout_s=args.comm_timeout)
                reducer_note = "fc_allreduce_loglik (RCCL ncclAllReduce through the C ABI, %d ranks)" % world
                reducer_kind = "abi"
            except parallel.CollectiveSetupFailed as e:
                log("rank %d: fc_allreduce_loglik unavailable (%s)" % (rank, e))
                stuck_helper = stuck_helper or e.stuck_helper
                degraded = True
                reducer_note += " [fc_allreduce_loglik failed to initialise on some rank: %s]" % e

    def step():
        with torch.no_grad():
            return parallel.sharded_log_prob_mean(flow.log_prob, x, chunk=chunk, reducer=reducer,
                                                  group=dist.group.WORLD if collective else None)

    torch.set_num_threads(host_cores())
    log("rank %d/%d on cuda:%d (%s): %d samples (rows %d..%d, seed %d), %s scaling, host cores %d"
        % (rank, world, plan["device_index"], torch.cuda.get_device_name(device), n_local, plan["row_lo"],
           plan["row_hi"], plan["seed"], args.scaling, host_cores()))
    # Steady state is the metric: the first two passes of a process run ~10 % slower (allocator growth,
    # clock ramp -- profiles/r01h_per_step_launch_us.json), so two set-up passes precede the W warm-up steps.
    for _ in range(2 + args.warmup):
        step()
    log("warm-up done")
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    timer = ops.KernelTimer("fc_rq_spline")
    timer_fused = ops.KernelTimer("fc_rq_spline_fused_linear")
    timer_hidden = ops.KernelTimer("fc_resnet_hidden")
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            # per-launch HIP-event pairs on the LAST timed step only: the 128 event records of a step cost 0.5 ms
            # of launch gaps (tools/probe/bench_step_overheads.py), which the other steps do not pay
            with timer, timer_fused, timer_hidden:
                mean_lp = step()
        else:
            mean_lp = step()
    torch.cuda.synchronize(device)
    local_elapsed = time.perf_counter() - t0      # this rank's own K steps (before waiting for the others)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    per_rank = [{"rank": rank, "device_index": plan["device_index"], "seed": plan["seed"], "rows": [plan["row_lo"], plan["row_hi"]],
                 "ms_per_step": 1e3 * local_elapsed / args.steps}]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    log("timed region done: %.3f s for %d steps" % (elapsed, args.steps))
    if rank == 0:
        total = sum(r["rows"][1] - r["rows"][0] for r in per_rank) * args.steps
        rows_per_launch = n_local if chunk is None else min(chunk, n_local)
        alg_bytes = algorithmic_bytes_per_sample_layer() * rows_per_launch
        fused_ms = timer_fused.durations_ms()
        hidden_ms = timer_hidden.durations_ms()
        fused_path = len(fused_ms) > 0
        if fused_path:
            # the stand-alone spline kernel does not run in the fused flow: time it in one extra, untimed
            # pass with the fusion switched off (same flow, same inputs) for the HBM-roofline entry
            with options.override(fused_final_layer=False), ops.KernelTimer("fc_rq_spline") as extra, torch.no_grad():
                parallel.local_log_prob(flow.log_prob, x, chunk=chunk)  # rank-local: no collective here
            torch.cuda.synchronize(device)
            kernel_ms = extra.durations_ms()
        else:
            kernel_ms = timer.durations_ms()
        launches = len(kernel_ms)
        avg_ms = sum(kernel_ms) / max(launches, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        out = {
            "metric": "log_prob samples/sec, 32xRQ-NSF coupling D=64 K=8",
            "value": total / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": "f32 inputs / outputs / spline arithmetic (VALU); the conditioner's matrix products run as "
                          "3-term scaled two-piece f16 splits on v_mfma_f32_16x16x32_f16 with f32 accumulation "
                          "(f32-GEMM accuracy: error vs float64 1.7e-7 of sum|W||h|, an f32 GEMM's own is 4.1e-7)",
            "data": "synthetic N(0,I) inputs generated on device (seed 1234 + rank); default-init weights under "
                    "manual_seed(0), replicated",
            "config": {"workload": ("BASELINE.json configs[2]: 32-layer RQ-NSF coupling flow log_prob, D=64, K=8 bins, "
                                    "linear tails, ResidualNet(64, 2 blocks) conditioners, 2^%d samples per GPU"
                                    % args.batch_log2) if args.scaling == "weak" else
                                   ("BASELINE.json configs[3]: the same flow, 2^%d samples sharded over %d GPU(s)"
                                    % (args.total_log2, world)),
                       "samples_per_gpu": n_local, "global_batch": total // args.steps,
                       "chunk_rows": rows_per_launch, "parallelism": "batch-sharded dp%d" % world,
                       "mean_log_prob": mean_lp},
            "rccl_ranks": world if (collective and args.dist_backend == "nccl") else 0,
            "loglik_allreduce": reducer_note,
            "reducer": reducer_kind,
            "degraded": degraded,
            "per_rank": per_rank,
            "rank_skew": {"max_ms_per_step": max(r["ms_per_step"] for r in per_rank),
                          "min_ms_per_step": min(r["ms_per_step"] for r in per_rank),
                          "max_over_min": max(r["ms_per_step"] for r in per_rank) / min(r["ms_per_step"] for r in per_rank)},
            "library": _hip.library_info(),
            "options": options.snapshot(),
        }
        out["traffic_profile"] = TRAFFIC_PROFILE
        src = TRAFFIC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH_SIZE x2 gfx950 correction)"
        unfused = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_per_launch("fc_rq_spline", rows_per_launch),
                   "traffic_source": src,
                   "kernel": "fc_rq_spline -> fc::rq_wave_kernel<8, true>", "launches_timed": launches,
                   "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                   "measured_in": "extra untimed pass with options.override(fused_final_layer=False)" if fused_path
                   else "timed region"}
        if not fused_path:
            out["roofline"] = unfused
        else:
            # Dominant kernel of the timed region: the conditioner's final Linear (64 -> 736) fused with the
            # spline.  Its algorithmic HBM bytes per sample: h in (4*64) + x in (4*D) + y out (4*D) + logabsdet (4);
            # the [N, 736] parameter tensor never exists in memory.  The kernel is bound by VALU issue (the
            # spline arithmetic, ~290 VALU instructions per element), not by HBM or the matrix pipe: DESIGN.md 4.
            f_avg = sum(fused_ms) / len(fused_ms)
            f_bytes = (4 * HIDDEN + 8 * FEATURES + 4) * rows_per_launch
            f_gbs = f_bytes / (f_avg * 1e-3) / 1e9
            flops = 2.0 * HIDDEN * (FEATURES // 2) * (3 * BINS - 1) * rows_per_launch
            # `bound` names the resource that actually binds the kernel; achieved / peak / frac stay
            # what the contract defines: algorithmic HBM bytes per launch over the launch time, against the HBM peak
            out["roofline"] = {"bound": "valu_issue", "roof": "hbm", "achieved": f_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": f_gbs / HBM_PEAK_GBS,
                               "traffic": measured_traffic_per_launch("fc_rq_spline_fused_linear", rows_per_launch),
                               "traffic_source": src,
                               "kernel": "fc_rq_spline_fused_linear -> fc::rq_fused_linear_kernel3<false, 64, 2, true, true>",
                               "launches_timed": len(fused_ms), "timed_in": "last step of the timed region",
                               "avg_launch_ms": f_avg,
                               "algorithmic_bytes_per_launch": f_bytes,
                               "share_of_step": sum(fused_ms) / (1e3 * elapsed / args.steps),
                               "binding_resource": "valu_issue",
                               "limiter": "VALU issue (spline arithmetic); SQ counters in " + FUSED_COUNTERS + "; "
                                          "`frac` is the distance to the HBM roof the contract asks for, "
                                          "`issue_bound.frac` the share of the SIMDs' issue cycles in use",
                               "issue_bound": issue_bound(FUSED_COUNTERS, f_avg, rows_per_launch, 3.6),
                               # BASELINE.md section 4 prices a coupling bijector at B = 4 d_t (P + 2) + 8 bytes per
                               # sample and layer (parameters read from HBM).  The fused kernel never moves them; in
                               # that accounting it delivers:
                               "baseline_md_accounting": {
                                   "bytes_per_sample_layer": algorithmic_bytes_per_sample_layer(),
                                   "kernel_GBs": alg_bytes / (f_avg * 1e-3) / 1e9,
                                   "kernel_frac_of_hbm_peak": alg_bytes / (f_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "whole_step_GBs": alg_bytes * LAYERS / (elapsed / args.steps) / 1e9,
                                   "whole_step_frac_of_hbm_peak": alg_bytes * LAYERS / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                   "note": "whole step = conditioner + bijector + base distribution, per GPU"},
                               "matrix_pipe": {"algorithmic_tflops": flops / (f_avg * 1e-3) / 1e12,
                                               "executed_tflops": 3.0 * (24.0 / 23.0) * flops / (f_avg * 1e-3) / 1e12,
                                               "peak_f16_dense_tflops": MFMA_F16_PEAK_TFLOPS,
                                               "note": "f32 product as 3 split-f16 MFMA terms, 23 -> 24 padded rows per dim"}}
            if hidden_ms:
                h_avg = sum(hidden_ms) / len(hidden_ms)
                h_bytes = (4 * FEATURES + 4 * HIDDEN) * rows_per_launch
                hflops = 2.0 * (HIDDEN * (FEATURES // 2) + 2 * BLOCKS * HIDDEN * HIDDEN) * rows_per_launch
                h_gbs = h_bytes / (h_avg * 1e-3) / 1e9
                out["roofline_hidden"] = {"bound": "mfma_issue", "roof": "hbm", "achieved": h_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": h_gbs / HBM_PEAK_GBS,
                                          "traffic": measured_traffic_per_launch("fc_resnet_hidden", rows_per_launch),
                                          "kernel": "fc_resnet_hidden -> fc::resnet_hidden_kernel<2, 1, 0, 0, 1>",
                                          "launches_timed": len(hidden_ms), "avg_launch_ms": h_avg,
                                          "algorithmic_bytes_per_launch": h_bytes,
                                          "share_of_step": sum(hidden_ms) / (1e3 * elapsed / args.steps),
                                          "binding_resource": "mfma_issue / dependent-latency (a wave walks the layers serially)",
                                          "issue_bound": issue_bound(HIDDEN_COUNTERS, h_avg,
                                                                     rows_per_launch, 2.6),
                                          "matrix_pipe": {"algorithmic_tflops": hflops / (h_avg * 1e-3) / 1e12,
                                                          "executed_tflops": 3.0 * hflops / (h_avg * 1e-3) / 1e12,
                                                          "peak_f16_dense_tflops": MFMA_F16_PEAK_TFLOPS}}
            # the stand-alone bijector kernel (one thread per (sample, dim), parameters read from HBM: the
            # north-star's own definition of the hot kernel), which the fused flow no longer launches
            out["roofline_unfused_rq_spline"] = unfused
        if world == 1:
            flow_cpu = build_flow()
            out["parity"] = parity(flow, flow_cpu, device)
            log("parity done: %s" % out["parity"])
            # the same check with trained-like conditioners (parameters ~ N(0, 1): every bin, steep derivatives)
            import copy
            flow_tl_cpu = trained_like(flow_cpu)
            flow_tl = copy.deepcopy(flow_tl_cpu).to(device).eval()
            out["parity_trained_like"] = parity(flow_tl, flow_tl_cpu, device)
            del flow_tl
            log("parity (trained-like weights) done: %s" % out["parity_trained_like"])
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(flow_cpu, 1 << args.cpu_sample_log2)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            if not args.no_configs:
                # the other BASELINE.json configurations + the reference's default layer shape, same instrumentation
                del x, flow
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_configs

                out["configs"] = bench_configs.run(device, log=log)
        if PROFILE_MISMATCH:
            out["profile_mismatch"] = PROFILE_MISMATCH
        print(json.dumps(out))
    if reducer is not None:
        reducer.close()
    if dist is not None:
        if stuck_helper:
            # a thread of this process is still parked inside ncclCommInitRank: do not run RCCL / interpreter tear-down
            # around it; the line above is out (with "degraded": true), leave at once -- with a NON-ZERO status, so that a
            # driver that only looks at return codes does not book a fallen-back run as a clean RCCL run
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
