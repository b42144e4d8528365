"""The BASELINE.json configurations other than the headline one, as bench.py's ``configs`` block (and on their own:
``python tools/bench_configs.py [cfg3_sample cfg1 cfg2 cfg5_shared cfg5_per_sample nsf_k10_h256 nsf_k10_h64 cfg1_sample
maf_rq_sample kernels]``, one JSON object per line).

Every entry carries what the headline line carries: throughput, the dominant kernel's ``roofline`` (algorithmic bytes or
flops of SURVEY.md 8d per launch / the average launch duration from HIP events on the launch stream), ``cpu_baseline``
(the CPU oracle on a bounded sample, host cores stated) and ``parity`` (GPU vs the oracle on the same weights / inputs).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import flowconductor_amd  # noqa: E402,F401
from flowconductor_amd import _hip, distributions, flows, ops, transforms, utils  # noqa: E402
from flowconductor_amd.nn import nets  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F16_PEAK_TFLOPS = 2500.0    # dense f16 MFMA peak, same guide (2:1-sparsity figures excluded)


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def _time_gpu(fn, steps, warmup):
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def _kernel_ms(fn, names):
    """Average launch duration (ms) of each named C-ABI entry over one call of ``fn`` (HIP events on the launch stream)."""
    timers = [ops.KernelTimer(n) for n in names]
    with torch.no_grad():
        for t in timers:
            t.__enter__()
        try:
            fn()
        finally:
            for t in reversed(timers):
                t.__exit__(None, None, None)
    torch.cuda.synchronize()
    res = {}
    for t in timers:
        d = t.durations_ms()
        res[t.name] = (sum(d) / len(d), len(d)) if d else (None, 0)
    return res


def _cpu_baseline(fn, units, what):
    """``fn()`` processes ``units`` samples on the host (the oracle): one warm-up, then repeated for ~2 s."""
    cores = _cores()
    torch.set_num_threads(cores)
    with torch.no_grad():
        fn()
        t0 = time.perf_counter()
        reps = 0
        while reps < 3 or time.perf_counter() - t0 < 2.0:
            fn()
            reps += 1
            if reps >= 200:
                break
        dt = time.perf_counter() - t0
    return {"value": units * reps / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%s, %d repeats, %.1f s" % (what, reps, dt)}


CONFIGS_TRAFFIC_PROFILE = "profiles/r04_configs_hbm_traffic.json"     # tools/profile_configs.py
_traffic_cache = {}


def _measured_traffic(config, row="main"):
    """HBM bytes per launch of a config's dominant kernel from the committed PMC passes (tools/profile_configs.py), or None:
    profile absent, row absent, or taken from a DIFFERENT library than the one loaded now (its sha256 is recorded)."""
    if "rec" not in _traffic_cache:
        from flowconductor_amd import _hip
        try:
            rec = json.load(open(os.path.join(ROOT, CONFIGS_TRAFFIC_PROFILE)))
            if rec.get("library", {}).get("sha256") != _hip.library_info()["sha256"]:
                print("[bench_configs] %s was taken from another library build: traffic left null" % CONFIGS_TRAFFIC_PROFILE,
                      file=sys.stderr)
                rec = None
        except (OSError, ValueError):
            rec = None
        _traffic_cache["rec"] = rec
    rec = _traffic_cache["rec"]
    try:
        return rec["configs"][config][row]["traffic_bytes_per_launch"]
    except (TypeError, KeyError):
        return None


KERNELS_SQ_PROFILE = "profiles/r04_kernels_sq_counters.json"          # tools/profile_configs.py --sq


def _sq_counters(row):
    if "sq" not in _traffic_cache:
        from flowconductor_amd import _hip
        try:
            rec = json.load(open(os.path.join(ROOT, KERNELS_SQ_PROFILE)))
            if rec.get("library", {}).get("sha256") != _hip.library_info()["sha256"]:
                rec = None
        except (OSError, ValueError):
            rec = None
        _traffic_cache["sq"] = rec
    rec = _traffic_cache["sq"]
    try:
        v = rec["kernels"][row]
        return {"valu_active": v["valu_active_frac_of_wave_cycles"], "wait_memory": v["wait_memory_frac_of_wave_cycles"],
                "wait_issue": v["wait_issue_frac_of_wave_cycles"], "profile": KERNELS_SQ_PROFILE}
    except (TypeError, KeyError):
        return None


def _hbm_roofline(kernel, entry, ms, launches, bytes_per_launch, note=None, bound="hbm"):
    """``bound``: the resource that binds the kernel; achieved / peak / frac are always the algorithmic HBM bytes over
    the launch time against the HBM peak (``roof``); ``traffic`` (measured HBM bytes per launch, PMC) is filled by ``run``."""
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    r = {"bound": bound, "roof": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "traffic": None, "kernel": "%s -> %s" % (entry, kernel), "launches_timed": launches, "avg_launch_ms": ms,
         "algorithmic_bytes_per_launch": bytes_per_launch}
    if note:
        r["note"] = note
    return r


def _maxdiff(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


# ---- configs[0]: README flow ----------------------------------------------------------------------------------------

def cfg1(device, steps=50, warmup=10):
    """examples/toy_2d.py / README flow: 2 x [MaskedAffineAutoregressiveTransform(2, hidden 4), RandomPermutation],
    N = 4096 -- launch-bound on a GPU (10 us of kernels behind ~50 us of Python per layer): eager and as ONE HIP graph."""
    from flowconductor_amd.utils.graphs import GraphedCall
    from oracle import torch_oracle as O

    torch.manual_seed(0)
    layers = []
    for _ in range(2):
        layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
        layers.append(transforms.RandomPermutation(features=2))
    flow_cpu = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2])).eval()
    import copy
    flow = copy.deepcopy(flow_cpu).to(device)
    n = 4096
    x = torch.randn(n, 2, device=device, generator=torch.Generator(device=device).manual_seed(1234))
    eager_s, lp = _time_gpu(lambda: flow.log_prob(x), steps, warmup)
    graphed = GraphedCall(flow.log_prob, x, clone=False)
    graph_s, lp_g = _time_gpu(lambda: graphed(x), steps, warmup)
    km = _kernel_ms(lambda: flow.log_prob(x), ["fc_affine_coupling_resnet"])
    ms, launches = km["fc_affine_coupling_resnet"]
    with torch.no_grad():
        ref = O.flow_log_prob(flow_cpu, x.cpu())
    xc = x.cpu()
    out = {"workload": "BASELINE.json configs[0]: README flow, 2 x [MAF(D=2, hidden 4), RandomPermutation], N=4096",
           "metric": "log_prob samples/sec", "unit": "samples/s", "value": n / graph_s, "ms_per_step": graph_s * 1e3,
           "eager": {"value": n / eager_s, "ms_per_step": eager_s * 1e3},
           "hip_graph": {"value": n / graph_s, "ms_per_step": graph_s * 1e3, "launches_per_replay": 1},
           "dtype": "f32",
           "roofline": _hbm_roofline("resnet_hidden_kernel (coupling tail, MAF rows)", "fc_affine_coupling_resnet", ms, launches,
                                     40 * n,
                                     "one kernel per MAF layer (pre-masked MADE + affine bijector); B = 4*d_t*(P+2)+8 = 40 B per "
                                     "sample and layer (SURVEY 8d); at N = 4096 the launch is latency-bound: 164 KB per launch "
                                     "cannot load 256 CUs", bound="launch_latency"),
           "parity": {"max_abs_dlog_prob": _maxdiff(lp, ref), "max_abs_dlog_prob_graphed": _maxdiff(lp_g, ref),
                      "rows": n},
           "cpu_baseline": _cpu_baseline(lambda: O.flow_log_prob(flow_cpu, xc), n, "4096 samples per call")}
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ---- configs[1]: 8-layer affine coupling ---------------------------------------------------------------------------

def cfg2(device, steps=20, warmup=5):
    """8 x AffineCouplingTransform, D = 32 (d_t = 16), ResidualNet(64, 2 blocks), N = 2^18."""
    import copy

    from oracle import torch_oracle as O

    torch.manual_seed(0)
    layers = [transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(32, even=(i % 2 == 0)),
                                                 lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2))
              for i in range(8)]
    flow_cpu = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([32])).eval()
    flow = copy.deepcopy(flow_cpu).to(device)
    n = 1 << 18
    x = torch.randn(n, 32, device=device, generator=torch.Generator(device=device).manual_seed(1234))
    from flowconductor_amd.utils.graphs import GraphedCall

    eager_s, lp_e = _time_gpu(lambda: flow.log_prob(x), steps, warmup)
    graphed = GraphedCall(flow.log_prob, x, clone=False)       # the 8 dependent launches + the base distribution as ONE HIP graph
    graph_s, lp_g = _time_gpu(lambda: graphed(x), steps, warmup)
    step_s = min(eager_s, graph_s)
    km = _kernel_ms(lambda: flow.log_prob(x), ["fc_affine_coupling_resnet"])
    a_ms, a_n = km["fc_affine_coupling_resnet"]
    xs = x[:2048].cpu()
    with torch.no_grad():
        z_ref, lad_ref = O.transform_apply(flow_cpu._transform, xs.clone())
        z, lad = flow._transform(x[:2048])
    sample = 1 << 15
    xc = torch.randn(sample, 32, generator=torch.Generator().manual_seed(99))
    out = {"workload": "BASELINE.json configs[1]: 8-layer affine-coupling flow, D=32, ResidualNet(64, 2 blocks), N=2^18",
           "metric": "log_prob samples/sec", "unit": "samples/s", "value": n / step_s, "ms_per_step": step_s * 1e3,
           "eager": {"value": n / eager_s, "ms_per_step": eager_s * 1e3},
           "hip_graph": {"value": n / graph_s, "ms_per_step": graph_s * 1e3, "launches_per_replay": 1,
                         "max_abs_dlog_prob_vs_eager": _maxdiff(lp_g, lp_e)},
           "dtype": "f32",
           "roofline": _hbm_roofline("resnet_hidden_kernel (coupling tail)", "fc_affine_coupling_resnet", a_ms, a_n, 264 * n,
                                     "one kernel per coupling layer (hidden stack + final Linear + affine bijector); "
                                     "B = 4*16*(2+2)+8 = 264 B per sample and layer (SURVEY 8d; the kernel itself moves "
                                     "x in + y out + logabsdet = 264 B: the parameters never exist in memory)", bound="mfma_issue"),
           "parity": {"max_rel_dsamples": float(((z.cpu().double() - z_ref.double()).abs()
                                                 / z_ref.double().abs().clamp_min(1.0)).max()),
                      "max_abs_dlogabsdet": _maxdiff(lad, lad_ref), "rows": 2048},
           "cpu_baseline": _cpu_baseline(lambda: O.flow_log_prob(flow_cpu, xc), sample, "2^15 samples per call")}
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ---- configs[4]: Sylvester flow ---------------------------------------------------------------------------------------

def _sylvester(device):
    torch.manual_seed(0)
    t = transforms.SylvesterTransform(features=128, num_householder=32, device="cpu").eval()
    with torch.no_grad():
        for p in t.parameters():
            p.add_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.1)
    return t


def cfg5_shared(device, steps=20, warmup=5):
    """SylvesterTransform(D = 128, 32 Householder vectors), batch-shared parameters, N = 2^18: with batch-independent
    weights the Householder / triangular chains fold into two dense [128, 128] maps -- a true dense contraction, the
    one place the path belongs on the matrix cores."""
    import copy

    from oracle import torch_oracle as O

    t_cpu = _sylvester(device)
    t = copy.deepcopy(t_cpu).to(device)
    n, d = 1 << 18, 128
    x = torch.randn(n, d, device=device, generator=torch.Generator(device=device).manual_seed(1234))
    step_s, _ = _time_gpu(lambda: t(x), steps, warmup)
    km = _kernel_ms(lambda: t(x), ["fc_sylvester_mm"])
    ms, launches = km["fc_sylvester_mm"]
    flops = 8.0 * d * d * n                 # 4 mat-vecs per sample (two remain after folding Q into R)
    executed = 3.0 * 4.0 * d * d * n        # two [128,128] products, three split-f16 terms each
    with torch.no_grad():
        y_ref, lad_ref = O.transform_apply(t_cpu, x[:1024].cpu())
        y, lad = t(x[:1024])
    sample = 1 << 13
    xc = torch.randn(sample, d, generator=torch.Generator().manual_seed(99))
    tf = flops / (ms * 1e-3) / 1e12
    out = {"workload": "BASELINE.json configs[4]: SylvesterTransform D=128, M=32 Householder vectors, shared parameters, "
                       "N=2^18 (forward + logabsdet)",
           "metric": "transform samples/sec", "unit": "samples/s", "value": n / step_s, "ms_per_step": step_s * 1e3,
           "dtype": "f32 results; products = 3-term split-f16 MFMA (v_mfma_f32_16x16x32_f16)",
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                        "frac": tf / (MFMA_F16_PEAK_TFLOPS / 3.0), "traffic": None,
                        "kernel": "fc_sylvester_mm -> fc::sylvester_mm_kernel<4, false>", "launches_timed": launches,
                        "avg_launch_ms": ms, "algorithmic_flops_per_launch": flops,
                        "executed_f16_flops_per_launch": executed,
                        "note": "peak = dense f16 MFMA peak / 3 (an f32 product costs three f16 terms); algorithmic "
                                "flops = SURVEY 8d's 8 D^2 per sample; HBM side: %d B/sample = %.0f GB/s"
                                % (8 * d + 4, (8 * d + 4) * n / (ms * 1e-3) / 1e9)},
           "parity": {"max_abs_doutputs": _maxdiff(y, y_ref), "max_abs_dlogabsdet": _maxdiff(lad, lad_ref), "rows": 1024},
           "cpu_baseline": _cpu_baseline(lambda: O.transform_apply(t_cpu, xc), sample, "2^13 samples per call")}
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def cfg5_per_sample(device, steps=5, warmup=2, log2n=18):
    """The conditional form of configs[4]: a hyper-network's per-sample q [N, 32, 128], R1 / R2 [N, 128, 128],
    bias [N, 128] (148 480 B of parameters per sample) -- HBM-bound by construction."""
    n, d, m = 1 << log2n, 128, 32
    gen = torch.Generator(device=device).manual_seed(1234)
    x = torch.randn(n, d, device=device, generator=gen)
    q = torch.randn(n, m, d, device=device, generator=gen)
    r1 = torch.randn(n, d, d, device=device, generator=gen).mul_(d ** -0.5).triu_()
    r2 = torch.randn(n, d, d, device=device, generator=gen).mul_(d ** -0.5).triu_()
    r1.diagonal(dim1=1, dim2=2).tanh_()
    r2.diagonal(dim1=1, dim2=2).tanh_()
    b = torch.randn(n, d, device=device, generator=gen) * 0.1
    step_s, _ = _time_gpu(lambda: ops.sylvester(x, q, r1, r2, b), steps, warmup)
    km = _kernel_ms(lambda: ops.sylvester(x, q, r1, r2, b), ["fc_sylvester"])
    ms, launches = km["fc_sylvester"]
    byts = 4 * (d * (d + 1) + 2 * d + m * d) * n      # upper triangles of R1 / R2 (diagonal included), x in, y out, q
    byts_survey = 4 * (2 * d * d + 2 * d + m * d) * n
    # float64 formula on 256 rows (planar.py:144-166 with per-sample parameters; the reference's own conditional class
    # only runs for D = 2, SURVEY headline facts: this leg is "parity unpinned" by the reference)
    k = 256
    xs, qs, r1s, r2s, bs = (t[:k].double().cpu() for t in (x, q, r1, r2, b))

    def refl(v, qq, reverse):
        order = range(m - 1, -1, -1) if reverse else range(m)
        for i in order:
            qi = qq[:, i]
            v = v - (v * qi).sum(-1, keepdim=True) * (2.0 / (qi * qi).sum(-1, keepdim=True)) * qi
        return v

    def f64_formula():
        qtz = refl(xs, qs, True)
        pre = torch.einsum("nij,nj->ni", r1s, qtz) + bs
        act = torch.tanh(pre)
        out = xs + refl(torch.einsum("nij,nj->ni", r2s, act), qs, False)
        diag = 1 + (1 - act ** 2) * (torch.diagonal(r1s, dim1=1, dim2=2) * torch.diagonal(r2s, dim1=1, dim2=2))
        return out, torch.log(diag).sum(-1)

    y64, lad64 = f64_formula()
    with torch.no_grad():
        y, lad = ops.sylvester(x[:k], q[:k], r1[:k], r2[:k], b[:k])
    out = {"workload": "BASELINE.json configs[4], conditional form: per-sample q [N,32,128], R1/R2 [N,128,128], N=2^%d"
                       % log2n,
           "metric": "transform samples/sec", "unit": "samples/s", "value": n / step_s, "ms_per_step": step_s * 1e3,
           "dtype": "f32",
           "roofline": _hbm_roofline("fc::sylvester_kernel (per-sample parameters)", "fc_sylvester", ms, launches, byts,
                                     "bytes the algorithm needs: the upper triangles of R1 / R2, 4*(D(D+1) + 2 D + M D) = 83 456 B per "
                                     "sample (the hyper-network emits full row-major [D, D] matrices; their lower triangles are zero "
                                     "and never read).  SURVEY 8d's figure prices the full matrices: 148 480 B per sample -> "
                                     "`survey_accounting`"),
           "parity": {"max_abs_doutputs_vs_f64_formula": _maxdiff(y, y64),
                      "max_abs_dlogabsdet_vs_f64_formula": _maxdiff(lad, lad64), "rows": k,
                      "note": "parity unpinned by the reference (its conditional Sylvester class runs for D = 2 only)"},
           "cpu_baseline": _cpu_baseline(f64_formula, k, "the float64 torch formula on 256 samples per call")}
    out["roofline"]["survey_accounting"] = {"bytes_per_launch": byts_survey, "achieved": byts_survey / (ms * 1e-3) / 1e9,
                                            "frac": byts_survey / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out["cpu_baseline"]["kind"] = "port (float64 formula: the reference class cannot run at D = 128)"
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    del q, r1, r2
    torch.cuda.empty_cache()
    return out


# ---- the reference's DEFAULT NSF layer shape ---------------------------------------------------------------------------

def nsf_k10_h256(device, steps=5, warmup=2, log2n=18, layers=16, hidden=256, bins=10):
    """A 16-layer RQ-NSF coupling flow with the reference's default bin count (coupling.py:507: num_bins=10) and a
    256-wide conditioner (nn/nets/resnet.py:62), D = 64, linear tails -- the general fused path."""
    import copy

    from oracle import torch_oracle as O

    torch.manual_seed(0)
    stack = [transforms.PiecewiseRationalQuadraticCouplingTransform(
        utils.create_alternating_binary_mask(64, even=(i % 2 == 0)),
        lambda a, b: nets.ResidualNet(a, b, hidden_features=hidden, num_blocks=2), num_bins=bins, tails="linear",
        tail_bound=3.0) for i in range(layers)]
    flow_cpu = flows.Flow(transforms.CompositeTransform(stack), distributions.StandardNormal([64])).eval()
    flow = copy.deepcopy(flow_cpu).to(device)
    n = 1 << log2n
    x = torch.randn(n, 64, device=device, generator=torch.Generator(device=device).manual_seed(1234))
    step_s, _ = _time_gpu(lambda: flow.log_prob(x), steps, warmup)
    hidden_entry = "fc_resnet_hidden" if hidden == 64 else "fc_resnet_hidden_wide"
    names = ["fc_rq_spline_fused_general", hidden_entry, "fc_rq_spline"]
    km = _kernel_ms(lambda: flow.log_prob(x), names)
    p = 3 * bins - 1
    out = {"workload": "%d-layer RQ-NSF coupling flow, D=64, K=%d (reference default), linear tails, "
                       "ResidualNet(%d, 2 blocks), N=2^%d" % (layers, bins, hidden, log2n),
           "metric": "log_prob samples/sec", "unit": "samples/s", "value": n / step_s, "ms_per_step": step_s * 1e3,
           "dtype": "f32 results; conditioner products = 3-term split-f16 MFMA"}
    f_ms, f_n = km["fc_rq_spline_fused_general"]
    if f_ms and hidden == 64:
        # resident weights: VALU-issue bound like the K = 8 kernel; bytes as in the headline line: what the fused kernel
        # moves (h + x in, y + logabsdet out = 772 B per sample), not SURVEY 8d's unfused 4 d_t (P + 2) + 8
        out["roofline"] = _hbm_roofline("fc::f4k%d::rq_fused_linear_kernel4 (resident weights, one accumulator set)" % bins,
                                        "fc_rq_spline_fused_general", f_ms, f_n, (4 * 64 + 8 * 64 + 4) * n, bound="valu_issue")
    elif f_ms:
        flops = 2.0 * hidden * 32 * p * n
        tf = flops / (f_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                           "frac": tf / (MFMA_F16_PEAK_TFLOPS / 3.0), "traffic": None,
                           "kernel": "fc_rq_spline_fused_general", "launches_timed": f_n, "avg_launch_ms": f_ms,
                           "algorithmic_flops_per_launch": flops,
                           "hbm_side": {"bytes_per_sample": 4 * hidden + 8 * 64 + 4,
                                        "GBs": (4 * hidden + 8 * 64 + 4) * n / (f_ms * 1e-3) / 1e9}}
    else:
        s_ms, s_n = km["fc_rq_spline"]
        if s_ms:
            out["roofline"] = _hbm_roofline("rq spline kernel (unfused path)", "fc_rq_spline", s_ms, s_n,
                                            (4 * 32 * (p + 2) + 8) * n)
    h_ms, h_n = km[hidden_entry]
    if h_ms and hidden == 64:
        out["roofline_hidden"] = _hbm_roofline("fc::resnet_hidden_kernel", "fc_resnet_hidden", h_ms, h_n, (4 * 64 + 4 * 64) * n,
                                               bound="mfma_issue")
    elif h_ms:
        hflops = 2.0 * (hidden * 32 + 4 * hidden * hidden) * n
        out["roofline_hidden"] = {"bound": "mfma", "achieved": hflops / (h_ms * 1e-3) / 1e12,
                                  "peak": MFMA_F16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s",
                                  "frac": hflops / (h_ms * 1e-3) / 1e12 / (MFMA_F16_PEAK_TFLOPS / 3.0),
                                  "kernel": "fc_resnet_hidden_wide", "launches_timed": h_n, "avg_launch_ms": h_ms}
    xs = x[:1024].cpu()
    with torch.no_grad():
        z_ref, lad_ref = O.transform_apply(flow_cpu._transform, xs.clone())
        z, lad = flow._transform(x[:1024])
    out["parity"] = {"max_rel_dsamples": float(((z.cpu().double() - z_ref.double()).abs()
                                                / z_ref.double().abs().clamp_min(1.0)).max()),
                     "max_abs_dlogabsdet": _maxdiff(lad, lad_ref), "rows": 1024}
    sample = 1 << 13
    xc = torch.randn(sample, 64, generator=torch.Generator().manual_seed(99))
    out["cpu_baseline"] = _cpu_baseline(lambda: O.flow_log_prob(flow_cpu, xc), sample, "2^13 samples per call")
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def nsf_k10_h64(device):
    """BASELINE.json configs[2] with the reference's DEFAULT bin count: 32 layers, D = 64, ResidualNet(64, 2 blocks),
    K = 10 (coupling.py:507), N = 2^20 -- the fused layer on the resident-weight K = 10 kernel (fc_rq_fused4_body.h)."""
    return nsf_k10_h256(device, steps=5, warmup=3, log2n=20, layers=32, hidden=64, bins=10)


def cfg3_sample(device, steps=8, warmup=5, log2n=20):      # (warm-up: the caching allocator settles after ~4 calls)
    """The INVERSE direction of the headline flow (north star: "forward + inverse"): ``Flow.sample(2^20)`` and
    ``Flow.sample_and_log_prob(2^20)`` of BASELINE.json configs[2]'s 32-layer RQ-NSF flow (flows/base.py:50-105 of the
    reference) -- base draws on the device, 32 coupling layers inverted last to first in the same fused kernels
    instantiated with the inverse spline (root of the rational-quadratic bin, rational_quadratic.py:133-160)."""
    import copy

    import bench
    from oracle import torch_oracle as O

    flow_cpu = bench.build_flow()
    flow = copy.deepcopy(flow_cpu).to(device).eval()
    n = 1 << log2n
    torch.manual_seed(4321)
    step_s, x = _time_gpu(lambda: flow.sample(n), steps, warmup)
    step_lp_s, (x2, lp2) = _time_gpu(lambda: flow.sample_and_log_prob(n), steps, warmup)
    km = _kernel_ms(lambda: flow.sample(n), ["fc_rq_spline_fused_linear", "fc_resnet_hidden"])
    out = {"workload": "Flow.sample / sample_and_log_prob, 32-layer RQ-NSF coupling flow D=64 K=8 (BASELINE.json configs[2]), "
                       "N=2^%d draws" % log2n,
           "metric": "samples drawn per second", "unit": "samples/s", "value": n / step_s, "ms_per_step": step_s * 1e3,
           "sample_and_log_prob": {"value": n / step_lp_s, "unit": "samples/s", "ms_per_step": step_lp_s * 1e3},
           "dtype": "f32 results; conditioner products = 3-term split-f16 MFMA"}
    f_ms, f_n = km["fc_rq_spline_fused_linear"]
    if f_ms:
        out["roofline"] = _hbm_roofline("fc::rq_fused_linear_kernel3<true (inverse), 64, 2, true, true>", "fc_rq_spline_fused_linear",
                                        f_ms, f_n, (4 * 64 + 8 * 64 + 4) * n,
                                        note="binding resource: VALU issue (the inverse spline adds a square root and a "
                                             "division per element to the forward evaluation)", bound="valu_issue")
    h_ms, h_n = km["fc_resnet_hidden"]
    if h_ms:
        out["roofline_hidden"] = _hbm_roofline("fc::resnet_hidden_kernel", "fc_resnet_hidden", h_ms, h_n, (4 * 64 + 4 * 64) * n,
                                               bound="mfma_issue")
    with torch.no_grad():
        # draws pushed forward again land on the noise they came from; densities agree with log_prob of the draws
        z_back, lad = flow._transform(x2)
        lp_fwd = flow.log_prob(x2)
        z0 = torch.randn(2048, 64, generator=torch.Generator().manual_seed(5))
        x_ref, lad_ref = O.transform_apply(flow_cpu._transform, z0.clone(), inverse=True)
        x_gpu, lad_gpu = flow._transform.inverse(z0.to(device))
        x64, lad64 = O.transform_apply(copy.deepcopy(flow_cpu._transform).double(), z0.double(), inverse=True)

    def rel(a, b):
        return float(((a.double().cpu() - b.double()).abs() / b.double().abs().clamp_min(1.0)).max())

    out["parity"] = {"max_abs_log_prob_sample_vs_forward": _maxdiff(lp2, lp_fwd),
                     "finite": bool(torch.isfinite(x2).all() and torch.isfinite(lp2).all()),
                     "inverse_vs_oracle_rows": 2048,
                     "max_rel_dsamples": rel(x_gpu, x_ref), "max_abs_dlogabsdet": _maxdiff(lad_gpu, lad_ref),
                     "vs_float64": {"gpu_max_rel_dsamples": rel(x_gpu, x64), "cpu_f32_oracle_max_rel_dsamples": rel(x_ref, x64),
                                    "gpu_max_abs_dlogabsdet": _maxdiff(lad_gpu, lad64),
                                    "cpu_f32_oracle_max_abs_dlogabsdet": _maxdiff(lad_ref, lad64)}}
    del z_back, lad
    sample = 1 << 13
    zc = torch.randn(sample, 64, generator=torch.Generator().manual_seed(99))
    out["cpu_baseline"] = _cpu_baseline(lambda: O.transform_apply(flow_cpu._transform, zc.clone(), inverse=True), sample,
                                        "inverse of the 32-layer stack on 2^13 noise rows per call (the oracle)")
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out




# ---- sampling through autoregressive layers (fc_made_inverse: the D passes inside one kernel) --------------------------

def _ar_sample(device, build, d, n, workload, steps=20, warmup=5, cpu_rows=4096):
    import copy
    from oracle import torch_oracle as O

    torch.manual_seed(0)
    flow_cpu = build().eval()
    flow = copy.deepcopy(flow_cpu).to(device)
    step_s, _ = _time_gpu(lambda: flow.sample(n), steps, warmup)
    km = _kernel_ms(lambda: flow.sample(n), ["fc_made_inverse"])
    ms, launches = km["fc_made_inverse"]
    z = torch.randn(1024, d, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        ref, ref_lad = O.transform_apply(flow_cpu._transform, z.clone(), inverse=True)
        ref64, _ = O.transform_apply(copy.deepcopy(flow_cpu._transform).double(), z.double(), inverse=True)
        got, got_lad = flow._transform.inverse(z.to(device))
    zc = torch.randn(cpu_rows, d, generator=torch.Generator().manual_seed(8))
    out = {"workload": workload, "metric": "sample samples/sec", "unit": "samples/s", "value": n / step_s,
           "ms_per_step": step_s * 1e3, "dtype": "f32",
           "roofline": _hbm_roofline("fc::made_inverse_kernel", "fc_made_inverse", ms, launches, (8 * d + 4) * n,
                                     "z row in, x row out, logabsdet: 8 D + 4 B per sample and layer; the D sequential conditioner "
                                     "passes (autoregressive.py:44-53) run inside the kernel on LDS-resident pre-masked weights: "
                                     "bound by the serial chain of D x 5 layers, not by HBM", bound="valu_issue"),
           "parity": {"max_abs_dsamples": _maxdiff(got, ref), "max_abs_dlogabsdet": _maxdiff(got_lad, ref_lad),
                      "max_abs_dsamples_vs_float64": _maxdiff(got, ref64), "f32_oracle_vs_float64": _maxdiff(ref, ref64),
                      "rows": 1024},
           "cpu_baseline": _cpu_baseline(lambda: O.transform_apply(flow_cpu._transform, zc.clone(), inverse=True), cpu_rows,
                                         "inverse of the stack on %d noise rows per call (the oracle's D full passes)" % cpu_rows)}
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def cfg1_sample(device):
    """BASELINE.json configs[0] in the sampling direction: Flow.sample(4096) of the README flow."""
    def build():
        layers = []
        for _ in range(2):
            layers.append(transforms.MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
            layers.append(transforms.RandomPermutation(features=2))
        return flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([2]))
    return _ar_sample(device, build, 2, 4096, "BASELINE.json configs[0], sampling: README flow, Flow.sample(4096)", steps=50, warmup=10)


def maf_rq_sample(device, d=8, log2n=20):
    """Sampling through RQ-spline autoregressive layers (autoregressive.py:529-621): 5 x [MaskedPiecewiseRationalQuadratic
    AutoregressiveTransform(D = 8, hidden 64, K = 8, linear tails), ReversePermutation], Flow.sample(2^20)."""
    def build():
        layers = []
        for _ in range(5):
            layers.append(transforms.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                d, 64, num_bins=8, tails="linear", tail_bound=3.0, num_blocks=2))
            layers.append(transforms.ReversePermutation(features=d))
        return flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([d]))
    return _ar_sample(device, build, d, 1 << log2n,
                      "5 x [RQ-spline MAF(D=%d, hidden 64, K=8, linear tails), ReversePermutation], Flow.sample(2^%d)" % (d, log2n),
                      steps=5, warmup=2, cpu_rows=2048)


# ---- one roofline row per remaining kernel family -----------------------------------------------------------------------

def kernels(device):
    """Stand-alone bijector kernels that the headline flow does not launch: each with its algorithmic bytes per unit
    (B = 4 d_t (P + 2) + 8 for a coupling bijector; rows in + out for the others), the average launch duration
    (HIP events), the fraction of the HBM peak, and parity of 1 024 rows against the CPU oracle."""
    import copy
    import types
    from oracle import torch_oracle as O

    rows = {}
    gen = torch.Generator(device=device).manual_seed(1234)

    def add(name, entry, kernel, fn, bytes_per_launch, parity, note=None, bound="hbm", reps=5):
        with torch.no_grad():
            for _ in range(2):
                fn()
        ms_list = []
        for _ in range(reps):
            ms, launches = _kernel_ms(fn, [entry])[entry]
            ms_list.append(ms)
        ms_list.sort()
        r = _hbm_roofline(kernel, entry, ms_list[len(ms_list) // 2], launches, bytes_per_launch, note, bound)
        r["parity"] = parity
        rows[name] = r

    def module_parity(t_gpu, x, inverse=False):
        t_cpu = copy.deepcopy(t_gpu).cpu()
        xs = x[:1024]
        with torch.no_grad():
            ref, ref_lad = O.transform_apply(t_cpu, xs.cpu().clone(), inverse=inverse)
            got, got_lad = (t_gpu.inverse if inverse else t_gpu)(xs)
        return {"max_abs_doutputs": _maxdiff(got, ref), "max_abs_dlogabsdet": _maxdiff(got_lad, ref_lad), "rows": 1024}

    def conditioner(i, o):
        return nets.ResidualNet(i, o, hidden_features=64, num_blocks=2)

    n18, n20, d = 1 << 18, 1 << 20, 64
    x18 = torch.randn(n18, d, device=device, generator=gen)
    mask = utils.create_alternating_binary_mask(d, even=True)
    # sibling splines as coupling bijectors (K = 8): P = K / 2K - 1 / 2K + 2 parameters per transformed dim
    torch.manual_seed(0)
    for name, cls, p in (("linear_spline_coupling", transforms.PiecewiseLinearCouplingTransform, 8),
                         ("quadratic_spline_coupling", transforms.PiecewiseQuadraticCouplingTransform, 15),
                         ("cubic_spline_coupling", transforms.PiecewiseCubicCouplingTransform, 18)):
        kw = dict(num_bins=8, tails="linear", tail_bound=3.0)
        t = cls(mask, conditioner, **kw).to(device).eval()
        add(name, "fc_piecewise_spline", "fc::spline tile kernel (%s)" % name.split("_")[0], lambda t=t: t(x18),
            (4 * 32 * (p + 2) + 8) * n18, module_parity(t, x18))
        del t
    # sum of sigmoids, S = 30, D = 8 all transformed, per-sample parameters: forward and numerical inverse
    ns, dsos = 30, 8
    xs = torch.randn(n18, dsos, device=device, generator=gen) * 2.0
    prm = torch.randn(n18, dsos * (3 * ns + 1), device=device, generator=gen)
    fake = types.SimpleNamespace(n_sigmoids=ns, features=dsos)
    with torch.no_grad():
        y_f, l_f = ops.sum_of_sigmoids(xs[:1024], prm[:1024], ns, offset=0.5)
        r_f, rl_f = O._ew_sos_ar(fake, xs[:1024].cpu() , prm[:1024].cpu(), False)
        y_i, l_i = ops.sum_of_sigmoids(y_f, prm[:1024], ns, inverse=True, offset=0.5)
    sos_bytes = (4 * dsos * (3 * ns + 3) + 8) * n18
    add("sum_of_sigmoids_forward", "fc_sum_of_sigmoids", "fc::sos tile kernel", lambda: ops.sum_of_sigmoids(xs, prm, ns, offset=0.5),
        sos_bytes, {"max_abs_doutputs": _maxdiff(y_f, r_f), "max_abs_dlogabsdet": _maxdiff(l_f, rl_f.sum(-1) if rl_f.dim() > 1 else rl_f), "rows": 1024})
    add("sum_of_sigmoids_inverse", "fc_sum_of_sigmoids", "fc::sos tile kernel (bracket + safeguarded Newton)",
        lambda: ops.sum_of_sigmoids(xs, prm, ns, inverse=True, offset=0.5), sos_bytes,
        {"round_trip_max_abs": _maxdiff(y_i, xs[:1024]), "max_abs_logabsdet_sum": float((l_f + l_i).abs().max()), "rows": 1024},
        "~10 evaluations of the S = 30 sigmoids per element: bound by vector issue, not by HBM", bound="valu_issue")
    del xs, prm
    # LU linear, D = 64: forward (folded into one dense matrix on the matrix cores for wide batches) and inverse
    torch.manual_seed(1)
    x20 = torch.randn(n20, d, device=device, generator=gen)
    lu = transforms.LULinear(d).to(device).eval()
    with torch.no_grad():
        lu.lower_entries.normal_(0, 0.1)
        lu.upper_entries.normal_(0, 0.1)
        lu.bias.normal_(0, 0.1)
    add("lu_linear_forward", "fc_dense_mm", "fc::sylvester_mm_kernel<2> (fc_dense_mm: L U folded, split-f16 MFMA)", lambda: lu(x20), (8 * d + 4) * n20,
        module_parity(lu, x20), "8 D B per sample; 2 D^2 flop per sample run on the matrix cores")
    add("lu_linear_inverse", "fc_dense_mm", "fc::sylvester_mm_kernel<2> (fc_dense_mm: U^-1 L^-1 folded in float64, split-f16 MFMA)", lambda: lu.inverse(x20),
        (8 * d + 4) * n20, module_parity(lu, x20, inverse=True))
    # shared Householder sequence D = 128, K = 32 (folded into one orthogonal matrix on the matrix cores)
    x128 = torch.randn(n18, 128, device=device, generator=gen)
    hh = transforms.HouseholderSequence(features=128, num_transforms=32).to(device).eval()
    with torch.no_grad():
        hh.q_vectors.normal_()
    add("householder_shared", "fc_dense_mm", "fc::sylvester_mm_kernel<4> (fc_dense_mm: 32 reflections folded)", lambda: hh(x128), (8 * 128) * n18,
        module_parity(hh, x128))
    del x128
    # planar, D = 64
    pl = transforms.PlanarTransform(features=d).to(device).eval()
    add("planar", "fc_planar", "fc::planar row-wave kernel", lambda: pl(x20), (8 * d + 4) * n20, module_parity(pl, x20))
    # permutation, D = 64 (bit-exact)
    perm = transforms.RandomPermutation(features=d).to(device)
    pp = module_parity(perm, x20)
    pp["bit_exact"] = pp["max_abs_doutputs"] == 0.0
    add("permutation", "fc_permute", "fc::permute kernel", lambda: perm(x20), (8 * d) * n20, pp)
    # element-wise family: tanh (forward), D = 64
    th = transforms.Tanh().to(device)
    add("elementwise_tanh", "fc_elementwise", "fc::elementwise kernel", lambda: th(x20), (8 * d + 4) * n20, module_parity(th, x20))
    # standard normal log-prob, D = 64
    sn = distributions.StandardNormal([d]).to(device)
    with torch.no_grad():
        got = sn.log_prob(x20[:1024])
        ref = O.standard_normal_log_prob(x20[:1024].cpu())
    add("standard_normal_log_prob", "fc_standard_normal_log_prob", "fc::std_normal kernel", lambda: sn.log_prob(x20), (4 * d + 4) * n20,
        {"max_abs_dlog_prob": _maxdiff(got, ref), "rows": 1024})
    del x20
    # RQ spline backward (stand-alone bijector, parameters from HBM): cfg-3 layer shape
    k, d_t = 8, 32
    p = 3 * k - 1
    nb = 1 << 19
    xb = torch.randn(nb, d, device=device, generator=gen) * 1.5
    prm = torch.randn(nb, d_t * p, device=device, generator=gen)
    cols = torch.arange(0, d, 2, dtype=torch.int32, device=device)
    gy, gl = torch.randn(nb, d, device=device, generator=gen), torch.randn(nb, device=device, generator=gen)
    kw = dict(num_bins=k, tails="linear", tail_bound=3.0, wh_divisor=8.0)
    rows_p = 256
    x64 = xb[:rows_p].double().cpu().requires_grad_(True)
    p64 = prm[:rows_p].double().cpu().requires_grad_(True)
    out64, lad64 = O.rq_from_rows(x64[:, cols.long().cpu()], p64.view(rows_p, d_t, p).clone(), k, "linear", 3.0, False, wh_divisor=8.0)
    y64 = x64.clone().index_copy(1, cols.long().cpu(), out64)
    loss = (y64 * gy[:rows_p].double().cpu()).sum() + (lad64.sum(dim=1) * gl[:rows_p].double().cpu()).sum()
    gx_ref, gp_ref = torch.autograd.grad(loss, (x64, p64))
    gx, gp = ops.rq_spline_backward(xb[:rows_p], prm[:rows_p], cols, gy[:rows_p], gl[:rows_p], **kw)
    add("rq_spline_backward", "fc_rq_spline_backward", "fc::rq_backward wave kernel",
        lambda: ops.rq_spline_backward(xb, prm, cols, gy, gl, **kw), (2 * 4 * d_t * p + 12 * d_t + 4) * nb,
        {"max_rel_dgrad_inputs_vs_f64_autograd": _maxdiff(gx, gx_ref) / max(1e-30, float(gx_ref.abs().max())),
         "max_rel_dgrad_params_vs_f64_autograd": _maxdiff(gp, gp_ref) / max(1e-30, float(gp_ref.abs().max())), "rows": rows_p})
    return {"workload": "stand-alone bijector kernels (SURVEY 8a rows S4, G1-G3, U1, H1, P1, M1, E1, F2, and the spline backward)",
            "kernels": rows}


ALL = {"cfg3_sample": cfg3_sample, "cfg1": cfg1, "cfg2": cfg2, "cfg5_shared": cfg5_shared, "cfg5_per_sample": cfg5_per_sample,
       "nsf_k10_h256": nsf_k10_h256, "nsf_k10_h64": nsf_k10_h64, "cfg1_sample": cfg1_sample, "maf_rq_sample": maf_rq_sample,
       "kernels": kernels}


def run(device, which=None, log=None):
    res = {}
    for name in (which or list(ALL)):
        t0 = time.perf_counter()
        try:
            res[name] = ALL[name](device)
            # measured HBM traffic of the dominant kernel(s), from the committed PMC passes of this same command
            rl = res[name].get("roofline")
            if isinstance(rl, dict) and rl.get("traffic") is None:
                rl["traffic"] = _measured_traffic(name)
                if rl["traffic"] is not None:
                    rl["traffic_profile"] = CONFIGS_TRAFFIC_PROFILE
            for row, r in (res[name].get("kernels") or {}).items():
                r["traffic"] = _measured_traffic(name, row)
                if r["traffic"] is not None:
                    r["traffic_profile"] = CONFIGS_TRAFFIC_PROFILE
                c = _sq_counters(row)
                if c is not None:      # what a row below the HBM roof is busy with (fractions of its wave cycles)
                    r["sq_counters"] = c
        except Exception as e:      # a secondary block must never take the headline line down with it
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if log:
            log("config %s done in %.1f s" % (name, time.perf_counter() - t0))
    return res


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for k, v in run(dev, sys.argv[1:] or None, log=lambda m: print("[bench_configs] " + m, file=sys.stderr)).items():
        print(json.dumps({k: v, "library": _hip.library_info()}))
