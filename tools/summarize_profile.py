"""Condense rocprofv3 output of tools/profile_bench.sh into profiles/<tag>_*: the kernel-stats table and
the per-launch HBM traffic of the hot kernels (FETCH_SIZE doubled: on gfx950 it reports half of a
wide coalesced read stream, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is; both in KiB)."""
import csv
import glob
import json
import os
import shutil
import sys

import hashlib

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBRARY = {"path": "flowconductor_amd/csrc/libflowcon_hip.so",
           "sha256": hashlib.sha256(open(os.path.join(root, "flowconductor_amd", "csrc", "libflowcon_hip.so"), "rb").read()).hexdigest()}
prof = os.path.join(root, "profiles")
stats = glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, "%s_bench_n1_kernel_stats.csv" % tag))
for name in ("bench_trace.json",):
    src = os.path.join(out, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, "%s_bench_n1_under_rocprof.json" % tag))

# C-ABI entry -> substring of the kernel symbol it launches in bench.py's flow
KERNELS = {
    "fc_rq_spline_fused_linear": "rq_fused_linear_kernel3",
    "fc_resnet_hidden": "resnet_hidden_kernel",
    "fc_rq_spline": "rq_wave_kernel",          # only in the extra FC_FUSED=0 pass
}
EXPECTED = {
    "fc_rq_spline_fused_linear": "reads h [2^20,64] + x [2^20,64] = 536.9 MB; writes y [2^20,64] + logabsdet = 272.6 MB; "
                                 "the [2^20,736] parameter tensor (3.09 GB written + read back when unfused) never reaches HBM",
    "fc_resnet_hidden": "reads x [2^20,64] = 268.4 MB; writes h [2^20,64] = 268.4 MB",
    "fc_rq_spline": "reads params [2^20,736] + x = 3.36 GB; writes y + logabsdet = 272.6 MB",
}


def per_launch(counter, key):
    vals = []
    for f in glob.glob(os.path.join(out, "pmc_" + counter, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and key in r["Kernel_Name"]:
                vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    if not vals:
        return None, 0
    big = max(g for g, _ in vals)
    sel = [v for g, v in vals if g == big]  # the full-batch launches (the parity run uses a tiny grid)
    return sum(sel) / len(sel), len(sel)


summary = {"tag": tag, "library": LIBRARY,
           "source": "tools/profile_bench.sh %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of "
                     "bench.py --steps 3 --warmup 3, per-launch means over the full-batch (2^20-row) launches; "
                     "FETCH_SIZE x2 (gfx950 wide-stream correction), KiB -> bytes" % tag}
for entry, key in KERNELS.items():
    fetch_kib, nf = per_launch("FETCH_SIZE", key)
    write_kib, nw = per_launch("WRITE_SIZE", key)
    if fetch_kib is None or write_kib is None:
        continue
    rd, wr = 2.0 * fetch_kib * 1024.0, write_kib * 1024.0
    summary[entry] = {"kernel": key, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
                      "launches_averaged": [nf, nw], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                      "traffic_bytes_per_launch": rd + wr, "expected": EXPECTED[entry]}
json.dump(summary, open(os.path.join(prof, "%s_hbm_traffic.json" % tag), "w"), indent=1)
print(json.dumps(summary))

# Per-step launch durations of the two hot kernels from the kernel trace (full-batch launches in launch order, 32
# per pass through the flow).  `--stats` averages every call of a kernel -- warm-up steps and the 2 048-row parity
# launches included -- while bench.py brackets the launches of the LAST timed step with HIP events: this table is the
# like-for-like cross-check of `roofline.avg_launch_ms` in the bench line of the same run.
traces = glob.glob(os.path.join(out, "trace", "*", "*kernel_trace.csv"))
if traces:
    rows = list(csv.DictReader(open(traces[0])))
    per_step = {"library": LIBRARY, "source": "rocprofv3 --kernel-trace of the profiled bench.py run (%s_bench_n1_under_rocprof.json); "
                          "microseconds; passes in launch order: warm-up steps, timed steps (the last one carries "
                          "bench.py's HIP-event pairs), then the untimed FC_FUSED=0 pass (hidden kernel only)" % tag}
    for entry in ("fc_rq_spline_fused_linear", "fc_resnet_hidden"):
        sel = [r for r in rows if KERNELS[entry] in r["Kernel_Name"]]
        sel.sort(key=lambda r: int(r["Start_Timestamp"]))
        big = max(int(r["Grid_Size_X"]) for r in sel)
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel if int(r["Grid_Size_X"]) == big]
        per_step[entry] = [{"pass": i // 32, "launches": len(dur[i:i + 32]),
                            "avg_us": round(sum(dur[i:i + 32]) / len(dur[i:i + 32]), 1),
                            "min_us": round(min(dur[i:i + 32]), 1), "max_us": round(max(dur[i:i + 32]), 1)}
                           for i in range(0, len(dur), 32)]
    json.dump(per_step, open(os.path.join(prof, "%s_per_step_launch_us.json" % tag), "w"), indent=1)
    print(json.dumps(per_step))
