"""Condense rocprofv3 output of tools/profile_bench.sh into profiles/<tag>_*: the kernel-stats table and
the per-launch HBM traffic of the dominant kernel (FETCH_SIZE doubled: on gfx950 it reports half of a
wide coalesced read stream, MI355X_MICROARCH.md section HBM; WRITE_SIZE as is; both in KiB)."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
stats = glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, "%s_bench_n1_kernel_stats.csv" % tag))
for name in ("bench_trace.json",):
    src = os.path.join(out, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, "%s_bench_n1_under_rocprof.json" % tag))

DOMINANT = ("rq_wave_kernel", "tile_kernel_pf", "tile_kernel")


def per_launch(counter):
    vals = []
    for f in glob.glob(os.path.join(out, "pmc_" + counter, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in DOMINANT) and "RQOp" in r["Kernel_Name"]:
                vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    if not vals:
        return None, 0
    big = max(g for g, _ in vals)
    sel = [v for g, v in vals if g == big]  # the full-batch launches (the parity run uses a tiny grid)
    return sum(sel) / len(sel), len(sel)


fetch_kib, nf = per_launch("FETCH_SIZE")
write_kib, nw = per_launch("WRITE_SIZE")
summary = {"tag": tag, "kernel": "fc_rq_spline (dominant bijector kernel)",
           "FETCH_SIZE_KiB_per_launch_raw": fetch_kib, "WRITE_SIZE_KiB_per_launch_raw": write_kib,
           "launches_averaged": [nf, nw]}
if fetch_kib is not None and write_kib is not None:
    summary["hbm_read_bytes_per_launch"] = 2.0 * fetch_kib * 1024.0
    summary["hbm_write_bytes_per_launch"] = write_kib * 1024.0
    summary["traffic_bytes_per_launch"] = summary["hbm_read_bytes_per_launch"] + summary["hbm_write_bytes_per_launch"]
    summary["correction"] = "FETCH_SIZE x2 (gfx950 wide-stream under-count), WRITE_SIZE x1; separate --pmc passes"
json.dump(summary, open(os.path.join(prof, "%s_rq_spline_hbm_traffic.json" % tag), "w"), indent=1)
print(json.dumps(summary))
