"""HBM traffic of the dominant kernel of every `configs` entry of bench.py (and of every row of `configs.kernels`), from
rocprofv3 PMC passes on the GPU box:  python3 tools/profile_configs.py <tag> [config ...]

One process per (config, counter): `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/bench_configs.py <config>`
(separate passes per counter; FETCH_SIZE doubled -- on gfx950 it reports half of a wide coalesced read stream,
MI355X_MICROARCH.md section HBM; both counters in KiB).  Per kernel symbol the launches of the LARGEST grid are averaged
(the parity launches use small grids).  Writes profiles/<tag>_configs_hbm_traffic.json with the sha256 of the library the
kernels came from; tools/bench_configs.py fills `roofline.traffic` from it when that sha256 is the loaded library's.
This driver never touches the GPU itself (no torch import)."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# config -> {row: substring of the kernel symbol}; "main" is the entry's own `roofline`
KERNELS = {
    "cfg1": {"main": "resnet_hidden_kernel"},
    "cfg2": {"main": "resnet_hidden_kernel"},
    "cfg3_sample": {"main": "rq_fused_linear_kernel3<true"},
    "cfg5_shared": {"main": "sylvester_mm_kernel"},
    "cfg5_per_sample": {"main": "sylvester_kernel"},
    "nsf_k10_h256": {"main": "rq_fused_general_kernel"},
    "nsf_k10_h64": {"main": "rq_fused_linear_kernel4"},
    "cfg1_sample": {"main": "made_inverse_kernel"},
    "maf_rq_sample": {"main": "made_inverse_kernel"},
    "kernels": {"linear_spline_coupling": "fc::LinearSplineOp", "quadratic_spline_coupling": "fc::QuadraticSplineOp",
                "cubic_spline_coupling": "fc::CubicSplineOp", "sum_of_sigmoids_forward": "fc::SoSOp", "sum_of_sigmoids_inverse": "fc::SoSOp",
                "lu_linear_forward": "fc::sylvester_mm_kernel<2", "lu_linear_inverse": "fc::sylvester_mm_kernel<2",
                "householder_shared": "fc::sylvester_mm_kernel<4", "planar": "fc::planar_", "permutation": "fc::permute_rows_kernel",
                "elementwise_tanh": "fc::elementwise_kernel", "standard_normal_log_prob": "std_normal_kernel",
                "rq_spline_backward": "rq_backward_wave_kernel"},
}


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def per_launch(dirname, counter, key, grid=None):
    groups = {}
    names = set()
    for f in glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and key in r["Kernel_Name"]:
                groups.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"][:120])
    if not groups:
        return None
    g = grid if grid in groups else max(groups)
    v = groups[g]
    return {"mean": sum(v) / len(v), "launches": len(v), "grid": g, "symbols": sorted(names)[:3]}


SQ_SETS = ["SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY",
           "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"]


def sq_counters(tag):
    """`--sq`: wave-level SQ counters of every row of `configs.kernels` (what a kernel below the HBM roof is busy with):
    profiles/<tag>_kernels_sq_counters.json."""
    out = os.path.join(ROOT, "gpurun_out", "prof_%s_configs" % tag)
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(ROOT, "flowconductor_amd", "csrc", "libflowcon_hip.so")
    env = dict(os.environ, TMPDIR="/tmp")
    acc = {}
    for i, cset in enumerate(SQ_SETS):
        d = os.path.join(out, "kernels_sq%d" % i)
        cmd = ["rocprofv3", "--pmc"] + cset.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                                          sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "kernels"]
        with open(os.path.join(out, "kernels_sq%d.log" % i), "w") as log:
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT).returncode
        print("[profile_configs] kernels sq set %d rc=%d" % (i, rc), flush=True)
        for row, key in KERNELS["kernels"].items():
            for counter in cset.split():
                r = per_launch(d, counter, key)
                if r:
                    acc.setdefault(row, {"kernel": key})[counter] = r["mean"]
    for row, v in acc.items():
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            v["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            v["wait_memory_frac_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            v["wait_issue_frac_of_wave_cycles"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
    res = {"tag": tag, "library": {"path": "flowconductor_amd/csrc/libflowcon_hip.so", "sha256": sha256(lib)},
           "source": "tools/profile_configs.py --sq: rocprofv3 --pmc passes of tools/bench_configs.py kernels; per-launch means over the "
                     "launches of the largest grid of each kernel symbol (rows that share a symbol and a grid share their numbers); "
                     "SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles", "kernels": acc}
    for dst in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out", "profiles_%s" % tag)):
        os.makedirs(dst, exist_ok=True)
        json.dump(res, open(os.path.join(dst, "%s_kernels_sq_counters.json" % tag), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.endswith("wave_cycles")} for k, v in acc.items()}))


def main():
    tag = sys.argv[1]
    if "--sq" in sys.argv:
        return sq_counters(tag)
    which = sys.argv[2:] or list(KERNELS)
    out = os.path.join(ROOT, "gpurun_out", "prof_%s_configs" % tag)
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(ROOT, "flowconductor_amd", "csrc", "libflowcon_hip.so")
    res = {"tag": tag, "library": {"path": "flowconductor_amd/csrc/libflowcon_hip.so", "sha256": sha256(lib)},
           "source": "tools/profile_configs.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, one process per "
                     "config and counter) of tools/bench_configs.py <config>; per-launch means over the launches of the largest "
                     "grid of each kernel symbol; FETCH_SIZE x2 (gfx950 wide-stream correction), KiB -> bytes",
           "configs": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for cfg in which:
        dirs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, "%s_%s" % (cfg, counter))
            dirs[counter] = d
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), cfg]
            with open(os.path.join(out, "%s_%s.log" % (cfg, counter)), "w") as log:
                rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT).returncode
            print("[profile_configs] %s %s rc=%d" % (cfg, counter, rc), flush=True)
        rows = {}
        for row, key in KERNELS[cfg].items():
            f = per_launch(dirs["FETCH_SIZE"], "FETCH_SIZE", key)
            w = per_launch(dirs["WRITE_SIZE"], "WRITE_SIZE", key, grid=f["grid"] if f else None)
            if not f or not w:
                rows[row] = {"kernel": key, "error": "no launches matched"}
                continue
            rd, wr = 2.0 * f["mean"] * 1024.0, w["mean"] * 1024.0
            rows[row] = {"kernel": key, "symbols": f["symbols"], "grid": f["grid"], "launches_averaged": [f["launches"], w["launches"]],
                         "FETCH_SIZE_KiB_raw": f["mean"], "WRITE_SIZE_KiB_raw": w["mean"], "hbm_read_bytes": rd,
                         "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr}
        res["configs"][cfg] = rows
    dst = os.path.join(ROOT, "profiles", "%s_configs_hbm_traffic.json" % tag)
    if os.path.exists(dst) and sys.argv[2:]:      # a partial re-run: merge into the existing file of the same library
        old = json.load(open(dst))
        if old.get("library", {}).get("sha256") == res["library"]["sha256"]:
            old["configs"].update(res["configs"])
            res = old
    json.dump(res, open(dst, "w"), indent=1)
    carry = os.path.join(ROOT, "gpurun_out", "profiles_%s" % tag)      # gpurun only carries gpurun_out/ back
    os.makedirs(carry, exist_ok=True)
    json.dump(res, open(os.path.join(carry, os.path.basename(dst)), "w"), indent=1)
    print(json.dumps(res["configs"]))


if __name__ == "__main__":
    main()
