"""Register / scratch / occupancy table of the kernels of one translation unit (compiled to gfx950 assembly):
    python tools/kernel_stats.py flowconductor_amd/csrc/fc_rq_fused_general_tails.hip [substring of the kernel symbol]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(src, extra=()):
    out = tempfile.mktemp(suffix=".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                    "-S", "--cuda-device-only", "-o", out, os.path.basename(src), *extra], check=True, stderr=subprocess.DEVNULL,
                   cwd=os.path.dirname(os.path.abspath(src)))
    txt = open(out).read()
    os.unlink(out)
    res, name, cur = [], None, {}
    for line in txt.split("\n"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name = m.group(1)
        for key in ("NumVgprs", "ScratchSize", "Occupancy"):
            m = re.match(r"^; %s: (\d+)" % key, line)
            if m:
                cur[key] = int(m.group(1))
        if len(cur) == 3:
            res.append((name, cur["NumVgprs"], cur["ScratchSize"], cur["Occupancy"]))
            cur = {}
    return res


if __name__ == "__main__":
    key = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, v, sc, occ in stats(sys.argv[1]):
        if key in name:
            try:
                name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
            except OSError:
                pass
            print("%-90s vgprs %3d scratch %4d occupancy %d" % (name[:90], v, sc, occ))
