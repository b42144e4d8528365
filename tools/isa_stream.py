"""Prints a one-character-per-instruction picture of a kernel's ISA (M = MFMA, v = VALU, T = transcendental,
r/w = LDS read/write, G/S = global load/store, B = barrier, . = s_waitcnt, s = other scalar) so the
MFMA / VALU interleave of a loop can be judged at a glance.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o k.s file.hip
    python tools/isa_stream.py k.s <substring of the kernel symbol>
"""
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if key in l and not l.startswith(("\t", " ", ".")) and l.split(";")[0].rstrip().endswith(":")][0]
end = [i for i, l in enumerate(lines) if i > start and ("s_endpgm" in l)][-1]
for i in range(start + 1, len(lines)):
    if lines[i].startswith(".Lfunc_end"):
        end = i
        break
out = ""
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if t.startswith(".LBB"):
        out += "\n|" + t.split(":")[0] + "| "
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): out += "M"
    elif op.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")): out += "T"
    elif op.startswith("v_"): out += "v"
    elif op.startswith("ds_read") or op.startswith("ds_load"): out += "r"
    elif op.startswith("ds_write") or op.startswith("ds_store"): out += "w"
    elif op.startswith("ds_"): out += "d"
    elif op.startswith("s_waitcnt"): out += "."
    elif op.startswith("s_barrier"): out += "B"
    elif op.startswith("s_nop"): out += "n"
    elif op.startswith("s_"): out += "s"
    elif op.startswith(("global_load", "buffer_load")): out += "G"
    elif op.startswith(("global_store", "buffer_store")): out += "S"
    else: out += "?"
print(out)
