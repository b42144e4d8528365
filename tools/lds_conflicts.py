"""LDS bank-conflict calculator for gfx950 access patterns (rules of /opt/skills/guides/MI355X_MICROARCH.md, section LDS):
per-instruction lane groups, bank = (a / 4) mod 64 (ds_read_b64 / b128) or mod 32 (everything else); every extra distinct
address on a busy bank within a group costs one more LDS cycle.

    from lds_conflicts import cycles
    cycles("ds_read_b128", [byte address of lane l for l in range(64)])  ->  (LDS cycles, conflict-free cycles)
"""
GROUPS = {
    "ds_read_b32": [list(range(0, 32)), list(range(32, 64))],
    "ds_read_b64": [list(range(0, 32)), list(range(32, 64))],
    "ds_write_b32": [list(range(0, 32)), list(range(32, 64))],
    "ds_read_b128": [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                     [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                     [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
                     [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]],
    "ds_write_b64": [list(range(16 * i, 16 * i + 16)) for i in range(4)],
    "ds_write_b128": [list(range(8 * i, 8 * i + 8)) for i in range(8)],
    "ds_write_b16": [list(range(0, 32)), list(range(32, 64))],
}
WIDTH = {"ds_read_b32": 4, "ds_read_b64": 8, "ds_write_b32": 4, "ds_read_b128": 16, "ds_write_b64": 8, "ds_write_b128": 16,
         "ds_write_b16": 2}
MOD = {"ds_read_b64": 64, "ds_read_b128": 64}


def cycles(instr, addrs, active=None):
    """(cycles with conflicts, cycles without) for one wave-instruction; ``addrs[l]`` byte address of lane l, ``active`` an
    optional list of booleans."""
    mod = MOD.get(instr, 32)
    total = 0
    for group in GROUPS[instr]:
        banks = {}
        for l in group:
            if active is not None and not active[l]:
                continue
            a = addrs[l]
            for dw in range(a // 4, (a + WIDTH[instr] + 3) // 4):
                banks.setdefault(dw % mod, set()).add(dw)
        total += max([len(v) for v in banks.values()] or [1])
    return total, len(GROUPS[instr])


if __name__ == "__main__":
    # fc_rq_fused3.hip, D = 64, R = 64
    kHB = 72
    for name, stride in (("hfrag, row stride 144 B (kHB = 72)", 144), ("row stride 136 B", 136), ("row stride 160 B", 160),
                         ("row stride 272 B", 272), ("row stride 132", 132)):
        for ks in (0, 1):
            a = [((l & 15) * stride + 64 * ks + 16 * (l >> 4)) for l in range(64)]
            print(name, "ks", ks, cycles("ds_read_b128", a))
    XS = 68
    for w in (0, 3, 7):
        a = [4 * ((l & 15) * XS + 2 * (4 * w + (l >> 4))) for l in range(64)]
        print("x read wave", w, cycles("ds_read_b32", a), "write", cycles("ds_write_b32", a))
