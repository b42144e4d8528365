"""How much of its bound every golden comparison uses (tests/test_gpu_golden.py): err, 1e-5 * scale, the reference's
own f32 noise floor, and the multiple of the floor that would be needed.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases  # noqa: E402
from _util import SIZES, build_case, golden, maxdiff  # noqa: E402

dev = torch.device("cuda:0")
pick = sys.argv[1] if len(sys.argv) > 1 else ""
for name in sorted(cases.CASES):
    if pick not in name:
        continue
    g = golden(name)
    t, spec = build_case(name, g)
    t = t.to(dev)
    if spec["tol"][0] == 0 or not spec["inverse"]:
        continue
    worst = {}
    for n in SIZES:
        ctx = torch.from_numpy(g["ctx_%d" % n]).to(dev) if spec["context"] else None
        yin = torch.from_numpy(g["yin_%d" % n]).to(dev)
        with torch.no_grad():
            xi, ladi = t.inverse(yin, ctx)
        for key, got in (("xinv", xi), ("ladinv", ladi)):
            ref = g["%s_%d" % (key, n)]
            k64 = "%s64_%d" % (key, n)
            if k64 not in g.files:
                continue
            scale = max(1.0, float(np.max(np.abs(ref))))
            floor = float(np.max(np.abs(ref.astype(np.float64) - g[k64])))
            err = max(maxdiff(got, ref), maxdiff(got, g[k64]))
            need = max(0.0, (err - 1e-5 * scale)) / floor if floor > 0 else float("inf") if err > 1e-5 * scale else 0.0
            worst[key] = max(worst.get(key, 0.0), need)
            print("%-40s n=%3d %-7s err %.3g  1e-5*scale %.3g  floor %.3g  -> multiple of the floor needed %.2f"
                  % (name, n, key, err, 1e-5 * scale, floor, need))
