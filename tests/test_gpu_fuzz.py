"""The six random-shape fuzzers of tools/probe/ as tests: each runs in this process (runpy, no child per case) on
fixed seeds with a case count sized so that the whole file stays inside about a minute on the GPU box.  The fuzzers
compare every kernel family and the fast-path dispatch with the CPU oracle (float64 oracle as the noise floor); in
round 2 they found both real defects of the round while sitting outside the suite.  Plus the frozen case of
fuzz_other_kernels.py seed 4 (cubic autoregressive inverse) against the reference's own results."""
import os
import runpy
import sys

import numpy as np
import pytest
import torch

from _util import GOLDEN_DIR, Lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "probe")

# (script, seeds, cases per seed)
FUZZERS = [("fuzz_kernels.py", (3, 5, 11), 30),
           ("fuzz_tile_kernels.py", (3, 4, 12), 40),
           ("fuzz_other_kernels.py", (4, 6, 13), 40),
           ("fuzz_flows.py", (3, 5, 21), 16),
           ("fuzz_backward.py", (1, 12, 17), 20),
           ("fuzz_ar_inverse.py", (1, 3), 12)]       # round 4: fc_made_inverse against the float64 oracle and the host loops


@pytest.mark.parametrize("script,seed,cases", [(s, seed, c) for s, seeds, c in FUZZERS for seed in seeds])
def test_fuzzer(device, script, seed, cases, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script, str(seed), str(cases)])
    runpy.run_path(os.path.join(PROBE, script), run_name="__main__")
    assert "fuzz ok" in capsys.readouterr().out


def test_frozen_cubic_autoregressive_inverse_case(device):
    """seed 4 / case 22 of fuzz_other_kernels.py.  The reference's float32 round trip on this batch
    is 2.9e-4 (tests/golden/make_fuzz_fixtures.py); the kernel must stay within 4x of it -- with the cancellation-free
    evaluation of Blinn's form it is ~1e-6 -- and within 4x the reference's own float32 error of the reference's
    float64 inverse, element by element where the reference is itself accurate."""
    g = np.load(os.path.join(GOLDEN_DIR, "fuzz_maf_cubic_seed4.npz"))
    t = Lib.transforms.MaskedPiecewiseCubicAutoregressiveTransform(int(g["k"]), int(g["d"]), int(g["hidden"])).eval()
    t.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}, strict=True)
    t = t.to(device)
    x = torch.from_numpy(g["x"]).to(device)
    with torch.no_grad():
        y, lad = t(x)
        back, ladinv = t.inverse(y)
        back_ref_y, _ = t.inverse(torch.from_numpy(g["y"]).to(device))
    ref_rt = float(np.abs(g["xinv"] - g["x"]).max())
    assert 2.5e-4 < ref_rt < 3.5e-4                      # the fixture is the case it claims to be
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= 1e-5
    rt = float((back - x).abs().max())
    assert rt <= 4 * ref_rt, (rt, ref_rt)
    assert rt <= 2e-5, "cancellation-free cubic inverse should round-trip to ~1e-6, got %.3g" % rt
    err64 = np.abs(back_ref_y.cpu().double().numpy() - g["xinv64"])
    floor = np.abs(g["xinv"].astype(np.float64) - g["xinv64"])
    assert np.all(err64 <= 1e-5 + 4 * floor), float((err64 - 4 * floor).max())
    assert float((lad + ladinv).abs().max()) <= 1e-3
