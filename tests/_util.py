"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

import cases  # tests/golden/cases.py

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = (7, 64, 257)


class Lib:
    """Namespace the golden cases build from: this repo's drop-in package."""
    from flowconductor_amd import distributions, flows, transforms, utils
    from flowconductor_amd.nn import nets


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def build_case(name, g=None):
    """Construct the case's transform from flowconductor_amd and load the reference's weights."""
    spec = cases.CASES[name]
    torch.manual_seed(1234)
    t = spec["build"](Lib)
    g = golden(name) if g is None else g
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    missing = t.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    t.eval()
    return t, spec


def maxdiff(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)))


def rel_close(actual, expected, rtol, atol):
    """max |a-e| <= atol + rtol*|e| element-wise; returns (ok, worst excess ratio)."""
    a = actual.detach().cpu().double().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual, np.float64)
    e = expected.detach().cpu().double().numpy() if isinstance(expected, torch.Tensor) else np.asarray(expected, np.float64)
    if a.size == 0:
        return True, 0.0
    ratio = np.abs(a - e) / (atol + rtol * np.abs(e))
    return bool(np.all(ratio <= 1.0)), float(np.max(ratio))
