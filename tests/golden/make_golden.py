"""Generate tests/golden/*.npz by importing the REFERENCE (FlowConductor) itself.

Runs only in the build container, where /root/reference exists:
    python tests/golden/make_golden.py
The reference never travels: only the vectors written here are committed.  The reference's
import chain needs the third-party ``UMNN`` package (not installed, not on the hot path); an
empty placeholder module for it is created in a temp dir for the duration of this script
(recipe recorded in SURVEY.md Appendix B).
"""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("FLOWCON_REFERENCE", "/root/reference")


def import_reference():
    stub = tempfile.mkdtemp(prefix="umnn_stub_")
    os.makedirs(os.path.join(stub, "UMNN"))
    with open(os.path.join(stub, "UMNN", "__init__.py"), "w") as f:
        f.write("class NeuralIntegral:\n    apply = None\n\nclass ParallelNeuralIntegral:\n    apply = None\n")
    sys.dont_write_bytecode = True
    sys.path.insert(0, stub)
    sys.path.insert(0, REFERENCE)
    import flowcon  # noqa: F401
    from flowcon import distributions, flows, transforms, utils
    from flowcon.nn import nets

    class L:
        pass

    L.transforms, L.nets, L.utils, L.flows, L.distributions = transforms, nets, utils, flows, distributions
    return L


def make_inputs(spec, n, gen):
    d = spec["features"]
    if spec["in_unit"]:
        x = torch.rand(n, d, generator=gen)
        x[0, 0], x[1, 1] = 0.0, 1.0  # domain edges
    else:
        x = torch.randn(n, d, generator=gen) * spec["x_scale"]
        if spec["clamp"]:
            x = x.clamp(*spec["clamp"])  # rows hitting the clamp sit exactly on the domain edge
    ctx = torch.randn(n, spec["context"], generator=gen) if spec["context"] else None
    return x, ctx


def add_edge_rows(name, t, x):
    """Inputs exactly on / beyond the tail bound for linear-tail splines."""
    tb = getattr(t, "tail_bound", None)
    tails = getattr(t, "tails", None)
    if tails == "linear" and tb is not None and x.shape[0] >= 6:
        x[2, :] = tb
        x[3, :] = -tb
        x[4, :] = tb * 1.5
        x[5, :] = -tb * 2.0
    return x


def main():
    sys.path.insert(0, HERE)
    import cases

    L = import_reference()
    torch.set_num_threads(4)
    for name, spec in cases.CASES.items():
        torch.manual_seed(1234)
        t = spec["build"](L)
        cases.boost_parameters(t, spec["boost"], seed=0)
        if spec["init"] is not None:
            spec["init"](t)
        t.eval()
        gen = torch.Generator().manual_seed(4321)
        t64 = copy.deepcopy(t).double()  # the reference itself in float64 = "truth" for noise floors
        out = {}
        for n in (7, 64, 257):
            x, ctx = make_inputs(spec, n, gen)
            x = add_edge_rows(name, t, x)
            with torch.no_grad():
                y, lad = t(x.clone(), None if ctx is None else ctx.clone())
                out["x_%d" % n] = x.numpy()
                if ctx is not None:
                    out["ctx_%d" % n] = ctx.numpy()
                out["y_%d" % n] = y.numpy()
                out["lad_%d" % n] = lad.numpy()
                if spec["inverse"]:
                    # invert the forward outputs (clamped into the inverse's domain where the
                    # reference's own forward can overshoot it by rounding, e.g. the cubic spline)
                    yin = y.clone()
                    if spec["inv_clamp"]:
                        yin = yin.clamp(*spec["inv_clamp"])
                    out["yin_%d" % n] = yin.numpy()
                    xi, ladi = t.inverse(yin.clone(), None if ctx is None else ctx.clone())
                    out["xinv_%d" % n] = xi.numpy()
                    out["ladinv_%d" % n] = ladi.numpy()
                ctx64 = None if ctx is None else ctx.double()
                try:
                    y64, lad64 = t64(x.double(), ctx64)
                    out["y64_%d" % n] = y64.numpy()
                    out["lad64_%d" % n] = lad64.numpy()
                    if spec["inverse"]:
                        xi64, ladi64 = t64.inverse(yin.double(), ctx64)
                        out["xinv64_%d" % n] = xi64.numpy()
                        out["ladinv64_%d" % n] = ladi64.numpy()
                except Exception as e:  # e.g. a float32 domain edge is outside the float64 box
                    print("  (no float64 truth for %s n=%d: %s)" % (name, n, type(e).__name__))
        for k, v in t.state_dict().items():
            out["sd::" + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, {k: v.shape for k, v in out.items() if not k.startswith("sd::")})


if __name__ == "__main__":
    main()
