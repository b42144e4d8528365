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


def make_knot_fixture(path):
    """Functional-level fixture: inputs EXACTLY on interior knots of the rational-quadratic
    spline, both directions, linear tails and the unit box.  The knots are computed with the reference's own op
    sequence (rational_quadratic.py:91-98, 106-113) so that the chosen inputs are bit-for-bit the knot positions the
    reference's bin search compares against (x == knot k  =>  bin k, theta = 0)."""
    from flowcon.transforms.splines import rational_quadratic as rq
    from torch.nn import functional as F

    def knots(u, lo, hi, min_bin=1e-3):
        k = u.shape[-1]
        w = min_bin + (1 - min_bin * k) * F.softmax(u, dim=-1)
        c = F.pad(torch.cumsum(w, dim=-1), pad=(1, 0), mode="constant", value=0.0)
        c = (hi - lo) * c + lo
        c[..., 0], c[..., -1] = lo, hi
        return c

    gen = torch.Generator().manual_seed(777)
    out = {}
    for tag, k, tails, bound in (("tails_k8", 8, "linear", 3.0), ("box_k10", 10, None, 1.0)):
        n, d = 96, 4
        uw = torch.randn(n, d, k, generator=gen) * 1.5
        uh = torch.randn(n, d, k, generator=gen) * 1.5
        ud = torch.randn(n, d, k - 1 if tails == "linear" else k + 1, generator=gen) * 1.5
        lo, hi = (-bound, bound) if tails == "linear" else (0.0, 1.0)
        which = torch.randint(1, k, (n, d), generator=gen)            # an interior knot per element
        xk = knots(uw, lo, hi).gather(-1, which[..., None])[..., 0]
        yk = knots(uh, lo, hi).gather(-1, which[..., None])[..., 0]
        for direction, x in (("fwd", xk), ("inv", yk)):
            for dt, suffix in ((torch.float32, ""), (torch.float64, "64")):
                args = dict(inputs=x.to(dt).clone(), unnormalized_widths=uw.to(dt).clone(),
                            unnormalized_heights=uh.to(dt).clone(), unnormalized_derivatives=ud.to(dt).clone(),
                            inverse=direction == "inv")
                if tails == "linear":
                    y, lad = rq.unconstrained_rational_quadratic_spline(tails="linear", tail_bound=bound, **args)
                else:
                    y, lad = rq.rational_quadratic_spline(**args)
                out["%s_%s_y%s" % (tag, direction, suffix)] = y.numpy()
                out["%s_%s_lad%s" % (tag, direction, suffix)] = lad.numpy()
            out["%s_%s_x" % (tag, direction)] = x.numpy()
        out["%s_uw" % tag], out["%s_uh" % tag], out["%s_ud" % tag] = uw.numpy(), uh.numpy(), ud.numpy()
        out["%s_knot" % tag] = which.numpy()
    np.savez_compressed(path, **out)
    print("wrote", path, sorted(out))


def make_softplus_4d_fixture(L, path):
    """The reference's ``Softplus`` on a 4-D batch (nonlinearities.py:172-189): it sums the log-Jacobian over the LAST dim
    only (``.sum(-1)``, :182 / :188), so its logabsdet has shape [N, C, H] there, not [N].  The fixture pins what this package
    does with that quirk (tests/test_gpu_round4.py): outputs identical, logabsdet = the reference's summed over the remaining
    non-batch dims."""
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(5, 3, 4, 6, generator=gen) * 2.0
    t = L.transforms.Softplus().eval()
    with torch.no_grad():
        y, lad = t(x.clone())
        xi, ladi = t.inverse(y.clone())
    out = {"x": x.numpy(), "y": y.numpy(), "lad": lad.numpy(), "xinv": xi.numpy(), "ladinv": ladi.numpy()}
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


def main():
    sys.path.insert(0, HERE)
    import cases

    L = import_reference()
    torch.set_num_threads(4)
    make_knot_fixture(os.path.join(HERE, "fn_rq_interior_knots.npz"))
    make_softplus_4d_fixture(L, os.path.join(HERE, "fn_softplus_4d.npz"))
    if "--knots-only" in sys.argv:
        return
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None      # --only name [name ...]
    for name, spec in cases.CASES.items():
        if only is not None and name not in only:
            continue
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
