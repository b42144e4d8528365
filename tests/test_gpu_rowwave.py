"""Row-per-wavefront kernels with PER-SAMPLE parameters (the conditional forms of SURVEY.md 8f #1)
and misc. element-wise ops that the golden cases do not reach: HIP vs CPU oracle."""
import copy

import pytest
import torch

from _util import maxdiff
from flowconductor_amd import ops, options
from flowconductor_amd import transforms as T
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k,d", [(5, 3, 7), (300, 4, 64), (64, 32, 128), (33, 2, 200)])
def test_householder_per_sample_q(n, k, d, device):
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, d, generator=gen)
    q = torch.randn(n, k, d, generator=gen)
    ref = O.householder_apply(x, q)
    ref_rev = O.householder_apply(x, q.flip(-2))
    with torch.no_grad():
        got = ops.householder(x.to(device), q.to(device))
        got_rev = ops.householder(x.to(device), q.to(device), reverse=True)
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(got, ref) <= 2e-5 * scale
    assert maxdiff(got_rev, ref_rev) <= 2e-5 * scale
    # orthogonality: norms are preserved; applying forward then reverse is the identity
    assert maxdiff(got.norm(dim=1), x.norm(dim=1)) <= 1e-4 * scale
    with torch.no_grad():
        back = ops.householder(got, q.to(device), reverse=True)
    assert maxdiff(back, x) <= 5e-5 * scale


def test_householder_matrix_is_orthogonal(device):
    h = T.HouseholderSequence(features=24, num_transforms=7)
    with torch.no_grad():
        h.q_vectors.copy_(torch.randn(7, 24))
    h = h.to(device)
    with torch.no_grad():
        q = h.matrix()
        x = torch.randn(50, 24, device=device)
        y, lad = h(x)
    assert maxdiff(q @ q.T, torch.eye(24)) <= 1e-5
    assert maxdiff(y, x @ q.T.T if False else x @ q) <= 2e-5 or maxdiff(y, x @ q.T) <= 2e-5
    assert float(lad.abs().max()) == 0.0
    # det = (-1)^K
    assert abs(float(torch.linalg.det(q.double().cpu())) - (-1.0) ** 7) <= 1e-4


@pytest.mark.parametrize("n,d,m", [(17, 6, 3), (128, 128, 32)])
def test_sylvester_per_sample_parameters(n, d, m, device):
    gen = torch.Generator().manual_seed(d)
    x = torch.randn(n, d, generator=gen)
    q = torch.randn(n, m, d, generator=gen)
    r1 = torch.triu(torch.randn(n, d, d, generator=gen) / d ** 0.5)
    r2 = torch.triu(torch.randn(n, d, d, generator=gen) / d ** 0.5)
    idx = torch.arange(d)
    r1[:, idx, idx] = torch.tanh(r1[:, idx, idx])
    r2[:, idx, idx] = torch.tanh(r2[:, idx, idx])
    bias = torch.randn(n, d, generator=gen) * 0.1
    y_ref, lad_ref = O.sylvester_forward(x, q, r1, r2, bias)
    with torch.no_grad():
        y, lad = ops.sylvester(x.to(device), q.to(device), r1.to(device), r2.to(device), bias.to(device))
    scale = max(1.0, float(y_ref.abs().max()))
    assert maxdiff(y, y_ref) <= 3e-5 * scale
    assert maxdiff(lad, lad_ref) <= 1e-4 * max(1.0, float(lad_ref.abs().max()))


def test_gated_linear_unit(device):
    x = torch.randn(40, 1)
    c = torch.randn(40, 1)
    glu = T.GatedLinearUnit()
    y_ref, lad_ref = O.transform_apply(glu, x, c)
    with torch.no_grad():
        y, lad = glu(x.to(device), c.to(device))
        xb, ladb = glu.inverse(y, c.to(device))
    assert maxdiff(y, y_ref) <= 1e-6 and maxdiff(lad, lad_ref) <= 1e-6
    assert maxdiff(xb, x) <= 1e-5 and maxdiff(ladb, -lad_ref) <= 1e-6


def test_extended_softplus_module(device):
    esp = T.ExtendedSoftplus(features=5).to(device)
    x = torch.randn(30, 5) * 4
    ref_y, ref_lj = O.extended_softplus(x, esp.shift.detach().cpu())
    with torch.no_grad():
        y, lj = esp(x.to(device))
    assert maxdiff(y, ref_y) <= 1e-5 and maxdiff(lj, ref_lj) <= 1e-5


@pytest.mark.parametrize("kind", ["Exp", "Tanh", "Sigmoid", "CauchyCDF"])
def test_inverse_domain_errors(kind, device):
    cls = getattr(T, kind)
    t = cls().to(device)
    bad = torch.full((4, 3), 0.5, device=device)
    bad[2, 1] = {"Exp": -0.1, "Tanh": 1.0, "Sigmoid": 1.2, "CauchyCDF": -0.01}[kind]
    with pytest.raises(T.InputOutsideDomain):
        with torch.no_grad():
            t.inverse(bad)


def test_sum_of_sigmoids_inverse_far_from_origin(device):
    """Reference test adaptive_sigmoid_test.py:64-79: inverse at |x| ~ 200 (outside the initial bracket)."""
    torch.manual_seed(0)
    t = T.SumOfSigmoids(features=3, n_sigmoids=10).to(device)
    x = torch.tensor([[-200.0, 0.3, 190.0], [150.0, -175.0, 0.0]], device=device)
    with torch.no_grad():
        z, lad = t(x)
        xb, ladb = t.inverse(z)
    assert maxdiff(xb, x) <= 2e-3  # float32 resolution at |x| = 200 is 1.5e-5; f' ~ 1
    assert maxdiff(lad + ladb, torch.zeros(2)) <= 1e-4


def test_lu_linear_matches_dense_weight(device):
    """Reference lu_test.py:24-65: outputs == x @ (L U)^T + b, inverse via weight_inverse()."""
    torch.manual_seed(1)
    t = T.LULinear(features=11, identity_init=False).to(device)
    x = torch.randn(20, 11, device=device)
    with torch.no_grad():
        w = t.weight()
        y, lad = t(x)
        xb, ladb = t.inverse(y)
        winv = t.weight_inverse()
    assert maxdiff(y, x @ w.T + t.bias.detach()) <= 1e-5
    assert maxdiff(xb, x) <= 1e-4
    assert maxdiff(winv @ w, torch.eye(11)) <= 1e-4
    assert maxdiff(lad, torch.full((20,), float(torch.linalg.slogdet(w.double().cpu())[1]))) <= 1e-5
    assert maxdiff(lad + ladb, torch.zeros(20)) == 0.0


def test_conditional_planar_and_general_sylvester_vs_oracle(device):
    """Classes the reference cannot run on CPU / for D != 2 (SURVEY.md headline facts): HIP vs the oracle's
    restatement of conditional.py:824-865 and :925-989 (D-general Q)."""
    torch.manual_seed(3)
    ctx = torch.randn(50, 4)
    for t, d in ((T.ConditionalPlanarTransform(features=7, hidden_features=16, context_features=4), 7),
                 (T.ConditionalSylvesterTransform(features=6, hidden_features=16, context_features=4), 6)):
        t.eval()
        x = torch.randn(50, d)
        with torch.no_grad():
            y_ref, lad_ref = O.transform_apply(t, x.clone(), ctx.clone())
            y, lad = t.to(device)(x.to(device), ctx.to(device))
        assert maxdiff(y, y_ref) <= 2e-5 * max(1.0, float(y_ref.abs().max())), type(t).__name__
        assert maxdiff(lad, lad_ref) <= 1e-4, type(t).__name__
        with pytest.raises(TypeError):
            t(x.to(device))
        with pytest.raises(NotImplementedError):
            t.inverse(x.to(device), ctx.to(device))


@pytest.mark.parametrize("d", [3, 64, 100])
def test_per_sample_lu_round_trip(d, device):
    torch.manual_seed(d)
    n = 37
    m = torch.randn(n, d, d, device=device) / d ** 0.5
    x = torch.randn(n, d, device=device)
    with torch.no_grad():
        y, lad = ops.linear_per_sample(x, m, mode=ops.PER_SAMPLE_LU_FORWARD, offdiag_scale=0.3, eps=1e-3,
                                       want_logabsdet=True)
        xb, ladb = ops.linear_per_sample(y, m, mode=ops.PER_SAMPLE_LU_INVERSE, offdiag_scale=0.3, eps=1e-3,
                                         want_logabsdet=True)
        mt = ops.linear_per_sample(x, m, mode=ops.PER_SAMPLE_DENSE)
        mtt = ops.linear_per_sample(x, m, mode=ops.PER_SAMPLE_DENSE_T)
    lower = 0.3 * torch.tril(m, -1) + torch.eye(d, device=device)
    upper = 0.3 * torch.triu(m, 1) + torch.diag_embed(torch.nn.functional.softplus(m.diagonal(0, -1, -2)) + 1e-3)
    ref = (lower @ (upper @ x.unsqueeze(-1))).squeeze(-1)
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(y, ref) <= 2e-5 * scale
    assert maxdiff(lad, upper.diagonal(0, -1, -2).log().sum(-1)) <= 1e-4
    assert maxdiff(xb, x) <= 5e-4 * scale and maxdiff(lad + ladb, torch.zeros(n)) <= 1e-4
    assert maxdiff(mt, (m @ x.unsqueeze(-1)).squeeze(-1)) <= 2e-5 * scale
    assert maxdiff(mtt, (m.transpose(-2, -1) @ x.unsqueeze(-1)).squeeze(-1)) <= 2e-5 * scale


@pytest.mark.parametrize("d,m,n", [(32, 8, 64), (64, 16, 1000), (96, 32, 48), (128, 32, 4096)])
def test_sylvester_matrix_core_path(d, m, n, device, monkeypatch):
    """Shared-weight Sylvester flow as two matrix-core products (fc_sylvester_mm) against the oracle's reflection-
    by-reflection evaluation in float64 and against the row-per-wave kernel."""
    from flowconductor_amd import ops
    from flowconductor_amd.transforms import SylvesterTransform
    from oracle import torch_oracle as O

    torch.manual_seed(d)
    t = SylvesterTransform(features=d, num_householder=m, device=None).eval()
    with torch.no_grad():
        t.Q_orth.q_vectors.copy_(torch.randn(m, d))
        t.bias.copy_(torch.randn(d) * 0.3)
        t.upper_entries1.mul_(2.0)
        t.upper_entries2.mul_(2.0)
    x = torch.randn(n, d)
    with torch.no_grad():
        ref_y, ref_lad = O.transform_apply(copy.deepcopy(t).double(), x.double())
        f32_y, f32_lad = O.transform_apply(t, x)
    t = t.to(device)
    with torch.no_grad():
        with ops.KernelTimer("fc_sylvester_mm") as timer:
            y, lad = t(x.to(device))
        assert len(timer.pairs) == 1, "the matrix-core kernel did not run"
        monkeypatch.setitem(options._values, "sylvester_mm", False)
        y2, lad2 = t(x.to(device))
    sy, sl = max(1.0, float(ref_y.abs().max())), max(1.0, float(ref_lad.abs().max()))
    fy, fl = maxdiff(f32_y.double(), ref_y), maxdiff(f32_lad.double(), ref_lad)
    assert maxdiff(y.cpu().double(), ref_y) <= 1e-5 * sy + 4 * fy
    assert maxdiff(lad.cpu().double(), ref_lad) <= 1e-5 * sl + 4 * fl
    assert maxdiff(y, y2) <= 2e-5 * sy and maxdiff(lad, lad2) <= 2e-5 * sl + 8 * fl


@pytest.mark.parametrize("d,k,n", [(64, 8, 4096), (128, 32, 8192)])
def test_householder_and_lu_dense_matrix_core_path(d, k, n, device):
    """Wide batches with batch-independent parameters: a Householder sequence folded into its orthogonal matrix and
    LULinear's W = L U run as one matrix-core product (fc_dense_mm); against the oracle in float64."""
    torch.manual_seed(k)
    h = T.HouseholderSequence(features=d, num_transforms=k).eval()
    with torch.no_grad():
        h.q_vectors.copy_(torch.randn(k, d))
    lu = T.LULinear(d).eval()
    x = torch.randn(n, d)
    for t, inverse in ((h, False), (h, True), (lu, False)):
        with torch.no_grad():
            ref_y, ref_lad = O.transform_apply(copy.deepcopy(t).double(), x.double(), inverse=inverse)
            f32_y, _ = O.transform_apply(t, x, inverse=inverse)
        td = copy.deepcopy(t).to(device)
        with torch.no_grad(), ops.KernelTimer("fc_dense_mm") as timer:
            y, lad = (td.inverse if inverse else td)(x.to(device))
        assert len(timer.pairs) == 1, "the matrix-core kernel did not run"
        scale = max(1.0, float(ref_y.abs().max()))
        assert maxdiff(y.cpu().double(), ref_y) <= 1e-5 * scale + 4 * maxdiff(f32_y.double(), ref_y)
        assert maxdiff(lad.cpu().double(), ref_lad) <= 1e-5 * max(1.0, float(ref_lad.abs().max()))


@pytest.mark.parametrize("d,m,n", [(12, 5, 300), (128, 32, 512), (70, 3, 65), (200, 5, 40), (300, 17, 24)])
def test_per_sample_sylvester_row_major_upper_triangles(d, m, n, device):
    """The conditional (per-sample) Sylvester form: q [N, M, D], R1 / R2 [N, D, D] row-major as a hyper-network emits
    them, only their upper triangles are read (garbage below the diagonal must not matter); against the formula of
    conditional.py:936-953 in float64."""
    torch.manual_seed(d + m)
    x = torch.randn(n, d)
    q = torch.randn(n, m, d)
    r1 = torch.triu(torch.randn(n, d, d) / d ** 0.5)
    r2 = torch.triu(torch.randn(n, d, d) / d ** 0.5)
    r1.diagonal(dim1=1, dim2=2).tanh_()
    r2.diagonal(dim1=1, dim2=2).tanh_()
    bias = torch.randn(n, d) * 0.1
    xd, qd, b64 = x.double(), q.double(), bias.double()

    def reflect(v, reverse):
        order = range(m - 1, -1, -1) if reverse else range(m)
        for i in order:
            qi = qd[:, i]
            v = v - (v * qi).sum(-1, keepdim=True) * (2.0 / (qi * qi).sum(-1, keepdim=True)) * qi
        return v

    pre = torch.einsum("nij,nj->ni", r1.double(), reflect(xd, True)) + b64
    act = torch.tanh(pre)
    ref_y = xd + reflect(torch.einsum("nij,nj->ni", r2.double(), act), False)
    ref_lad = torch.log(1 + (1 - act ** 2) * (r1.double().diagonal(dim1=1, dim2=2) * r2.double().diagonal(dim1=1, dim2=2))).sum(-1)
    junk = torch.tril(torch.randn(n, d, d) * 100, -1)       # below the diagonal: never read
    with torch.no_grad():
        y, lad = ops.sylvester(x.to(device), q.to(device), (r1 + junk).to(device), (r2 + junk).to(device),
                               bias.to(device))
    assert maxdiff(y, ref_y) <= 2e-5 * max(1.0, float(ref_y.abs().max()))
    assert maxdiff(lad, ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max()) / 10)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 33, 64, 65, 100, 128, 200, 257, 512])
def test_per_sample_linear_modes_over_widths(d, device):
    """fc_linear_per_sample (rows taken eight at a time, LU forward in one pass) over widths that are not multiples of
    the row batch / the wave: M x, M^T x, L (U x) and its inverse against float64 torch."""
    torch.manual_seed(d)
    n = 37
    x = torch.randn(n, d)
    m = torch.randn(n, d, d) / max(1.0, d ** 0.5)
    sp, eps = 0.3, 1e-3
    xd, md = x.double(), m.double()
    lower = sp * torch.tril(md, -1) + torch.eye(d, dtype=torch.float64)
    diag = torch.nn.functional.softplus(md.diagonal(dim1=1, dim2=2)) + eps
    upper = sp * torch.triu(md, 1) + torch.diag_embed(diag)
    refs = {ops.PER_SAMPLE_DENSE: (torch.einsum("nij,nj->ni", md, xd), None),
            ops.PER_SAMPLE_DENSE_T: (torch.einsum("nij,ni->nj", md, xd), None),
            ops.PER_SAMPLE_LU_FORWARD: (torch.einsum("nij,nj->ni", lower, torch.einsum("nij,nj->ni", upper, xd)),
                                        torch.log(diag).sum(-1))}
    with torch.no_grad():
        for mode, (ref, ref_lad) in refs.items():
            out = ops.linear_per_sample(x.to(device), m.to(device), mode=mode, offdiag_scale=sp, eps=eps,
                                        want_logabsdet=ref_lad is not None)
            y, lad = out if isinstance(out, tuple) else (out, None)
            assert maxdiff(y, ref) <= 2e-5 * max(1.0, float(ref.abs().max())), (mode, d)
            if ref_lad is not None:
                assert maxdiff(lad, ref_lad) <= 2e-4 * max(1.0, float(ref_lad.abs().max())), (mode, d)
        y_fwd = refs[ops.PER_SAMPLE_LU_FORWARD][0].float()
        back, lad_inv = ops.linear_per_sample(y_fwd.to(device), m.to(device), mode=ops.PER_SAMPLE_LU_INVERSE,
                                              offdiag_scale=sp, eps=eps, want_logabsdet=True)
    cond = float(torch.linalg.cond(lower @ upper).max())
    assert maxdiff(back, x) <= 1e-5 * max(1.0, cond) * max(1.0, float(x.abs().max())), (d, cond)
    assert maxdiff(lad_inv, -refs[ops.PER_SAMPLE_LU_FORWARD][1]) <= 2e-4 * max(1.0, float(refs[ops.PER_SAMPLE_LU_FORWARD][1].abs().max()))


@pytest.mark.parametrize("d,k,n,per_sample", [(1, 1, 5, True), (2, 3, 1000, True), (5, 7, 333, True), (16, 16, 4099, True),
                                              (16, 5, 64, False), (3, 2, 7, False), (15, 9, 250, True)])
@pytest.mark.parametrize("reverse", [False, True])
def test_narrow_householder_rows(d, k, n, per_sample, reverse, device):
    """Rows of <= 16 features run four samples per wave (one per DPP row); shared and per-sample q, both orders, batch
    sizes that are not multiples of four; against the reflections in float64."""
    torch.manual_seed(10 * d + k)
    x = torch.randn(n, d)
    q = torch.randn(n, k, d) if per_sample else torch.randn(k, d)
    v = x.double()
    qd = q.double()
    order = range(k - 1, -1, -1) if reverse else range(k)
    for i in order:
        qi = qd[:, i] if per_sample else qd[i].expand(n, d)
        v = v - (v * qi).sum(-1, keepdim=True) * (2.0 / (qi * qi).sum(-1, keepdim=True)) * qi
    with torch.no_grad():
        y = ops.householder(x.to(device), q.to(device), reverse=reverse)
    assert y.shape == (n, d)
    assert maxdiff(y, v) <= 2e-5 * max(1.0, float(v.abs().max()))


@pytest.mark.parametrize("scale", [1.0, 40.0])
def test_sum_of_sigmoids_inverse_residual_and_bisection_agreement(scale, device):
    """The inverse is a safeguarded Newton search (5-9 evaluations) instead of the reference's 50 bisection steps
    (no_analytic_inv/base.py:23-103): |f(x^) - z| <= 1e-5 (relative to max(1, |z|)) on random parameters, also far from the
    origin, and the same root as the plain bisection (negative `iterations` = the A/B switch of the C ABI)."""
    torch.manual_seed(4)
    n, d, s = 4096, 5, 30
    params = torch.randn(n, d * (3 * s + 1), device=device) * 1.5
    z = torch.randn(n, d, device=device) * scale
    z[0, 0], z[1, 1] = 200.0, -180.0
    x_newton, lad_n = ops.sum_of_sigmoids(z, params, s, inverse=True)
    x_bisect, lad_b = ops.sum_of_sigmoids(z, params, s, inverse=True, iterations=-50)
    z_back, lad_f = ops.sum_of_sigmoids(x_newton, params, s)
    tol = 1e-5 * z.abs().clamp_min(1.0)
    assert bool(((z_back - z).abs() <= tol).all()), float(((z_back - z).abs() / tol).max())
    assert maxdiff(x_newton, x_bisect) <= 2e-5 * max(1.0, float(x_bisect.abs().max()))
    assert maxdiff(lad_n, lad_b) <= 1e-4 and maxdiff(lad_n + lad_f, torch.zeros(n)) <= 1e-4
