"""Host-side behaviour of the Transform mirror that needs no GPU: constructor checks, buffers,
state_dict keys, error types raised before any kernel launch, mask builders, MADE masks."""
import os

import pytest
import torch

from _util import Lib
from flowconductor_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T, nets, utils = Lib.transforms, Lib.nets, Lib.utils


def _net(i, o):
    return nets.ResidualNet(i, o, hidden_features=8)


def test_coupling_constructor_checks_and_buffers():
    with pytest.raises(ValueError):
        T.AffineCouplingTransform(torch.zeros(2, 2), _net)
    with pytest.raises(ValueError):
        T.AffineCouplingTransform(torch.zeros(0), _net)
    t = T.PiecewiseRationalQuadraticCouplingTransform([1, -1, 1, 0, 1], _net, num_bins=4, tails="linear")
    assert t.identity_features.tolist() == [1, 3] and t.transform_features.tolist() == [0, 2, 4]
    assert t.identity_features.dtype == torch.int64
    assert t.transform_net.final_layer.out_features == 3 * (3 * 4 - 1)
    assert set(k.split(".")[0] for k in t.state_dict()) == {"identity_features", "transform_features", "transform_net"}
    with pytest.raises(ValueError, match="2D or a 4D"):
        t(torch.zeros(3, 5, 2))
    with pytest.raises(ValueError, match="Expected features"):
        t(torch.zeros(3, 4))


def test_transform_dim_multipliers():
    m = utils.create_alternating_binary_mask(6)
    assert T.PiecewiseRationalQuadraticCouplingTransform(m, _net, num_bins=8)._transform_dim_multiplier() == 25
    assert T.PiecewiseRationalQuadraticCouplingTransform(m, _net, num_bins=8, tails="linear")._transform_dim_multiplier() == 23
    assert T.PiecewiseQuadraticCouplingTransform(m, _net, num_bins=8, tails="linear")._transform_dim_multiplier() == 15
    assert T.PiecewiseQuadraticCouplingTransform(m, _net, num_bins=8)._transform_dim_multiplier() == 17
    assert T.PiecewiseCubicCouplingTransform(m, _net, num_bins=8)._transform_dim_multiplier() == 18
    assert T.PiecewiseLinearCouplingTransform(m, _net, num_bins=8)._transform_dim_multiplier() == 8
    assert T.AffineCouplingTransform(m, _net)._transform_dim_multiplier() == 2
    assert T.AdditiveCouplingTransform(m, _net)._transform_dim_multiplier() == 1


def test_masks():
    assert utils.create_alternating_binary_mask(5).tolist() == [1, 0, 1, 0, 1]
    assert utils.create_alternating_binary_mask(5, even=False).tolist() == [0, 1, 0, 1, 0]
    assert utils.create_mid_split_binary_mask(5).tolist() == [1, 1, 1, 0, 0]
    m = utils.create_random_binary_mask(7)
    assert int(m.sum()) == 4 and m.dtype == torch.uint8


def test_made_is_autoregressive():
    """Output block for feature i may depend on inputs < i only (reference autoregressive_test.py:36-41)."""
    torch.manual_seed(0)
    made = T.made.MADE(features=5, hidden_features=16, output_multiplier=3, num_blocks=2)
    x = torch.randn(4, 5, requires_grad=True)
    out = made(x).view(4, 5, 3)
    for i in range(5):
        g = torch.autograd.grad(out[:, i].sum(), x, retain_graph=True)[0]
        assert float(g[:, i:].abs().max()) == 0.0, i
        if i > 0:
            assert float(g[:, :i].abs().max()) > 0.0


def test_permutation_checks():
    with pytest.raises(ValueError):
        T.Permutation(torch.zeros(2, 2, dtype=torch.long))
    with pytest.raises(ValueError):
        T.Permutation(torch.arange(3), dim=0)
    with pytest.raises(ValueError):
        T.RandomPermutation(0)
    p = T.ReversePermutation(4)
    assert p._permutation.tolist() == [3, 2, 1, 0] and list(p.state_dict()) == ["_permutation"]
    with pytest.raises(ValueError, match="No dimension"):
        T.Permutation(torch.arange(3), dim=2)(torch.zeros(2, 3))
    with pytest.raises(ValueError, match="must be of size"):
        p(torch.zeros(2, 5))


def test_misc_constructor_errors():
    with pytest.raises(ValueError):
        T.PointwiseAffineTransform(scale=torch.tensor([1.0, 0.0]))
    with pytest.raises(TypeError):
        T.ActNorm(0)
    with pytest.raises(TypeError):
        T.BatchNorm(1.5)
    with pytest.raises(TypeError):
        T.LULinear(0)
    with pytest.raises(ValueError):
        T.LogTanh(cut_point=0)
    with pytest.raises(ValueError):
        T.LeakyReLU(negative_slope=0)
    with pytest.raises(TypeError):
        T.HouseholderSequence(3, 0)
    with pytest.raises(T.InverseNotAvailable):
        T.Transform().inverse(torch.zeros(1, 1))
    bn = T.BatchNorm(3)
    bn.train()
    with pytest.raises(T.InverseNotAvailable):
        bn.inverse(torch.zeros(2, 3))


def test_state_dict_names_match_reference_layout():
    """Parameter / buffer names are the de-facto checkpoint format."""
    assert set(T.ActNorm(3).state_dict()) == {"initialized", "log_scale", "shift"}
    assert set(T.BatchNorm(3).state_dict()) == {"unconstrained_weight", "bias", "running_mean", "running_var"}
    assert set(T.LULinear(3).state_dict()) == {"bias", "lower_entries", "upper_entries", "unconstrained_upper_diag"}
    assert set(T.HouseholderSequence(4, 2).state_dict()) == {"q_vectors"}
    assert set(T.PlanarTransform(3).state_dict()) == {"w", "b", "u"}
    assert set(T.SylvesterTransform(3, device="cpu").state_dict()) == {
        "upper_entries1", "log_upper_diag1", "upper_entries2", "log_upper_diag2", "bias", "Q_orth.q_vectors"}
    maf = T.MaskedAffineAutoregressiveTransform(features=3, hidden_features=8)
    keys = set(maf.state_dict())
    assert "autoregressive_net.initial_layer.mask" in keys and "autoregressive_net.final_layer.degrees" in keys
    assert "autoregressive_net.blocks.0.linear_layers.1.weight" in keys


def test_householder_default_init_is_identity_pairs():
    q = T.HouseholderSequence(5, 5).q_vectors.detach()
    assert q.tolist() == [[1, 0, 0, 0, 0], [1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0]]


def test_linear_cache_semantics():
    t = T.LULinear(4, using_cache=True)
    assert t.cache.weight is None
    t.eval()
    t._check_forward_cache()
    assert t.cache.weight is not None and t.cache.logabsdet is not None
    t.train()
    assert t.cache.weight is None and t.cache.inverse is None
    with pytest.raises(TypeError):
        t.use_cache("yes")


def test_deferred_error_state_resets():
    from flowconductor_amd import ops

    with pytest.raises(RuntimeError):
        with ops.deferred_errors():
            raise RuntimeError("boom")
    assert ops._state.depth == 0 and not ops._state.dirty


def test_pack_final_layer_layout():
    w = torch.arange(736 * 64, dtype=torch.float32).reshape(736, 64)
    b = torch.arange(736, dtype=torch.float32)
    wpad, bpad = ops.pack_final_layer(w, b)
    assert wpad.shape == (768, 64) and bpad.shape == (768,)
    for prow in (0, 22, 23, 24, 500, 767):
        j, i = divmod(prow, 24)
        expect = torch.zeros(64) if i == 23 else w[j * 23 + i]
        assert torch.equal(wpad[prow], expect)
        assert float(bpad[prow]) == (0.0 if i == 23 else float(b[j * 23 + i]))


def test_pack_resnet_hidden_stacks_linear_weights():
    from flowconductor_amd import ops
    from flowconductor_amd.nn import nets

    net = nets.ResidualNet(32, 8, hidden_features=64, num_blocks=2)
    w0, b0, wb, bb, wc, bc = ops.pack_resnet_hidden(net)
    assert wc is None and bc is None
    assert w0.shape == (64, 32) and b0.shape == (64,) and wb.shape == (4, 64, 64) and bb.shape == (4, 64)
    assert torch.equal(wb[1], net.blocks[0].linear_layers[1].weight) and torch.equal(bb[2], net.blocks[1].linear_layers[0].bias)
    w0, b0, wb, bb, wc, bc = ops.pack_resnet_hidden(nets.ResidualNet(6, 8, hidden_features=64, num_blocks=0))
    assert wb is None and bb is None and w0.shape == (64, 6)
    # a narrower net is zero-padded to the kernel's 64 hidden units
    narrow = nets.ResidualNet(6, 8, hidden_features=20, num_blocks=1)
    w0, b0, wb, bb, wc, bc = ops.pack_resnet_hidden(narrow)
    assert w0.shape == (64, 6) and wb.shape == (2, 64, 64) and float(w0[20:].abs().max()) == 0.0
    assert torch.equal(wb[0, :20, :20], narrow.blocks[0].linear_layers[0].weight) and float(wb[:, 20:].abs().max()) == 0.0
    assert float(wb[:, :, 20:].abs().max()) == 0.0 and float(bb[:, 20:].abs().max()) == 0.0
    wp, bp = ops.pack_final_layer(torch.ones(2 * 23, 20), torch.ones(2 * 23))
    assert wp.shape == (4 * 24, 64) and float(wp[:, 20:].abs().max()) == 0.0 and float(wp[:23, :20].min()) == 1.0
    assert narrow.hidden_padded(torch.zeros(3, 6)).shape == (3, 64)
    assert torch.allclose(narrow.final_from_padded(narrow.hidden_padded(torch.ones(3, 6))), narrow(torch.ones(3, 6)), atol=1e-6)
    # with a context: the initial layer is [identity | context] wide, the blocks' gate layers are stacked
    net = nets.ResidualNet(10, 8, hidden_features=64, context_features=5, num_blocks=3)
    w0, b0, wb, bb, wc, bc = ops.pack_resnet_hidden(net)
    assert w0.shape == (64, 15) and wb.shape == (6, 64, 64) and wc.shape == (3, 64, 5) and bc.shape == (3, 64)
    assert torch.equal(wc[2], net.blocks[2].context_layer.weight)
    # the predicate is device-independent up to the tensor checks: CPU context -> PyTorch path
    assert not net.hip_hidden_supported(20, torch.zeros(4, 5))
    assert not nets.ResidualNet(10, 8, hidden_features=64, context_features=5, num_blocks=4).hip_hidden_supported(
        20, torch.zeros(4, 5))


def test_generated_fused_eval_is_current():
    """flowconductor_amd/csrc/fc_rq_fused3_eval.inc is generated; the committed copy must be the generator's output."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rc = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fused_eval.py"), "--check"]).returncode
    assert rc == 0, "run python tools/gen_fused_eval.py and commit the result"
    for k in (4, 5, 6, 7, 9, 10, 11):       # the K-generic kernel's files (fc_rq_fused4_k<K>.hip)
        rc = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fused_eval.py"), "--bins", str(k), "--check"]).returncode
        assert rc == 0, "run python tools/gen_fused_eval.py --bins %d and commit the result" % k
    for k in (4, 5, 6, 7, 8, 9, 10):        # ... and of the form without tails
        rc = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fused_eval.py"), "--bins", str(k), "--box", "--check"]).returncode
        assert rc == 0, "run python tools/gen_fused_eval.py --bins %d --box and commit the result" % k


def test_pack_final_layer_pads_dims_to_groups_of_four():
    from flowconductor_amd import ops

    for d_t in (1, 5, 16, 21, 32):
        w = torch.randn(d_t * 23, 64)
        b = torch.randn(d_t * 23)
        wpad, bpad = ops.pack_final_layer(w, b)
        dp = -(-d_t // 4) * 4
        assert wpad.shape == (dp * 24, 64) and bpad.shape == (dp * 24,)
        wv, bv = wpad.view(dp, 24, 64), bpad.view(dp, 24)
        assert torch.equal(wv[:d_t, :23], w.view(d_t, 23, 64)) and torch.equal(bv[:d_t, :23], b.view(d_t, 23))
        assert float(wv[:, 23].abs().sum()) == 0.0 and float(wv[d_t:].abs().sum()) == 0.0
        assert float(bv[:, 23].abs().sum()) == 0.0 and float(bv[d_t:].abs().sum()) == 0.0


def test_fused_path_shape_predicates():
    from flowconductor_amd import ops

    ok = dict(n=4096, d=64, d_t=32, hidden=64, num_bins=8, tails="linear")
    assert ops.fused_linear_supported(**ok)
    for key, bad in (("hidden", 128), ("num_bins", 10), ("tails", None), ("d", 132), ("d_t", 33), ("n", 16)):
        assert not ops.fused_linear_supported(**{**ok, key: bad}), (key, bad)
    assert ops.fused_linear_supported(**{**ok, "d_t": 7, "d": 12})
    assert ops.sylvester_mm_supported(1024, 128) and ops.sylvester_mm_supported(16, 32)
    assert not ops.sylvester_mm_supported(1024, 100) and not ops.sylvester_mm_supported(8, 64)
    assert not ops.sylvester_mm_supported(1024, 160)


def test_householder_matrix_matches_sequential_reflections():
    from flowconductor_amd import ops
    from oracle import torch_oracle as O

    torch.manual_seed(2)
    q = torch.randn(5, 12)
    v = torch.randn(7, 12)
    for reverse in (False, True):
        m = ops.householder_matrix(q, reverse=reverse)
        ref = O.householder_apply(v.double(), (q.flip(0) if reverse else q).double())
        assert torch.allclose(v.double() @ m, ref, atol=1e-12)
    w1, w2, rd = ops.pack_sylvester(q, torch.triu(torch.randn(12, 12)), torch.triu(torch.randn(12, 12)))
    assert w1.shape == (12, 12) and w2.shape == (12, 12) and rd.shape == (12,) and w1.dtype == torch.float32


def test_graphed_call_rejects_cpu_tensors():
    from flowconductor_amd.utils.graphs import GraphedCall

    with pytest.raises(ValueError):
        GraphedCall(lambda t: t, torch.zeros(4, 2))


def test_user_defined_transforms_are_left_alone():
    """The boundary is duck-typed: a user's Transform subclass with trainable parameters and plain torch ops
    composes with the package's layers and trains; the missing-backward guard only wraps the package's own classes."""
    class Scale(T.Transform):
        def __init__(self):
            super().__init__()
            self.log_s = torch.nn.Parameter(torch.zeros(3))

        def forward(self, inputs, context=None):
            return inputs * torch.exp(self.log_s), self.log_s.sum().expand(inputs.shape[0])

    t = Scale()
    assert not getattr(Scale.forward, "_guarded", False)
    y, lad = t(torch.ones(4, 3))
    (y.sum() + lad.sum()).backward()
    assert t.log_s.grad is not None
    # a class of the package itself without a backward (none of the shipped ones is left in that state) is wrapped
    def fwd(self, inputs, context=None):
        return inputs, inputs.new_zeros(inputs.shape[0])

    def init(self):
        T.Transform.__init__(self)
        self.p = torch.nn.Parameter(torch.zeros(1))

    internal = type("NoBackward", (T.Transform,), {"__module__": "flowconductor_amd.transforms.fake",
                                                   "__init__": init, "forward": fwd})
    assert getattr(internal.forward, "_guarded", False)
    with pytest.raises(RuntimeError, match="no backward kernel"):
        internal()(torch.ones(4, 3))
    with torch.no_grad():
        internal()(torch.ones(4, 3))


def test_activation_codes_and_made_predicates():
    from torch.nn import functional as F

    assert ops.activation_code(F.relu) == (ops.ACT_RELU, 0.0) and ops.activation_code(torch.nn.ReLU()) == (ops.ACT_RELU, 0.0)
    assert ops.activation_code(torch.tanh)[0] == ops.ACT_TANH and ops.activation_code(F.silu)[0] == ops.ACT_SILU
    assert ops.activation_code(torch.nn.ELU(0.7)) == (ops.ACT_ELU, 0.7) and ops.activation_code(F.elu) == (ops.ACT_ELU, 1.0)
    assert ops.activation_code(torch.nn.LeakyReLU(0.2)) == (ops.ACT_LEAKY_RELU, 0.2)
    assert ops.activation_code(torch.nn.Softsign()) is None and ops.activation_code(lambda v: v) is None
    made = T.made.MADE(features=5, hidden_features=20, num_blocks=2, output_multiplier=2)
    assert made.hip_hidden_supported() and not made.hip_hidden_supported(torch.zeros(4, 3))
    cmade = T.made.MADE(features=5, hidden_features=20, context_features=3, num_blocks=2, output_multiplier=2)
    assert not cmade.hip_hidden_supported()                    # needs its context
    assert not cmade.hip_hidden_supported(torch.zeros(4, 3))   # ... on the device
    assert not T.made.MADE(features=5, hidden_features=80, output_multiplier=2).hip_hidden_supported()
    w, b = made.masked_final(64)
    assert w.shape == (10, 64) and float(w[:, 20:].abs().max()) == 0.0
    assert torch.equal(w[:, :20], made.final_layer.weight.detach() * made.final_layer.mask)


def test_rank_plan_maps_ranks_to_devices_seeds_and_shards():
    """one process per GPU (device = LOCAL_RANK), inputs seeded 1234 + rank, contiguous shards; weak
    scaling keeps 2^20 rows per GPU (world 8 = BASELINE configs[3]), strong scaling splits configs[3]'s 2^23 rows."""
    from flowconductor_amd import parallel

    world = 8
    weak = [parallel.rank_plan(r, world, r, "weak", rows_per_gpu=1 << 20) for r in range(world)]
    assert [p["device_index"] for p in weak] == list(range(8))
    assert [p["seed"] for p in weak] == [1234 + r for r in range(8)]
    assert all(p["n_local"] == 1 << 20 for p in weak)
    assert weak[0]["row_lo"] == 0 and weak[-1]["row_hi"] == 1 << 23
    assert all(a["row_hi"] == b["row_lo"] for a, b in zip(weak, weak[1:]))
    for w in (1, 2, 4, 8, 3):
        strong = [parallel.rank_plan(r, w, r, "strong", total_rows=1 << 23) for r in range(w)]
        assert strong[0]["row_lo"] == 0 and strong[-1]["row_hi"] == 1 << 23
        assert sum(p["n_local"] for p in strong) == 1 << 23
        assert all(a["row_hi"] == b["row_lo"] for a, b in zip(strong, strong[1:]))
        assert max(p["n_local"] for p in strong) - min(p["n_local"] for p in strong) <= 1
    with pytest.raises(ValueError):
        parallel.rank_plan(8, 8, 0)
    with pytest.raises(ValueError):
        parallel.rank_plan(0, 2, 0, "strong")


def test_options_override_nests_and_restores():
    from flowconductor_amd import options

    assert options.get("fused_final_layer") is True
    with options.override(fused_final_layer=False):
        assert options.get("fused_final_layer") is False
        with options.override(fused_hidden=False, fused_final_layer=True):
            assert options.get("fused_final_layer") is True and options.get("fused_hidden") is False
        assert options.get("fused_final_layer") is False and options.get("fused_hidden") is True
    assert options.get("fused_final_layer") is True
    with pytest.raises(KeyError):
        with options.override(no_such_switch=1):
            pass
    try:
        with options.override(sylvester_mm=False):
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert options.get("sylvester_mm") is True


def test_product_path_reads_no_environment_switches():
    """A stray environment variable must not change what is timed: no os.environ / getenv under flowconductor_amd/."""
    import re

    pkg = os.path.join(ROOT, "flowconductor_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(base, f)).read()
                if re.search(r"os\.environ|getenv\s*\(", text):
                    offenders.append(os.path.relpath(os.path.join(base, f), ROOT))
    assert not offenders, offenders


def test_device_pack_merge_and_refresh_launch_once_per_weight_version(monkeypatch):
    """ops.DevicePack: merged packs refresh together, once per parameter version / cache epoch; staging sources that a
    prepare callable rewrites do not re-trigger the launch (host logic only: the launch itself is stubbed)."""
    from flowconductor_amd import ops

    launches = []
    monkeypatch.setattr(ops.DevicePack, "run", lambda self: launches.append(len(self.root().all_jobs())))
    w1, b1 = torch.nn.Parameter(torch.randn(8, 4)), torch.nn.Parameter(torch.randn(8))
    w2 = torch.nn.Parameter(torch.randn(4, 4))
    out = torch.empty(64, dtype=torch.float16)
    un = torch.empty(2)
    a, b = ops.DevicePack(torch.device("cpu")), ops.DevicePack(torch.device("cpu"))
    a.add(ops.PACK_HIDDEN, w1, b1, out, un[:1], nks=1, nt=1)
    b.add(ops.PACK_HIDDEN, w2, None, out, un[1:], nks=1, nt=1)
    stage = torch.zeros(4, 4)
    b.add(ops.PACK_HIDDEN, stage, None, out, un[1:], nks=1, nt=1, track=False)
    b.prepare.append(lambda: stage.copy_(w2.detach()))
    a.merge(b)
    assert b.root() is a and len(a.all_jobs()) == 3 and len(b.jobs) == 2 and len(b.prepare) == 1
    a.refresh()
    b.refresh()                      # same versions: the merged pack has run already
    assert launches == [3]
    stage.add_(1.0)                  # a staging buffer is not a source
    a.refresh()
    assert launches == [3]
    with torch.no_grad():
        w2.mul_(2.0)                 # an in-place optimizer-style update bumps the version
    b.refresh()
    assert launches == [3, 3]
    ops.invalidate_hip_caches()      # .data surgery: the caller bumps the epoch
    a.refresh()
    assert launches == [3, 3, 3]
    # a REBUILT parent (the final layer's storage moved) adopts the live child; the old parent lets go of it and keeps
    # only its own -- now stale -- jobs to itself: nothing that is launched again points into freed storage
    a2 = ops.DevicePack(torch.device("cpu"))
    a2.add(ops.PACK_HIDDEN, torch.nn.Parameter(torch.randn(8, 4)), None, out, un[:1], nks=1, nt=1)
    a2.merge(b)
    assert b.root() is a2 and len(a2.all_jobs()) == 3 and len(a.all_jobs()) == 1 and a.children == []
    a2.merge(b)                      # idempotent
    assert len(a2.all_jobs()) == 3
    b.refresh()
    assert launches == [3, 3, 3, 3]


def test_has_hooks_sees_hooks_added_after_the_module_list_was_memoised():
    from flowconductor_amd import ops
    from flowconductor_amd.nn import nets

    net = nets.ResidualNet(4, 6, hidden_features=8, num_blocks=1)
    assert not ops.has_hooks(net)
    handle = net.blocks[0].linear_layers[0].register_forward_pre_hook(lambda m, i: None)
    assert ops.has_hooks(net)
    handle.remove()
    assert not ops.has_hooks(net)
    net.add_module("extra", torch.nn.Linear(2, 2))
    net.extra.register_forward_hook(lambda m, i, o: None)
    assert ops.has_hooks(net)


def test_affine_tail_lds_budget():
    from flowconductor_amd import ops

    assert ops.affine_tail_fits(16, 2, 32) and ops.affine_tail_fits(48, 2, 80) and ops.affine_tail_fits(32, 3, 64)
    assert not ops.affine_tail_fits(16, 4, 32)        # four blocks + the final layer: the image alone is too large
    assert not ops.affine_tail_fits(16, 2, 200)       # row tiles of D > 128
    assert not ops.affine_tail_fits(64, 3, 128)


def test_runtime_caches_do_not_travel_with_deepcopy_or_pickle():
    """Packed-weight plans (fc_pack_job structs with raw device pointers, closures) live on the modules; a deep copy /
    pickle / torch.save of a module that has run must succeed and start cold (EMA snapshots, whole-module
    checkpoints -- the reference supports both)."""
    import copy
    import io
    import pickle

    from flowconductor_amd import _hip

    t = T.PiecewiseRationalQuadraticCouplingTransform(utils.create_alternating_binary_mask(8), _net, num_bins=4)
    pack = ops.DevicePack(torch.device("cpu"))
    pack.prepare.append(lambda: None)                  # a local closure, as the real plans hold
    t._train_pack = [t.transform_net.final_layer.weight, 0, pack, []]
    t._tail_image = [0, pack, (torch.zeros(3),)]
    t.transform_net._hip_image = [0, _hip.PackJob(), (torch.zeros(3),)]
    t.transform_net._hip_packed = ((1, 2), torch.zeros(5))
    for name in ("_train_pack", "_tail_image"):
        assert name in ops.RUNTIME_CACHE_ATTRS
    with pytest.raises(Exception):
        pickle.dumps(t._train_pack)                    # what used to break deepcopy of the whole module
    t2 = copy.deepcopy(t)
    assert t2._train_pack is None and t2._tail_image is None
    assert t2.transform_net._hip_image is None and t2.transform_net._hip_packed is None
    assert t._train_pack is not None and t.transform_net._hip_image is not None    # the original keeps its plans
    assert t2.transform_net.final_layer.weight.data_ptr() != t.transform_net.final_layer.weight.data_ptr()
    buf = io.BytesIO()
    torch.save(t, buf)
    buf.seek(0)
    t3 = torch.load(buf, weights_only=False)
    assert t3._train_pack is None and t3.transform_net._hip_image is None
    assert all(torch.equal(a, b) for a, b in zip(t3.state_dict().values(), t.state_dict().values()))
    made = T.made.MADE(features=4, hidden_features=8)
    made._hip_packed = (0, _hip.PackJob())
    assert copy.deepcopy(made)._hip_packed is None


def test_param_list_memo_sees_replaced_parameter_objects():
    """ResidualNet._param_list() (the source list of every packed-weight key) follows Parameter objects that are
    REPLACED rather than written in place: direct assignment and load_state_dict(assign=True)."""
    net = nets.ResidualNet(4, 6, hidden_features=8)
    before = net._param_list()
    assert before is net._param_list()                                   # memoised
    old = net.final_layer.weight
    net.final_layer.weight = torch.nn.Parameter(torch.zeros_like(old))
    after = net._param_list()
    assert any(p is net.final_layer.weight for p in after) and not any(p is old for p in after)
    key = net._storage_key()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd, assign=True)
    assert net._storage_key() != key
    assert all(a is b for a, b in zip(net._param_list(), net.parameters()))


@pytest.mark.parametrize("features,hidden,blocks,per_dim", [(8, 64, 2, 23), (64, 64, 2, 2), (5, 24, 1, 16), (33, 50, 3, 2)])
def test_made_pass_prefix_follows_the_masks(features, hidden, blocks, per_dim):
    """fc_made_inverse's units_needed: the renumbering is a permutation, the counts grow with the pass, they are the units
    of degree <= d for the reference's degrees (made.py:13-24), and no unit inside a pass's prefix reads one outside it."""
    from flowconductor_amd.transforms.made import MADE

    made = MADE(features, hidden, num_blocks=blocks, output_multiplier=per_dim)
    order, need = ops._made_pass_prefix(made, features, per_dim, 64)
    assert sorted(order.tolist()) == list(range(64))
    need = need.tolist()
    assert need == sorted(need) and need[0] == 0 and need[-1] <= hidden
    degrees = torch.arange(hidden) % max(1, features - 1) + min(1, features - 1)
    assert need == [int((degrees <= d).sum()) for d in range(features)]
    rank = torch.empty(64, dtype=torch.long)
    rank[order] = torch.arange(64)
    for lin in [l for b in made.blocks for l in b.linear_layers]:
        reads = (lin.mask != 0).nonzero()
        for d in range(features):
            inside = rank[reads[:, 0]] < need[d]
            assert bool((rank[reads[inside, 1]] < need[d]).all())
    final = (made.final_layer.mask != 0).reshape(features, per_dim, -1).any(dim=1)
    for d in range(features):
        assert bool((rank[final[d].nonzero().reshape(-1)] < need[d]).all())
