"""Generates flowconductor_amd/csrc/fc_rq_fused3_eval.inc: the straight-line RQ-spline evaluation of one
element in the fused final-Linear + spline kernel (K = 8 bins, linear tails), with the element's 23 raw
parameters read from the lane's own MFMA accumulators (macros FC_WH(i): width / height logit i < 16, already
divided by sqrt(hidden_features) and multiplied by log2(e); FC_UD(j): derivative logit j < 7, multiplied by the softplus beta) and 36 MFMA hook points spread evenly
over its instruction stream.  hipcc's sched_group_barrier pipeline clusters about half of the MFMAs,
so the interleave is explicit in the source: FC_HOOK(n) issues MFMA number n of the NEXT block and pins its
position with a sched_barrier.

    python tools/gen_fused_eval.py          # rewrites the .inc (committed; the build does not run this)
    python tools/gen_fused_eval.py --check  # exit status 1 if the committed .inc differs

The arithmetic and its order are those of RQOp::eval_tails_straight (fc_rq_op.h), which restates
flowcon/transforms/splines/rational_quadratic.py:26-38 (tails) and :78-188 (spline).
"""
import os
import sys

# K = 8 (default): the kernel-3 file, two accumulator sets, 36 hooks spread over the whole evaluation.
# --bins K: fc_rq_fused4_eval_k<K>.inc for the K-generic kernel (fc_rq_fused4.hip): ONE accumulator set, so the next
# block's MFMAs (6 per 16-row tile of 4 parameters) may only start once the last raw parameter has left the
# accumulators -- its hooks are spread over the part of the evaluation that follows.
# --box: the form without tails (tails=None, coupling.py:543-547: 3K + 1 parameters, all K + 1 knot derivatives from the
# conditioner, inputs outside the box are an error), fc_rq_fused4_eval_k<K>_box.inc.
K = int(sys.argv[sys.argv.index("--bins") + 1]) if "--bins" in sys.argv else 8
BOX = "--box" in sys.argv
SINGLE_ACC = K != 8 or BOX
HOOKS = 6 * ((3 * K + (1 if BOX else -1) + 3) // 4)
chunks = []  # (code, weight ~ VALU issue slots)


def add(code, w):
    chunks.append((code, w))


add("const bool inside = (x >= q.left) && (x <= q.right);\nconst float xc = inside ? x : q.left;", 3)
if BOX:       # rational_quadratic.py:81-82 (the same interval in both directions); the element passes through unchanged
    add("if (!inside) err |= kErrOutsideDomain;", 1)
add("float mx = -INFINITY, my = -INFINITY;", 1)
# FC_WH(i) hands out the width / height logits in LOG2 units (the kernel folds log2(e) into the constants of the
# fma that undoes the operand scaling), so exp_softmax(d) is a bare v_exp_f32 of the difference.
for i in range(K):
    add("FC_F2 t%d = FC_F2{FC_WH(%d), FC_WH(%d)};\nmx = fmaxf(mx, t%d.x);\nmy = fmaxf(my, t%d.y);" % (i, i, K + i, i, i), 3)
# the derivative logits leave the accumulators early, so the next block's MFMAs can reuse those registers
for j in range(K + 1 if BOX else K - 1):
    add("FC_DER_ST(%d, FC_UD(%d));" % (j if BOX else j + 1, j), 1)
READS_END = len(chunks)   # the accumulators are free from here on
add("const FC_F2 m = {mx, my};", 0)
# Software scheduling (the hooks pin the order, so independent work is laid out by hand): all subtractions first,
# then the 16 exponentials, then the running sum -- a dependent instruction is never the next one issued (packed
# f32 results need a wait state, transcendentals several).
for i in range(K):
    add("t%d = t%d - m;" % (i, i), 1)
for i in range(K):
    add("t%d = FC_F2{__builtin_amdgcn_exp2f(t%d.x), __builtin_amdgcn_exp2f(t%d.y)};" % (i, i, i), 4)
# Partial sums of the exponentials, from the left for the lower knots and from the right for the upper ones:
# l_i = e_0 + .. + e_i (i < K/2), r_i = e_{i+1} + .. + e_{K-1} (i >= K/2), total = l_{K/2-1} + r_{K/2-1}.  The knots
# are affine in them:
#   knot_{i+1} = lo + span sum_{j<=i} (min + c1 e_j / total) = kc_i + l_i (span c1 / total)          (i <  K/2)
#              = hi - span sum_{j>i}  (min + c1 e_j / total) = kc_i - r_i (span c1 / total)          (i >= K/2)
# with kc_i = lo + span min (i + 1) resp. hi - span min (K - 1 - i) and sc1 = span c1 formed once per kernel (in
# double): one packed fma per knot pair instead of normalising, offsetting, accumulating and scaling each bin, and
# every partial sum is at most K/2 - 1 float additions deep and about half of the total in size.  (ATen's cumsum
# accumulates in double on the CPU and in float on the GPU.)
H = K // 2
add("const FC_F2 l0 = t0, r%d = t%d;" % (K - 2, K - 1), 0)
lchain = ["const FC_F2 l%d = l%d + t%d;" % (i, i - 1, i) for i in range(1, H)]
rchain = ["const FC_F2 r%d = r%d + t%d;" % (j, j + 1, j + 1) for j in range(K - 3, H - 2, -1)]
for i in range(max(len(lchain), len(rchain))):     # (odd K: the right chain is one longer)
    if i < len(lchain):
        add(lchain[i], 1)
    if i < len(rchain):
        add(rchain[i], 1)
add("const FC_F2 tot = l%d + r%d;" % (H - 1, H - 1), 1)
add("const float rsx = div_lean(1.f, tot.x);", 5)
add("const float rsy = div_lean(1.f, tot.y);\nconst FC_F2 gk = sc1 * FC_F2{rsx, rsy};\nint idx = 0;", 6)
# Bin search without per-knot selects: the interior knots go to a lane-private LDS table as they are produced
# (slots 0 and K hold the interval ends, written once per kernel), the bin index is a count of compares, and
# the two knots / two derivative logits of the bin come back with four LDS reads.
for i in range(K - 1):
    if i < H:
        add("const FC_F2 next%d = __builtin_elementwise_fma(l%d, gk, kc%d);\nFC_KNOT_ST(%d, next%d);" % (i, i, i, i + 1, i), 2)
    else:
        add("const FC_F2 next%d = __builtin_elementwise_fma(r%d, -gk, kc%d);\nFC_KNOT_ST(%d, next%d);" % (i, i, i, i + 1, i), 2)
    # idx += (xc >= knot) as a VOPC compare into vcc and an add-with-carry of 0: two 4-byte instructions (the compiler's
    # choice, a 64-bit compare into an SGPR pair + v_cndmask + v_addc per pair of knots, costs a third more issue time)
    add("FC_COUNT_GE(idx, xc, kInv ? next%d.y : next%d.x);" % (i, i), 2)
add("const FC_F2 sel_lo = FC_KNOT_LD(idx, 0), sel_hi = FC_KNOT_LD(idx, 1);\n"
    "const float u0 = FC_DER_LD(idx, 0), u1 = FC_DER_LD(idx, 1);", 3)
add("const float xk = sel_lo.x, yk = sel_lo.y;\nconst float wk = sel_hi.x - sel_lo.x, hk = sel_hi.y - sel_lo.y;", 2)
# one v_rcp of the bin width serves both divisions (div_lean: q = a * r, then one residual correction)
add("const float rwk = __builtin_amdgcn_rcpf(wk);\nconst float dq = hk * rwk;\n"
    "const float delta = __builtin_fmaf(__builtin_fmaf(-wk, dq, hk), rwk, dq);", 5)
add("float theta;\nif constexpr (!kInv) {\n  const float tq = (xc - xk) * rwk;\n"
    "  theta = __builtin_fmaf(__builtin_fmaf(-wk, tq, xc - xk), rwk, tq);\n}", 4)
# Two softplus evaluations, written out so hooks can sit inside them.  exp(x) = exp2(x log2e) and log(u) = log2(u) ln2
# without the hi / lo compensation of exp_lean / log_lean: the relative error of the derivative grows by <= 4e-8 |x|
# (<= 2e-7 over the range where the softplus is not yet linear), measured effect on the kernel's logabsdet error
# against float64 in tools/probe/fused_accuracy.py.
for n in (0, 1):
    # (FC_UD hands out the derivative logits already multiplied by the softplus beta -- folded into the unscaling fma)
    add("const float xb%d = u%d;\nconst float xm%d = fminf(xb%d, 20.f);\n"
        "const float ex%d = __builtin_amdgcn_exp2f(xm%d * 1.4426950408889634f);" % (n, n, n, n, n, n), 6)
    # log1p(e) = log(u) + (e - (u - 1)) / u with u = fl(1 + e): the second term restores what the rounding of
    # 1 + e lost (|.| <= 2^-24, so a plain v_rcp is accurate enough for it); no special case for tiny e
    add("const float up%d = 1.f + ex%d;\nconst float rr%d = ex%d - (up%d - 1.f);" % (n, n, n, n, n), 3)
    add("const float l1p%d = __builtin_fmaf(__builtin_amdgcn_logf(up%d), 0.6931471805599453f, rr%d * __builtin_amdgcn_rcpf(up%d));"
        % (n, n, n, n), 10)
    add("const float d%d = q.min_d + (xb%d > 20.f ? xb%d : l1p%d) * inv_beta;" % (n, n, n, n), 3)
add("const float dsum = d0 + d1 - 2.f * delta;", 3)
add("""if constexpr (kInv) {
  const float rr = xc - yk;
  const float qa = rr * dsum + hk * (delta - d0);
  const float qb = hk * d0 - rr * dsum;
  const float qc = -delta * rr;
  const float disc = qb * qb - 4.f * qa * qc;
  if (inside && !(disc >= 0.f)) err |= kErrDiscriminant;
  theta = div_lean(2.f * qc, -qb - sqrt_lean(disc));
}""", 2)
add("const float t1mt = theta * (1.f - theta);\nconst float den = delta + dsum * t1mt;", 4)
add("const float omt = 1.f - theta;\nconst float dn1 = d1 * (theta * theta) + 2.f * delta * t1mt;", 5)
add("const float dnum = (delta * delta) * (dn1 + d0 * (omt * omt));", 4)
# log(dnum) - 2 log(den) = ln2 (log2 dnum - 2 log2 den)
add("const float l2v = __builtin_fmaf(-2.f, __builtin_amdgcn_logf(den), __builtin_amdgcn_logf(dnum));", 3)
add("const float lval = l2v * 0.6931471805599453f;", 1)
add("""float ys;
if constexpr (!kInv) {
  const float num = hk * (delta * (theta * theta) + d0 * t1mt);
  ys = yk + div_lean(num, den);
} else {
  ys = theta * wk + xk;
}""", 8)
add("y = inside ? ys : x;\nlad = inside ? (kInv ? -lval : lval) : 0.f;", 3)

out = ["// GENERATED by tools/gen_fused_eval.py -- do not edit by hand.",
       "// Straight-line RQ-spline evaluation (K = %d, %s) of one element with %d MFMA hook points."
       % (K, "no tails" if BOX else "linear tails", HOOKS),
       "// Expects in scope: FC_WH(i) / FC_UD(j) (logits of the element), FC_KNOT_ST / FC_KNOT_LD / FC_DER_ST / FC_DER_LD",
       "// (lane-private LDS tables of K + 1 knots and K + 1 derivative logits), FC_F2, x, q, inv_beta, err, the knot",
       "// constants sc1, kc0 .. kc%d (FC_F2: x = widths axis, y = heights axis), kInv (constexpr bool), outputs y / lad," % (K - 2),
       "// FC_COUNT_GE(count, a, b): count += (a >= b), and FC_HOOK(n).  FC_WH(i) is expected in log2 units (logit * log2(e))."]
if SINGLE_ACC:
    out.append("// One accumulator set: no hook before the last FC_WH / FC_UD read.")
# hook placement: hook k sits where the accumulated weight passes (k + 1 - SHIFT) / HOOKS of the total
# (FC_GEN_SHIFT: probe knob for tools/probe/search_hooks.sh; the committed file uses 0)
SHIFT = float(os.environ.get("FC_GEN_SHIFT", "0"))
acc = 0.0
hook = 0
if SINGLE_ACC:
    out.extend(code for code, _ in chunks[:READS_END])
    chunks = chunks[READS_END:]
total = sum(w for _, w in chunks)
for code, w in chunks:
    out.append(code)
    acc += w
    while hook < HOOKS and acc >= (hook + 1 - SHIFT) * total / HOOKS:
        out.append("FC_HOOK(%d)" % hook)
        hook += 1
while hook < HOOKS:
    out.append("FC_HOOK(%d)" % hook)
    hook += 1
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flowconductor_amd", "csrc",
                    "fc_rq_fused4_eval_k%d%s.inc" % (K, "_box" if BOX else "") if SINGLE_ACC else "fc_rq_fused3_eval.inc")
path = os.environ.get("FC_GEN_OUT", path)     # probe builds (tools/probe/build_f4_variants.sh)
text = "\n".join(out) + "\n"
if "--check" in sys.argv:      # tests/test_host_logic.py: the committed file is what this script generates
    sys.exit(0 if open(path).read() == text else 1)
open(path, "w").write(text)
print("wrote", path, "chunks", len(chunks), "weight", total)
