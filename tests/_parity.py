"""Whole-model gradient acceptance, per tensor.

A parameter tensor k passes iff  err_k(build, fp64 oracle) <= max(floor, 3 * err_k(reference fp32, fp64 oracle))
with floor = min(5e-2, 3 * the reference's worst tensor) - App. D's 5e-2 rel-L2, or less where the reference's own
fp32-vs-fp64 noise is smaller than that on EVERY tensor (the 64 x 64 goldens).  No tensor is held to another tensor's
noise, except the short allow-list below.

ALLOW: the parameters in front of `aspp.global_avg_pool.2` (networks/tools/aspp.py:56-61).  That BatchNorm normalises
over N samples only (a 1 x 1 map; N = 2 in every parity case: its output is sign(x1 - x2), 1/std up to 302, App. D), so
the gradients of its gamma / beta and of the 1 x 1 conv feeding it amplify any rounding difference by that 1/std; they
are held to 3 x the reference's WORST tensor instead (the rule every tensor had until round 3)."""
import numpy as np

# (the 2 x 65 x 65 / 4 x 64 x 64 goldens of tests/test_model_gpu.py - 9 x 9 feature maps, BatchNorm over 162 samples - are held
# per tensor through their gradient NORMS; their fixed-cosine PROJECTIONS are checked rank-wise, see check_rankwise)

ALLOW = ("aspp.global_avg_pool.1.weight", "aspp.global_avg_pool.2.weight", "aspp.global_avg_pool.2.bias")
FLOOR = 5e-2


def per_tensor_bounds(ref, names):
    ref = np.asarray(ref, dtype=np.float64)
    noise = float(ref.max())
    floor = min(FLOOR, 3 * noise)
    bound = np.maximum(floor, 3 * ref)
    for i, k in enumerate(names):
        if k in ALLOW:
            bound[i] = max(bound[i], 3 * noise)
    return bound, noise


def check_per_tensor(mine, ref, names, what, capsys=None, dump=None):
    """mine / ref: per-tensor errors against the fp64 run (build, reference fp32).  Asserts the per-tensor bound and
    reports the five tensors closest to it (on success too: the margin belongs in the log)."""
    mine = np.asarray(mine, dtype=np.float64)
    bound, noise = per_tensor_bounds(ref, names)
    ratio = mine / bound
    order = np.argsort(-ratio)
    worst = [(names[i], float(mine[i]), float(ref[i]), float(bound[i]), float(ratio[i])) for i in order[:5]]
    msg = f"[{what}] reference worst tensor {noise:.2e}; closest to their bound (name, mine, ref fp32, bound, mine/bound): " + \
        "; ".join(f"{n} {m:.2e} {r:.2e} {b:.2e} {q:.2f}" for n, m, r, b, q in worst)
    if dump:
        import json
        with open(dump, "w") as f:
            json.dump({"noise": noise, "rows": [(names[i], float(mine[i]), float(ref[i]), float(bound[i])) for i in order]}, f)
    if capsys is not None:
        with capsys.disabled():
            print("\n" + msg)
    else:
        print(msg)
    assert (ratio <= 1.0).all(), msg
    return worst


def check_rankwise(mine, ref, names, what, capsys=None):
    """Distribution-aware bound for heavy-tailed per-tensor errors: the k-th LARGEST error of the build is within
    max(floor, 3 x the k-th largest error of the reference fp32 run), for every k.

    Used for the fixed-cosine gradient projections of the small goldens only.  There a tensor's error is one draw from a
    heavy-tailed distribution - which of a few hundred near-zero pre-activations get the other ReLU mask (App. D) decides
    whether a 256-element bias gradient's projection moves by 1e-3 or by 1.7e-1 - and the build's draw on a tensor is
    independent of the reference's draw on the same tensor (measured: reference 1.2e-2 / build 1.7e-1 on
    backbone.layer3.17.bn1.bias of v3-R101 2x65x65 while the reference's own worst tensor is 1.7e-1).  Tensor by tensor that
    comparison is noise; rank by rank it says the build has no more tensors at any error level than 3x the reference -
    strictly tighter than the round-3 rule (every tensor within 3x the reference's worst)."""
    mine = np.asarray(mine, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    om, orf = np.argsort(-mine), np.argsort(-ref)
    noise = float(ref.max())
    floor = min(FLOOR, 3 * noise)
    bound = np.maximum(floor, 3 * ref[orf])
    ratio = mine[om] / bound
    worst = np.argsort(-ratio)[:5]
    msg = f"[{what}] rank-wise; reference worst tensor {noise:.2e}; closest (rank, name, mine, reference at that rank, mine/bound): " + \
        "; ".join(f"#{k} {names[om[k]]} {mine[om[k]]:.2e} {ref[orf[k]]:.2e} {ratio[k]:.2f}" for k in worst)
    if capsys is not None:
        with capsys.disabled():
            print("\n" + msg)
    else:
        print(msg)
    assert (ratio <= 1.0).all(), msg


def trajectory_band(g, factor=3.0):
    """Per-step band for the loss of a multi-step training run against tests/golden/trajectory_*.npz: the REFERENCE's own
    train.py:255-270 loop in fp64 and in five fp32 variants that differ in summation order only (8 / 4 / 2 / 1 threads,
    oneDNN off).  Gradients decorrelate after a step or two at lr = 0.01 (App. D item 2), the loss CURVE does not: the
    variants stay within 0.2 of the fp64 run over 30 steps while the loss falls by 0.8.  envelope[t] = the largest
    distance any fp32 variant has had from the fp64 run up to step t; band = factor x envelope + 1e-5 |loss| (the
    single-step loss tolerance).  factor 3 is calibrated leave-one-out on the reference itself: each variant against the
    envelope of the OTHER four peaks at 0.9 / 0.9 / 1.3 / 1.8 / 3.6.  Returns (fp64 losses, band)."""
    l64 = g["loss64"]
    spread = np.max([np.abs(g["loss" + str(v)] - l64) for v in g["fp32_variants"]], axis=0)
    return l64, factor * np.maximum.accumulate(spread) + 1e-5 * np.abs(l64)
