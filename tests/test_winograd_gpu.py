"""Winograd F(2x2, 3x3) path (dcfp_amd/csrc/conv_winograd.hip) for the wide 3x3 stride-1 convs
(networks/backbone/resnet.py:27-28, networks/tools/aspp.py:37-39, networks/deeplabv3.py:25-41): forward and dgrad
against fp64 on the CPU, next to the direct LDS-DMA kernels on the same inputs (the library reads
DCFP_CONV_WINOGRAD once per process: 0 = direct kernels, 1 = cost model (default), 2 = wherever eligible; hence
child processes), and the direct-kernel test files once more with the switch off, so that both algorithms stay
covered whichever one the dispatcher prefers for a shape."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (N, Cin, H, W, Cout, dilation): even / odd sizes against the 2d x 2d super-blocks, every dilation of the model (1, 2, 4, 8, 16, 12, 24, 36),
# column counts whose tile rows need padding to 16-byte quads, one shape per GEMM regime (K = 256 / >= 512)
SHAPES = [(2, 256, 64, 128, 256, 2), (4, 256, 50, 68, 512, 1), (5, 288, 33, 60, 256, 4), (2, 512, 64, 128, 256, 12),
          (1, 256, 96, 192, 512, 24), (2, 256, 47, 129, 256, 2),
          # layer4's multi-grid dilations (resnet.py:124-141: 8 and 16; 2 * 16 divides the 128-row map exactly, no tile
          # padding) and the ASPP's 36 (aspp.py:40-47; 2 * 36 does not divide 128 / 256: 27 % of the tiles are padding)
          (2, 512, 64, 128, 512, 8), (1, 512, 128, 256, 512, 16), (1, 256, 128, 256, 256, 36)]


def _child():
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from dcfp_amd import _lib, ops
    dev = torch.device("cuda:0")
    out = {}
    for (N, Cin, H, W, Cout, d) in SHAPES:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(N, Cin, H, W, generator=g)
        x = torch.relu(x) + 0.05 * x                                     # post-ReLU-like statistics
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        dy = torch.randn(N, Cout, H, W, generator=g)
        seed = torch.randn(N, Cin, H, W, generator=g)
        xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
        desc = ops._desc(x.shape, w.shape, 1, d, d)
        names = [ops.conv_kernel_name(desc, k) for k in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)]
        y = ops.conv2d_fwd(xd, wd, None, 1, d, d)
        # forward into a channel slice of a wider tensor (the ASPP concat, aspp.py:77)
        wide = torch.full((N, Cout + 64, H, W), 7.0, device=dev)
        ops.conv2d_fwd(xd, wd, None, 1, d, d, out=wide[:, 32:32 + Cout])
        dx = ops.conv2d_dgrad(dyd, wd, tuple(x.shape), 1, d, d)
        acc = seed.to(dev).clone()
        ops.conv2d_dgrad(dyd, wd, tuple(x.shape), 1, d, d, out=acc, accumulate=True)
        # dgrad of a channel slice of a wider gradient (the ASPP backward reads its branch's slice in place)
        dwide = torch.randn(N, Cout + 64, H, W, generator=g).to(dev)
        dwide[:, 32:32 + Cout] = dyd
        dx_slice = ops.conv2d_dgrad(dwide[:, 32:32 + Cout], wd, tuple(x.shape), 1, d, d)
        dw, _ = ops.conv2d_wgrad(dyd, xd, tuple(w.shape), 1, d, d)
        dw_slice, _ = ops.conv2d_wgrad(dwide[:, 32:32 + Cout], xd, tuple(w.shape), 1, d, d)
        # the forward call can leave its transformed input behind for the weight gradient: same y, same dw
        kp = {}
        y_keep = ops.conv2d_fwd(xd, wd, None, 1, d, d, keep=kp)
        dw_keep, _ = ops.conv2d_wgrad(dyd, xd, tuple(w.shape), 1, d, d, xform=kp.get("xform"))
        rec = {"kernels": names, "wgrad_slice_equal": bool(torch.equal(dw, dw_slice)),
               "kept": "xform" in kp, "keep_equal": bool(torch.equal(y_keep, y) and torch.equal(dw_keep, dw)), "slice_equal": bool(torch.equal(wide[:, 32:32 + Cout], y)),
               "slice_untouched": bool((wide[:, :32] == 7.0).all() and (wide[:, 32 + Cout:] == 7.0).all()),
               "dgrad_slice_equal": bool(torch.equal(dx_slice, dx)),
               "frac": [ops.conv_executed_fraction(desc, k) for k in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)],
               "scratch": [int(_lib.lib().dcfp_conv2d_workspace_is_scratch(ops.C.byref(desc), k))
                           for k in (_lib.CONV_FWD, _lib.CONV_DGRAD)]}
        # row-pitched operands (zero tail behind each row) where the conv takes them
        pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), 1, d, d)
        if pitch:
            xp = ops.pitched_buffer(tuple(x.shape), pitch, "t_x", dev); xp.copy_(xd)
            dyp = ops.pitched_buffer(tuple(dy.shape), pitch, "t_dy", dev); dyp.copy_(dyd)
            rec["pitched_equal"] = bool(torch.equal(ops.conv2d_fwd(xp, wd, None, 1, d, d), y) and
                                        torch.equal(ops.conv2d_dgrad(dyp, wd, tuple(x.shape), 1, d, d), dx))
            # the weight gradient of pitched operands may be another kernel (the fused Winograd weight gradient needs the
            # zero tails for dilation 1 / 2: conv_winograd3.hip): same bits where it is the same kernel, else against fp64
            pdesc = ops._desc(x.shape, w.shape, 1, d, d, pitch, pitch)
            rec["pitched_wgrad_kernel"] = ops.conv_kernel_name(pdesc, _lib.CONV_WGRAD)
            dw_p = ops.conv2d_wgrad(dyp, xp, tuple(w.shape), 1, d, d)[0]
            rec["pitched_wgrad_equal"] = bool(torch.equal(dw_p, dw))
        # fused statistics (where every tile is interior) against the separate statistics kernel on the same y
        y_s, st = ops.conv2d_fwd(xd, wd, None, 1, d, d, want_stats=True)
        rec["stats_equal_y"] = bool(torch.equal(y_s, y))
        if st is not None:
            m_ref, v_ref = ops.bn_stats(y)
            rec["stats_err"] = [float((st[0] - m_ref).abs().max() / m_ref.abs().max()),
                                float(((st[1] - v_ref).abs() / v_ref).max())]
        # inference: eval-mode BatchNorm (+residual) (+ReLU) folded into the op's epilogue
        sc = (torch.rand(Cout, generator=g) + 0.5).to(dev); sh = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        rs = torch.randn(N, Cout, H, W, generator=g).to(dev)
        yf = ops.conv2d_fused_infer(xd, wd, sc, sh, 1, d, d, rs, True)
        torch.cuda.synchronize()
        ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
        reff = torch.relu(ref * sc.cpu().double()[None, :, None, None] + sh.cpu().double()[None, :, None, None] + rs.cpu().double())
        rec["fused_infer_max"] = float((yf.cpu().double() - reff).abs().max() / reff.abs().max())
        refdx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
        refdw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, d, d)

        def emax(a, b):
            return float((a.cpu().double() - b).abs().max() / b.abs().max())

        def erel(a, b):
            return float((a.cpu().double() - b).norm() / b.norm())
        rec.update(fwd_max=emax(y, ref), fwd_rel=erel(y, ref), dgrad_max=emax(dx, refdx), dgrad_rel=erel(dx, refdx),
                   acc_max=emax(acc, refdx + seed.double()), wgrad_max=emax(dw, refdw), wgrad_rel=erel(dw, refdw))
        if pitch:
            rec.update(pitched_wgrad_max=emax(dw_p, refdw), pitched_wgrad_rel=erel(dw_p, refdw))
        out[f"{N}x{Cin}x{H}x{W}->{Cout} d{d}"] = rec
    print("WINO_RESULT " + json.dumps(out))


def _run(mode):
    env = dict(os.environ, DCFP_CONV_WINOGRAD=mode, DCFP_CONV_MATH="f32")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("WINO_RESULT ")][-1]
    return json.loads(line[len("WINO_RESULT "):])


def test_winograd_forward_dgrad_vs_fp64_and_vs_direct(cuda):
    wino, direct = _run("2"), _run("0")
    assert wino.keys() == direct.keys() and len(wino) == len(SHAPES)
    assert any("stats_err" in r for r in wino.values())          # the statistics epilogue of the output transform ran
    wg = [r["kernels"][2] for r in wino.values()]
    assert any("fused" in n for n in wg) and any("fused" not in n for n in wg), wg      # both Winograd weight gradients ran
    for k, rec in wino.items():
        ref = direct[k]
        # (every pass, also the 288-channel dgrad off the 256 grid: the batched GEMM pads M on its edge tiles)
        want = [True, True, True]
        assert [n.startswith("winograd_f2x2_3x3") for n in rec["kernels"]] == want, (k, rec["kernels"])
        assert not any(n.startswith("winograd") for n in ref["kernels"]), (k, ref["kernels"])
        # a Winograd pass' workspace is pure scratch (three passes) or nothing but its kept transformed filters (the fused
        # kernel, round 4: dcfp_conv2d_workspace_is_scratch == 0, wp_valid honoured); the direct kernels keep permuted copies
        assert all(v in (0, 1) for v in rec["scratch"]) and ref["scratch"] == [0, 0]
        if os.environ.get("DCFP_WINO_KEEP_U", "1") == "0":
            assert rec["scratch"] == [1, 1]
        assert all(0.44 <= f <= 0.60 for f, v in zip(rec["frac"], want) if v), (k, rec["frac"])   # 16/36 x tile padding
        assert rec["slice_equal"] and rec["slice_untouched"] and rec["dgrad_slice_equal"] and rec["wgrad_slice_equal"], (k, rec)
        assert rec.get("pitched_equal", True), k
        # the forward leaves its transformed input behind exactly where the weight gradient is the BATCHED Winograd path
        # (dense dilation-1 / 2 operands here); the fused weight gradient (dilation >= 4, pitched operands) transforms x itself
        fused_wg = "fused" in rec["kernels"][2]
        assert rec["kept"] == (not fused_wg) and rec["keep_equal"] and not ref["kept"] and ref["keep_equal"], \
            (k, rec["kernels"][2], rec["kept"], rec["keep_equal"])
        if "pitched_wgrad_kernel" in rec:
            assert rec["pitched_wgrad_kernel"].startswith("winograd_f2x2_3x3 wgrad fused"), (k, rec["pitched_wgrad_kernel"])
            if rec["pitched_wgrad_kernel"] == rec["kernels"][2]:
                assert rec["pitched_wgrad_equal"], k
            assert rec["pitched_wgrad_max"] < 2e-6 and rec["pitched_wgrad_rel"] <= 2.5 * ref["wgrad_rel"] + 1e-8, \
                (k, rec["pitched_wgrad_max"], rec["pitched_wgrad_rel"], ref["wgrad_rel"])
        if "pitched_wgrad_kernel" in ref:
            assert ref["pitched_wgrad_equal"], k          # direct kernels: pitched == dense, bit for bit
        assert rec["stats_equal_y"] and rec["fused_infer_max"] < 3e-6, (k, rec["fused_infer_max"])
        if "stats_err" in rec:
            assert rec["stats_err"][0] < 2e-5 and rec["stats_err"][1] < 2e-5, (k, rec["stats_err"])
        # the stated fp32 tolerance of the conv tests (tests/test_ops_gpu.py: 3e-6 * max(1, sqrt(K) / 8), K = 9 Cin) ...
        K = 9 * int(k.split("x")[1])
        tol = 3e-6 * max(1.0, math.sqrt(K) / 8)
        for f in ("fwd_max", "dgrad_max", "acc_max"):
            assert rec[f] < tol, (k, f, rec[f], tol)
        # ... and no less accurate than the direct kernels (one K = 9 Cin accumulation chain there, 16 chains of
        # length Cin plus 24 additions per output here): relative error within 1.25x of theirs
        assert rec["fwd_rel"] <= 1.25 * ref["fwd_rel"] + 1e-8, (k, rec["fwd_rel"], ref["fwd_rel"])
        assert rec["dgrad_rel"] <= 1.25 * ref["dgrad_rel"] + 1e-8, (k, rec["dgrad_rel"], ref["dgrad_rel"])
        # weight gradient: dw = G^T dU G takes differences of the 16 transformed sums (each a few times larger than
        # the result): measured 1.1...1.9x the direct kernels' error, an order of magnitude inside the stated
        # tolerance of the conv tests (2e-5 on weight gradients)
        assert rec["wgrad_max"] < 2e-6 and rec["wgrad_rel"] <= 2.5 * ref["wgrad_rel"] + 1e-8, (k, rec["wgrad_max"],
                                                                                             rec["wgrad_rel"], ref["wgrad_rel"])


# narrow 3x3 convs (stem 64 -> 64 -> 128, layer1 / layer2 conv2: resnet.py:88-96, 107-122) and ragged pruned widths: below
# the three-pass path's 129 / 128-channel floor, Winograd through the fused kernels only - forward, dgrad and (round 4)
# weight gradient on row-pitched operands
# (sizes from which the cost model prefers Winograd: a launch of a few GFLOP)
NARROW = [(2, 64, 128, 256, 64, 1), (2, 64, 96, 160, 128, 1), (3, 128, 64, 128, 128, 1), (4, 100, 64, 128, 120, 2), (2, 96, 96, 160, 200, 4)]


@pytest.mark.parametrize("shape", NARROW)
def test_narrow_layers_on_the_fused_winograd_kernel(cuda, shape):
    import torch
    import torch.nn.functional as F
    from dcfp_amd import _lib, ops
    if os.environ.get("DCFP_CONV_WINOGRAD", "1") == "0" or os.environ.get("DCFP_WINO_FUSED", "1") == "0":
        pytest.skip("Winograd / fused kernel switched off")
    N, Cin, H, W, Cout, d = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = torch.relu(x) + 0.05 * x
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(N, Cout, H, W, generator=g)
    pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), 1, d, d)
    assert (pitch > 0) == (d <= 2), (shape, pitch)            # dilation 1 / 2 take the pitch, dilation 4 reads dense rows
    xs, dys = x.to(cuda), dy.to(cuda)
    if pitch:
        xs = ops.new_pitched(tuple(x.shape), pitch, cuda); xs.copy_(x.to(cuda))
        dys = ops.new_pitched(tuple(dy.shape), pitch, cuda); dys.copy_(dy.to(cuda))
    desc = ops._desc(x.shape, w.shape, 1, d, d, pitch, pitch)
    names = [ops.conv_kernel_name(desc, k) for k in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)]
    assert names[0].startswith("winograd_f2x2_3x3 fused") and names[1].startswith("winograd_f2x2_3x3 fused"), names
    # weight gradient: the fused Winograd weight-gradient kernel (conv_winograd3.hip, round 4: 64 x 64-channel blocks, so it
    # applies to these widths) where the operands allow it - pitched rows of a multiple of 4 pixels for dilation 1 / 2, any
    # even width for dilation >= 4 - and the cost model prefers it; never the batched path (needs >= 128 channels)
    assert names[2].startswith("winograd_f2x2_3x3 wgrad fused") or not names[2].startswith("winograd"), names
    # (which of the two the cost model takes at these small sizes is its business - tests/test_wino_wgrad_fused_gpu.py holds
    #  the kernel itself to fp64, tests/test_fullsize_gpu.py asserts it on the model's own stem / layer1 / layer2 geometries)
    wd = w.to(cuda)
    y, st = ops.conv2d_fwd(xs, wd, None, 1, d, d, want_stats=True)
    dx = ops.conv2d_dgrad(dys, wd, tuple(x.shape), 1, d, d)
    seed = torch.randn(x.shape, generator=g).to(cuda)
    acc = seed.clone()
    ops.conv2d_dgrad(dys, wd, tuple(x.shape), 1, d, d, out=acc, accumulate=True)
    dw = ops.conv2d_wgrad(dys, xs, tuple(w.shape), 1, d, d)[0]
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    refdx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
    refdw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, d, d)

    def emax(a, b):
        return float((a.cpu().double() - b).abs().max() / b.abs().max())
    tol = 3e-6 * max(1.0, math.sqrt(9 * Cin) / 8)
    assert emax(y, ref) < tol and emax(dx, refdx) < 3e-6 * max(1.0, math.sqrt(9 * Cout) / 8), (emax(y, ref), emax(dx, refdx))
    assert emax(acc, refdx + seed.cpu().double()) < 3e-6 * max(1.0, math.sqrt(9 * Cout) / 8)
    assert emax(dw, refdw) < 2e-5, emax(dw, refdw)
    if st is not None:
        m_ref, v_ref = ops.bn_stats(y)
        assert float((st[0] - m_ref).abs().max() / m_ref.abs().max()) < 2e-5
        assert float(((st[1] - v_ref).abs() / v_ref).max()) < 2e-5
    # inference: the folded BatchNorm (+residual) (+ReLU) rides on the fused kernel's epilogue (dense input, dilation 4)
    if not pitch:
        sc = (torch.rand(Cout, generator=g) + 0.5).to(cuda); sh = (torch.randn(Cout, generator=g) * 0.2).to(cuda)
        rs = torch.randn(N, Cout, H, W, generator=g).to(cuda)
        yf = ops.conv2d_fused_infer(x.to(cuda), wd, sc, sh, 1, d, d, rs, True)
        reff = torch.relu(ref * sc.cpu().double()[None, :, None, None] + sh.cpu().double()[None, :, None, None] + rs.cpu().double())
        assert emax(yf, reff) < 3e-6


def test_kept_winograd_filters_follow_the_weights(cuda):
    """The fused Winograd kernel's transformed filters are kept per conv (wp_valid) and rebuilt for every conv at once by
    the multi-tensor refresh after an optimizer step (ops.refresh_wp): a second call on unchanged weights must give the
    same bits without transforming again, an in-place weight edit must be noticed, and after a raw-pointer update (what
    FusedSGD does: WEIGHT_EPOCH) + refresh the outputs must be those of the NEW weights - forward and dgrad (flipped taps)."""
    import torch
    import torch.nn.functional as F
    if os.environ.get("DCFP_CONV_WINOGRAD", "1") == "0":
        pytest.skip("Winograd switched off")
    from dcfp_amd import ops, _lib
    g = torch.Generator().manual_seed(31)
    N, Cin, H, W, Cout, d = 2, 96, 96, 160, 200, 4          # (a NARROW shape: dilation 4 reads dense rows)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    dy = torch.randn(N, Cout, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(cuda)
    desc = ops._desc(x.shape, w.shape, 1, d, d)
    assert ops.conv_kernel_name(desc, _lib.CONV_FWD).startswith("winograd_f2x2_3x3")
    L = _lib.lib()
    import ctypes as C
    if L.dcfp_conv2d_workspace_is_scratch(C.byref(desc), _lib.CONV_FWD):
        pytest.skip("DCFP_WINO_KEEP_U=0: the transformed filters are scratch")

    def both():
        return ops.conv2d_fwd(x, w, None, 1, d, d), ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, d, d)

    def ref(wt):
        y = F.conv2d(x.double(), wt.double(), None, 1, d, d)
        dx = torch.nn.grad.conv2d_input(x.shape, wt.double(), dy.double(), 1, d, d)
        return y, dx

    def close(a, b):
        return float((a.double() - b).abs().max() / b.abs().max()) < 1e-5
    y0, dx0 = both()                       # builds the copies (wp_valid = 0)
    y1, dx1 = both()                       # kept copies (wp_valid = 1)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    r = ref(w)
    assert close(y0, r[0]) and close(dx0, r[1])
    w.mul_(1.5)                            # torch sees this edit: the version counter invalidates the copies
    y2, dx2 = both()
    r = ref(w)
    assert close(y2, r[0]) and close(dx2, r[1]) and not torch.equal(y2, y0)
    # a raw-pointer update (FusedSGD writes through the arena): the epoch marks every copy stale, ONE launch rebuilds them
    w.data.view(-1)[::7] += 0.01
    w._version  # (the in-place add above bumped it too; the epoch is what FusedSGD relies on)
    ops.WEIGHT_EPOCH[0] += 1
    ops.refresh_wp()
    y3, dx3 = both()                       # valid again: no transform inside the calls
    r = ref(w)
    assert close(y3, r[0]) and close(dx3, r[1])
    torch.cuda.synchronize()


# What the model still runs on the DIRECT 9-tap kernels with Winograd on (the dilation-36 forward / dgrad, the weight
# gradients of the <= 128-channel layers, pruned widths off the 64-channel grid, stride 2), and what DCFP_CONV_WINOGRAD=0
# (bench.py's `direct_conv` leg) runs everything on: the tests below are run once more in ONE child pytest with the switch
# off - kernel-name assertions there follow the switch, so every wide 3x3 shape in them is then a direct-kernel parity
# case against fp64.  (Until round 3 this re-ran three whole files, 358 s; the whole-iteration-vs-oracle tests, the
# config-3 / config-5 full steps and the property tests in those files do not depend on which conv algorithm ran in a way
# the per-shape cases below do not already cover - DESIGN.md section 4 lists what moved.)
DIRECT_NODES = [
    "tests/test_conv_large_gpu.py::test_large_conv_parity[f32]",                 # 256 x 256 LDS-DMA tiles, ragged M / K / P, d12, +-1 taps
    "tests/test_conv_large_gpu.py::test_row_pitched_operands_bit_identical",      # un-mixed kernels on pitched rows == dense
    "tests/test_conv_large_gpu.py::test_ragged_m_kernel",                         # igemm2_dma8 / wgrad WIDE on pruned widths
    "tests/test_conv_large_gpu.py::test_bottleneck_pitched_path_equals_dense",
    "tests/test_fullsize_gpu.py::test_config3_dgrad_wgrad_vs_fp64_slice",         # config-3 launch geometries, kernel names asserted
    "tests/test_fullsize_gpu.py::test_config5_ragged_shapes_vs_fp64_slice",       # the slim model's shapes
    "tests/test_model_gpu.py::test_forward_backward_vs_reference_golden",         # whole model vs the reference goldens
    "tests/test_model_gpu.py::test_fused_bottleneck_matches_unfused",
]


def test_direct_conv_kernels_still_covered_with_winograd_off(cuda):
    env = dict(os.environ, DCFP_CONV_WINOGRAD="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "--durations=15"] + DIRECT_NODES,
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    log = os.environ.get("DCFP_TEST_LOG_DIR")
    if log:
        with open(os.path.join(log, "winograd_off_child.txt"), "w") as f:
            f.write(r.stdout[-20000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


if __name__ == "__main__" and "--child" in sys.argv:
    _child()
