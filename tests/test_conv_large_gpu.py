"""Conv parity at sizes that reach the 256 x 256-tile kernels (test_ops_gpu.py's shapes are too small
to): the exact-fp32 path (conv_igemm2.hip / conv_wgrad.hip, incl. the interior fast epilogue and the
pipelined accumulate-dgrad) and the opt-in 3-way bf16 split (DCFP_CONV_MATH=bf16x3: conv_igemm3.hip,
conv_wgrad3.hip), both against fp64 CPU convolutions at the SAME tolerances.  The library reads the
switch once, so each mode runs in a child process; routing is asserted through
dcfp_conv2d_kernel_name.  Covered edges: channel counts off the 16/256 grid, pixel counts off the
256 grid, image borders with padding > dilation reach, accumulate-dgrad."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
# the wide 3x3 convs go to the Winograd path unless it is switched off (tests/test_winograd_gpu.py runs this file
# once more with DCFP_CONV_WINOGRAD=0, so the direct kernels named below stay covered)
WINO = os.environ.get("DCFP_CONV_WINOGRAD", "1") != "0"

CASES = [
    # N, Cin, H, W, Cout, k, pad, dil
    (2, 264, 100, 132, 320, 3, 2, 2),     # ragged M (320/264), K (264), P (13200): generic epilogue
    (2, 256, 128, 256, 512, 1, 0, 1),     # interior tiles only: fast epilogue
    (2, 272, 128, 200, 256, 3, 12, 12),   # ASPP-like dilation, whole taps in the padding
    (2, 256, 100, 256, 256, 3, 1, 1),     # unaligned +-1 taps (4 x dword loads on row ends)
]


def _child():
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dcfp_amd import _lib, ops
    dev = torch.device("cuda:0")
    out = []

    def rel(a, b):
        a = a.double().cpu(); b = b.double().cpu()
        return ((a - b).norm() / b.norm()).item()

    for (N, Cin, H, W, Cout, k, p, d) in CASES:
        g = torch.Generator().manual_seed(99)
        x = torch.randn(N, Cin, H, W, generator=g)
        x = torch.relu(x) + 0.05 * x                       # post-ReLU-like statistics
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        desc = ops._desc(x.shape, w.shape, 1, p, d)
        names = [ops.conv_kernel_name(desc, wh) for wh in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)]
        xg, wg = x.to(dev), w.to(dev)
        y = ops.conv2d_fwd(xg, wg, None, 1, p, d)
        dy = torch.randn(y.shape, generator=g) * 1e-2
        dyg = dy.to(dev)
        dx = ops.conv2d_dgrad(dyg, wg, tuple(x.shape), 1, p, d)
        seed = torch.randn(x.shape, generator=g)
        dxa = seed.to(dev)
        ops.conv2d_dgrad(dyg, wg, tuple(x.shape), 1, p, d, out=dxa, accumulate=True)
        dw = ops.conv2d_wgrad(dyg, xg, tuple(w.shape), 1, p, d)[0]
        # inference epilogue: folded eval-mode BatchNorm + residual + ReLU in the conv kernel
        sc = torch.rand(Cout, generator=g) + 0.5
        sh = torch.randn(Cout, generator=g) * 0.3
        resid = torch.randn(y.shape, generator=g)
        yf = ops.conv2d_fused_infer(xg, wg, sc.to(dev), sh.to(dev), 1, p, d, resid.to(dev), True)
        torch.cuda.synchronize()
        # fp64 reference on a slice of output channels (forward / wgrad) or input channels (dgrad)
        mo = slice(Cout - 40, Cout)                        # includes the ragged last M tile
        y64 = F.conv2d(x.double(), w[mo].double(), None, 1, p, d)
        ci = slice(Cin - 24, Cin)
        x64 = x[:, ci].double().requires_grad_(True)
        w64 = w[:, ci].double().requires_grad_(True)
        F.conv2d(x64, w64, None, 1, p, d).backward(dy.double())
        yf64 = torch.relu(y64 * sc[mo].double()[None, :, None, None] + sh[mo].double()[None, :, None, None]
                          + resid[:, mo].double())
        out.append({
            "case": [N, Cin, H, W, Cout, k, p, d], "kernels": names,
            "fwd": rel(y[:, mo], y64), "fused_infer": rel(yf[:, mo], yf64), "dgrad": rel(dx[:, ci], x64.grad),
            "dgrad_acc": rel(dxa[:, ci], x64.grad + seed[:, ci].double()),
            "wgrad": rel(dw[:, ci], w64.grad),
        })
    print("BF16X3_RESULT " + json.dumps(out))


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_large_conv_parity(cuda, mode):
    env = dict(os.environ, DCFP_CONV_MATH=mode)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("BF16X3_RESULT ")][-1]
    res = json.loads(line[len("BF16X3_RESULT "):])
    assert len(res) == len(CASES)
    fwd_kernel, wgrad_kernel = ("igemm3_kernel", "wgrad3_kernel") if mode == "bf16x3" else \
        ("igemm2_", "wgrad2_kernel")     # igemm2_kernel<...>, igemm2_dma_kernel<9,..> or the persistent igemm2_dma1p_kernel
    wino = WINO and mode == "f32"        # (the wide 3x3 cases then take the Winograd weight gradient)
    assert sum(rec["kernels"][2].startswith(wgrad_kernel) or (wino and rec["kernels"][2].startswith("winograd_f2x2_3x3"))
               for rec in res) >= 2, res
    if mode == "f32" and not wino:   # the +-1 tap case goes through the LDS-DMA wgrad with shifted 16-byte copies
        assert res[3]["kernels"][2] == "wgrad_dma_kernel<9,true>", res[3]["kernels"]
    for rec in res:
        N, Cin, H, W, Cout, k, p, d = rec["case"]
        for name in rec["kernels"][:2]:
            assert name.startswith(fwd_kernel) or (WINO and mode == "f32" and name.startswith("winograd_f2x2_3x3")), rec
        K = Cin * k * k
        tol = 3e-6 * max(1.0, math.sqrt(K) / 8)            # as test_conv_fwd_dgrad_wgrad
        assert rec["fwd"] < tol, rec
        assert rec["fused_infer"] < tol, rec
        assert rec["dgrad"] < max(tol, 1e-5), rec
        assert rec["dgrad_acc"] < max(tol, 1e-5), rec
        assert rec["wgrad"] < 2e-5, rec


def _persist_child():
    """1x1 convs through the 256 x 256 LDS-DMA path: digests of every output (forward, forward with fused
    BatchNorm statistics, dgrad, accumulate-dgrad) for the parent to compare between DCFP_IGEMM_PERSIST=0/1."""
    import hashlib
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dcfp_amd import _lib, ops
    dev = torch.device("cuda:0")
    out = {}
    # interior tiles with several tiles per CU; ragged M / K / pixel tail; one image, more tiles than CUs
    for (N, Cin, H, W, Cout) in [(4, 256, 128, 256, 1024), (2, 264, 100, 132, 320), (1, 512, 96, 256, 2048)]:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, Cin, H, W, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev)
        d = ops._desc(x.shape, w.shape, 1, 0, 1)
        want = "igemm2_dma1p_kernel" if os.environ.get("DCFP_IGEMM_PERSIST") != "0" else "igemm2_dma_kernel<1"
        name = ops.conv_kernel_name(d, _lib.CONV_FWD)       # (the ragged 264 -> 320 case goes to the ragged-M kernel either way)
        assert name.startswith(want) or (Cout % 256 and name == "igemm2_dma8_kernel<1>"), name
        y = ops.conv2d_fwd(x, w, None, 1, 0, 1)
        y2, stats = ops.conv2d_fwd(x, w, None, 1, 0, 1, want_stats=True)
        dy = torch.randn(y.shape, generator=g).to(dev)
        dx = ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, 0, 1)
        seed = torch.randn(x.shape, generator=g).to(dev)
        ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, 0, 1, out=seed, accumulate=True)
        torch.cuda.synchronize()
        items = [y, y2, dx, seed] + ([stats[0], stats[1]] if stats is not None else [])
        out[f"{N}x{Cin}x{H}x{W}->{Cout}"] = [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in items]
    print("PERSIST_RESULT " + json.dumps(out))


def test_persistent_1x1_kernel_bit_identical_to_one_tile_kernel(cuda):
    """conv_igemm2p.hip (persistent workgroups, cross-tile prefetch) against igemm2_dma_kernel<1> (one tile per
    workgroup): same tile order and arithmetic, so every output bit must agree."""
    res = []
    for v in ("0", "1"):
        env = dict(os.environ, DCFP_IGEMM_PERSIST=v, DCFP_CONV_MATH="f32")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--persist-child"], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("PERSIST_RESULT ")][-1]
        res.append(json.loads(line[len("PERSIST_RESULT "):]))
    assert res[0].keys() == res[1].keys() and len(res[0]) == 3
    for k in res[0]:
        assert res[0][k] == res[1][k], k
    assert any(len(v) == 6 for v in res[0].values())        # the fused-statistics epilogue was exercised


@pytest.mark.parametrize("shape", [(3, 256, 64, 256, 256, 2), (2, 256, 128, 256, 512, 1), (4, 320, 96, 128, 264, 2)])
def test_row_pitched_operands_bit_identical(cuda, shape):
    """3x3 convs with dilation 1 / 2 reading row-pitched x / dy (zero tail behind each row: DcfpConvDesc.x_pitch,
    dy_pitch) against the same convs on dense tensors: the un-mixed LDS-DMA kernels copy every shifted quad
    as is, the dense path handles image borders with 4-byte copies - same products, same order, same bits.
    Also the BatchNorm kernels that produce the pitched tensors (y_pitch / dx_pitch)."""
    import torch
    from dcfp_amd import ops, _lib
    N, Cin, H, W, Cout, d = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(cuda)
    dy = torch.randn(N, Cout, H, W, generator=g).to(cuda)
    pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), 1, d, d)
    assert pitch >= W + d and pitch % 4 == 0
    desc = ops._desc(x.shape, w.shape, 1, d, d, pitch, pitch)
    names = [ops.conv_kernel_name(desc, k) for k in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)]
    # un-mixed LDS-DMA kernels (the ragged-M one where the channel count is off the 256 grid); same K order everywhere
    assert all(n in ("igemm2_dma_kernel<9,false>", "igemm2_dma8_kernel<9>") or (WINO and n.startswith("winograd_f2x2_3x3"))
               for n in names[:2]), names
    assert names[2] in ("wgrad_dma_kernel<9,false>", "wgrad_dma_kernel<9,false,true>") or \
        (WINO and names[2].startswith("winograd_f2x2_3x3")), names
    xp = ops.pitched_buffer(tuple(x.shape), pitch, "test_x", cuda); xp.copy_(x)
    dyp = ops.pitched_buffer(tuple(dy.shape), pitch, "test_dy", cuda); dyp.copy_(dy)
    assert ops._pitch_of(xp) == pitch and float(xp.as_strided((N, Cin, H, pitch - W), xp.stride(), xp.storage_offset() + W).abs().sum()) == 0.0
    y0, st0 = ops.conv2d_fwd(x, w, None, 1, d, d, want_stats=True)
    y1, st1 = ops.conv2d_fwd(xp, w, None, 1, d, d, want_stats=True)
    assert torch.equal(y0, y1)
    if st0 is not None:    # the fused BatchNorm statistics group pixels by tile (8 x 32 vs 256 x 1 tiles): same values, other order
        assert st1 is not None
        assert (st0[0] - st1[0]).abs().max().item() <= 2e-6 * max(1.0, st0[0].abs().max().item())
        assert ((st0[1] - st1[1]).abs() / st0[1]).max().item() <= 2e-6
    assert torch.equal(ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, d, d), ops.conv2d_dgrad(dyp, w, tuple(x.shape), 1, d, d))
    dw_d, dw_p = ops.conv2d_wgrad(dy, x, tuple(w.shape), 1, d, d)[0], ops.conv2d_wgrad(dyp, xp, tuple(w.shape), 1, d, d)[0]
    if names[2] == ops.conv_kernel_name(ops._desc(x.shape, w.shape, 1, d, d), _lib.CONV_WGRAD):
        assert torch.equal(dw_d, dw_p)
    elif WINO:     # pitched operands take the fused Winograd weight gradient, dense ones the batched one (or a direct kernel):
        # two summation orders of the same products (tests/test_winograd_gpu.py holds each against fp64)
        assert names[2].startswith("winograd_f2x2_3x3 wgrad fused"), names
        assert ((dw_d - dw_p).norm() / dw_d.norm()).item() < 2e-6
    else:         # direct kernels: the un-mixed LDS-DMA kernel on pitched rows, the mixed one on dense rows - the same sums
        assert torch.equal(dw_d, dw_p)
    # producers: BN(+ReLU) forward into a pitched buffer, BN backward dx into a pitched buffer
    gamma = (torch.rand(Cin, generator=g) + 0.5).to(cuda); beta = (torch.randn(Cin, generator=g) * 0.2).to(cuda)
    mean, var = ops.bn_stats(x)
    yd = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, None, True)
    yp = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, None, True, out=ops.pitched_buffer(tuple(x.shape), pitch, "test_y", cuda))
    assert torch.equal(yd, yp) and float(yp.as_strided((N, Cin, H, pitch - W), yp.stride(), yp.storage_offset() + W).abs().sum()) == 0.0
    gin = torch.randn(x.shape, generator=g).to(cuda)
    s1, s2, _ = ops.bn_bwd_reduce(gin, x, None, mean, var, gamma, beta, 1e-5, 2)
    cnt = float(N * H * W)
    dxd, _ = ops.bn_bwd_apply(gin, x, None, mean, var, gamma, beta, 1e-5, s1, s2, cnt, 2, False)
    dxp, _ = ops.bn_bwd_apply(gin, x, None, mean, var, gamma, beta, 1e-5, s1, s2, cnt, 2, False,
                              dx_out=ops.pitched_buffer(tuple(x.shape), pitch, "test_dx", cuda))
    assert torch.equal(dxd, dxp)


RAGGED = [
    # N, Cin, H, W, Cout, k, pad, dil, pitched, (fwd kernel, dgrad kernel)
    ((4, 256, 128, 256, 154, 1, 0, 1, False), ("igemm2_dma8_kernel<1>", None)),            # pruned 1x1: 5 of 8 row blocks live
    ((2, 256, 128, 256, 154, 3, 4, 4, False), ("igemm2_dma8_kernel<9>", None)),            # 3x3, column shifts multiples of 4
    ((2, 83, 128, 256, 256, 3, 12, 12, False), (None, "igemm2_dma8_kernel<9>")),           # dgrad with M = Cin = 83
    ((3, 256, 64, 256, 150, 3, 2, 2, True), ("igemm2_dma8_kernel<9>", None)),              # dilation 2 on a row-pitched source
    ((2, 128, 128, 256, 300, 1, 0, 1, False), ("igemm2_dma8_kernel<1>", None)),            # two M tiles: 8 + 2 live row blocks
    ((4, 2048, 96, 128, 83, 3, 12, 12, False), ("igemm2_dma8_kernel<9>", None)),           # pruned ASPP branch: M = 83, long K
    ((4, 512, 128, 256, 19, 1, 0, 1, False), ("igemm2_dma8_kernel<1>", None)),             # the 19-class classifier (bias)
    ((4, 2048, 96, 128, 40, 3, 24, 24, False), ("igemm2_dma8_kernel<9>", None)),           # ragged M <= 64
]


@pytest.mark.parametrize("case,kernels", RAGGED)
def test_ragged_m_kernel(cuda, case, kernels):
    """conv_igemm2n.hip (8 x 2 wave tiles, dead row blocks skipped, permuted weight rows) on pruned widths:
    forward / dgrad / accumulate-dgrad against fp64 on channel slices, routing asserted."""
    import torch
    import torch.nn.functional as F
    from dcfp_amd import ops, _lib
    N, Cin, H, W, Cout, k, p, d, pitched = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = (torch.relu(x) + 0.05 * x).to(cuda)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(cuda)
    dy = (torch.randn(N, Cout, H, W, generator=g) * 1e-2).to(cuda)
    xin, dyin, pitch = x, dy, 0
    if pitched:
        pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), 1, p, d)
        assert pitch
        xin = ops.new_pitched(tuple(x.shape), pitch, cuda); xin.copy_(x)
        dyin = ops.new_pitched(tuple(dy.shape), pitch, cuda); dyin.copy_(dy)
    desc = ops._desc(x.shape, w.shape, 1, p, d, pitch, pitch)
    for want, which in zip(kernels, (_lib.CONV_FWD, _lib.CONV_DGRAD)):
        if want is not None:      # (ragged 3x3 shapes the fused Winograd kernel takes since round 3: 64-row blocks; the
            name = ops.conv_kernel_name(desc, which)     # direct ragged-M kernel is covered with DCFP_CONV_WINOGRAD=0)
            assert name == want or (WINO and k == 3 and name.startswith("winograd_f2x2_3x3")), name
    y = ops.conv2d_fwd(xin, w, None, 1, p, d)
    y_again = ops.conv2d_fwd(xin, w, None, 1, p, d)              # second call: cached (permuted) Wp, wp_valid = 1
    assert torch.equal(y, y_again)
    bias = torch.randn(Cout, generator=g).to(cuda)               # the classifiers' bias goes through the same epilogue
    yb = ops.conv2d_fwd(xin, w, bias, 1, p, d)
    # (same kernel for both where the conv is direct: 1e-6; a 3x3 conv without bias may run as Winograd since round 3,
    #  with bias never - two fp32 algorithms then, each within the stated conv tolerance of the exact result)
    same = ops.conv_kernel_name(desc, _lib.CONV_FWD).startswith("igemm2")
    btol = 1e-6 if same else 2 * 3e-6 * max(1.0, math.sqrt(Cin * k * k) / 8)
    assert (yb - (y + bias.view(1, -1, 1, 1))).abs().max().item() <= btol * max(1.0, yb.abs().max().item())
    dx = ops.conv2d_dgrad(dyin, w, tuple(x.shape), 1, p, d)
    seed = torch.randn(x.shape, generator=g).to(cuda)
    dxa = seed.clone()
    ops.conv2d_dgrad(dyin, w, tuple(x.shape), 1, p, d, out=dxa, accumulate=True)
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double() - b).norm() / b.norm()).item()
    mo = slice(max(0, Cout - 40), Cout)                             # the ragged last row blocks
    y64 = F.conv2d(x.double(), w[mo].double(), None, 1, p, d)
    ci = slice(max(0, Cin - 24), Cin)
    x64 = x[:, ci].double().requires_grad_(True)
    F.conv2d(x64, w[:, ci].double(), None, 1, p, d).backward(dy.double())
    K = Cin * k * k
    tol = 3e-6 * max(1.0, math.sqrt(K) / 8)
    assert rel(y[:, mo], y64) < tol, rel(y[:, mo], y64)
    y64b = F.conv2d(x.double(), w[:8].double(), None, 1, p, d)      # first row block too
    assert rel(y[:, :8], y64b) < tol
    assert rel(dx[:, ci], x64.grad) < max(tol, 1e-5), rel(dx[:, ci], x64.grad)
    assert rel(dxa[:, ci], x64.grad + seed[:, ci].double()) < max(tol, 1e-5)
    # weight gradient: ragged Cout goes to the WIDE layout of the LDS-DMA wgrad kernel (dead dy row blocks skipped)
    # (or, for a 3x3 conv, to the fused Winograd weight gradient, whose blocks are 64 x 64 channels - round 4; the WIDE
    #  kernel stays covered by the 1x1 cases here and by the DCFP_CONV_WINOGRAD=0 pass of tests/test_winograd_gpu.py)
    if Cout % 256 and Cin * k * k > 128 and W % 16 == 0:
        wname = ops.conv_kernel_name(desc, _lib.CONV_WGRAD)
        assert wname == f"wgrad_dma_kernel<{k * k},false,true>" or \
            (WINO and k == 3 and wname.startswith("winograd_f2x2_3x3 wgrad fused")), wname
    dw = ops.conv2d_wgrad(dyin, xin, tuple(w.shape), 1, p, d)[0]
    w64 = w[:, ci].double().requires_grad_(True)
    F.conv2d(x[:, ci].double(), w64, None, 1, p, d).backward(dy.double())
    assert rel(dw[:, ci], w64.grad) < 2e-5, rel(dw[:, ci], w64.grad)


def test_bottleneck_pitched_path_equals_dense(cuda):
    """One dilation-2 Bottleneck at a size where conv2 goes row-pitched (256 channels, 4 x 64 x 256 pixels),
    forward + backward, against the same block with DCFP_PITCHED off (ops.PITCHED): every output and
    gradient bit for bit; a second forward before the first backward must not clobber the saved y1."""
    import torch
    from dcfp_amd import ops
    from dcfp_amd.networks.backbone.resnet import Bottleneck

    def run(pitched):
        ops.PITCHED = pitched
        fuse, ops.FUSE_BN_STATS = ops.FUSE_BN_STATS, False   # (fused statistics group pixels by tile shape: 8 x 32 vs 256 x 1)
        try:
            torch.manual_seed(7)
            blk = Bottleneck(1024, 256, stride=1, dilation=2).to(cuda).train()
            gx = torch.Generator().manual_seed(8)
            x = torch.randn(4, 1024, 64, 256, generator=gx).to(cuda).requires_grad_(True)
            out = blk(x)
            out2 = blk(x)                      # second graph while the first one is alive
            gy = torch.randn(out.shape, generator=gx).to(cuda)
            out.backward(gy)
            grads = [p.grad.clone() for p in blk.parameters()] + [x.grad.clone()]
            return out.detach().clone(), out2.detach().clone(), grads, getattr(blk, "_dcfp_pitch", None) is not None
        finally:
            ops.PITCHED = True
            ops.FUSE_BN_STATS = fuse
    o_p, o2_p, g_p, used = run(True)
    o_d, o2_d, g_d, _ = run(False)
    assert used, "conv2 of this block should have taken the row-pitched path"
    assert torch.equal(o_p, o_d) and torch.equal(o2_p, o2_d)
    for a, b in zip(g_p, g_d):
        assert torch.equal(a, b)



# (N, Cin, H, W, Cout, dilation): ASPP-like geometry - the dilation is comparable to the image height, so whole kernel
# rows fall into the padding for the tiles near the top / bottom edge.  One-row tiles (W = 256), two-row tiles,
# a ragged last tile (P % 256 != 0), shifted (dilation % 4 != 0) taps on linear and on 8 x 32 tiles, and a
# dilation larger than the image (only the centre row is ever live).
TAPSKIP_SHAPES = [(2, 48, 96, 256, 512, 24), (2, 48, 96, 256, 512, 36), (6, 40, 64, 128, 256, 12),
                  (8, 32, 50, 64, 512, 16), (3, 32, 44, 256, 512, 6), (3, 32, 64, 64, 1024, 18),
                  (3, 32, 32, 256, 512, 40), (2, 32, 96, 256, 300, 24)]


def _tapskip_child():
    """Dilated 3x3 convs through the 9-tap LDS-DMA kernels: digests of forward, forward + BatchNorm statistics,
    dgrad, accumulate-dgrad and wgrad for the parent to compare between DCFP_IGEMM_TAPSKIP / DCFP_WGRAD_TAPSKIP = 0 / 1."""
    import hashlib
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dcfp_amd import _lib, ops
    dev = torch.device("cuda:0")
    out = {}
    for (N, Cin, H, W, Cout, d) in TAPSKIP_SHAPES:
        g = torch.Generator().manual_seed(9)
        x = torch.randn(N, Cin, H, W, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev)
        desc = ops._desc(x.shape, w.shape, 1, d, d)
        name = ops.conv_kernel_name(desc, _lib.CONV_FWD)
        assert name.startswith("igemm2_dma_kernel<9") or name.startswith("igemm2_dma8_kernel<9"), name
        y = ops.conv2d_fwd(x, w, None, 1, d, d)
        r = ops.conv2d_fwd(x, w, None, 1, d, d, want_stats=True)
        y2, stats = r if isinstance(r, tuple) else (r, None)
        dy = torch.randn(y.shape, generator=g).to(dev)
        dx = ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, d, d)
        seed = torch.randn(x.shape, generator=g).to(dev)
        ops.conv2d_dgrad(dy, w, tuple(x.shape), 1, d, d, out=seed, accumulate=True)
        dw, _ = ops.conv2d_wgrad(dy, x, tuple(w.shape), 1, d, d)
        torch.cuda.synchronize()
        items = [y, y2, dx, seed] + ([stats[0], stats[1]] if stats is not None else [])
        # reference: fp64 on the CPU (the skipped K-steps must not change the result at all, and the result is right)
        ref = torch.nn.functional.conv2d(x.cpu().double(), w.cpu().double(), None, 1, d, d)
        err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        refw = torch.nn.grad.conv2d_weight(x.cpu().double(), w.shape, dy.cpu().double(), 1, d, d)
        errw = float((dw.cpu().double() - refw).abs().max() / refw.abs().max())
        out[f"{N}x{Cin}x{H}x{W}->{Cout} d{d}"] = {"hash": [hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in items],
                                                  "wgrad_hash": hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest(),
                                                  "err": err, "errw": errw, "kernel": name,
                                                  "wkernel": ops.conv_kernel_name(desc, _lib.CONV_WGRAD)}
    print("TAPSKIP_RESULT " + json.dumps(out))


def test_dead_tap_rows_skipped_without_changing_a_bit(cuda):
    """igemm2_dma_kernel<9, ...> skips the K-steps of kernel rows that lie wholly in the padding for a tile
    (ASPP dilations 12 / 24 / 36 on 128 rows: 6 / 12 / 19 % of the MFMAs).  The remaining K-steps keep their order,
    so forward, fused statistics, dgrad and accumulate-dgrad must be the same bits as with DCFP_IGEMM_TAPSKIP=0;
    the weight-gradient kernel trims each tap's pixel range instead (a different split of the same sum: compared
    with the fp64 reference at the usual tolerance, and both settings against it)."""
    res = []
    for v in ("0", "1"):
        env = dict(os.environ, DCFP_IGEMM_TAPSKIP=v, DCFP_WGRAD_TAPSKIP=v, DCFP_CONV_MATH="f32")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--tapskip-child"], env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("TAPSKIP_RESULT ")][-1]
        res.append(json.loads(line[len("TAPSKIP_RESULT "):]))
    assert res[0].keys() == res[1].keys() and len(res[0]) == len(TAPSKIP_SHAPES)
    for k in res[0]:
        assert res[0][k]["hash"] == res[1][k]["hash"], (k, res[0][k]["kernel"])
        assert res[1][k]["err"] < 2e-6 and res[1][k]["errw"] < 2e-5, (k, res[1][k])
        assert res[0][k]["errw"] < 2e-5, (k, res[0][k])
    assert any(len(v["hash"]) == 6 for v in res[0].values())        # the fused-statistics epilogue was exercised

if __name__ == "__main__" and "--child" in sys.argv:
    _child()
if __name__ == "__main__" and "--persist-child" in sys.argv:
    _persist_child()
if __name__ == "__main__" and "--tapskip-child" in sys.argv:
    _tapskip_child()
