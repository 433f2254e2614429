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
    assert sum(rec["kernels"][2].startswith(wgrad_kernel) for rec in res) >= 2, res
    if mode == "f32":   # the +-1 tap case goes through the LDS-DMA wgrad with shifted 16-byte copies
        assert res[3]["kernels"][2] == "wgrad_dma_kernel<9,true>", res[3]["kernels"]
    for rec in res:
        N, Cin, H, W, Cout, k, p, d = rec["case"]
        assert rec["kernels"][0].startswith(fwd_kernel), rec
        assert rec["kernels"][1].startswith(fwd_kernel), rec
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
        assert ops.conv_kernel_name(d, _lib.CONV_FWD).startswith(want), ops.conv_kernel_name(d, _lib.CONV_FWD)
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


if __name__ == "__main__" and "--child" in sys.argv:
    _child()
if __name__ == "__main__" and "--persist-child" in sys.argv:
    _persist_child()
