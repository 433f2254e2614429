import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcfp_amd import ops
from tools.conv_bench import SHAPES, bench
dev = torch.device("cuda:0")
for name in sys.argv[1].split(","):
    N, Cin, H, W, Cout, k, s, p, d = SHAPES[name]
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5
    dy = torch.randn(N, Cout, H, W, device=dev)
    pitch = W + 4
    xp = ops.new_pitched(tuple(x.shape), pitch, dev); xp.copy_(x)
    dp = ops.new_pitched(tuple(dy.shape), pitch, dev); dp.copy_(dy)
    y = torch.empty(N, Cout, H, W, device=dev); dx = torch.empty(N, Cin, H, W, device=dev)
    desc = ops._desc(x.shape, w.shape, s, p, d, pitch, pitch)
    tf = bench(lambda: ops.conv2d_fwd(xp, w, None, s, p, d, out=y), 10)
    td = bench(lambda: ops.conv2d_dgrad(dp, w, tuple(x.shape), s, p, d, out=dx), 10)
    yr = torch.nn.functional.conv2d(x, w, None, s, p, d)
    err = ((y - yr).norm() / yr.norm()).item()
    fl = 2.0 * N * Cout * H * W * Cin * 9
    print(f"{name:12s} {ops.conv_kernel_name(desc, 0)[:40]:40s} fwd {tf:.3f} ms {fl/tf/1e9:6.1f} TF  dgrad {td:.3f} ms {fl/td/1e9:6.1f} TF  rel err {err:.2e}", flush=True)
