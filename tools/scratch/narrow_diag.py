import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dcfp_amd import ops, _lib
dev = torch.device("cuda:0")
m = bench.build_model("resnet101", dev)
x, lab = bench.synthetic_batch(4, 1024, 2048, 1, dev)
for _ in range(2):
    for p in m.parameters(): p.grad = None
    m(x, lab, deepsup=True)["loss"].backward()
ops.profile_start()
for p in m.parameters(): p.grad = None
m(x, lab, deepsup=True)["loss"].backward()
recs = ops.profile_stop()
which = {"conv_fwd": _lib.CONV_FWD, "conv_dgrad": _lib.CONV_DGRAD, "conv_wgrad": _lib.CONV_WGRAD}
for kind, d, work, ms in recs:
    if kind in which and d.KH == 3 and max(d.Cin, d.Cout) <= 128:
        print(kind, (d.N, d.Cin, d.H, d.W, d.Cout, d.stride, d.dil, d.x_pitch, d.dy_pitch), ops.conv_kernel_name(d, which[kind]), f"{ms:.3f} ms {work/ms/1e9:.1f} TF")
