"""Time of the accumulate-dgrad (residual-gradient fan-in) on the layer3 conv1 shape."""
import sys, torch
sys.path.insert(0, ".")
from dcfp_amd import ops
dev = torch.device("cuda:0")
dy = torch.randn(4, 256, 128, 256, device=dev); w = torch.randn(256, 1024, 1, 1, device=dev) * 0.03
out = torch.randn(4, 1024, 128, 256, device=dev)
def run(): ops.conv2d_dgrad(dy, w, (4, 1024, 128, 256), 1, 0, 1, out=out, accumulate=True)
for _ in range(30): run()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize(); print("acc dgrad l3c1 ms", s.elapsed_time(e) / 50)
