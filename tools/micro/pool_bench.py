"""Time of max-pool 3x3/2 forward + backward on the stem shape (4 x 128 x 512 x 1024)."""
import sys, torch
sys.path.insert(0, ".")
from dcfp_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(4, 128, 512, 1024, device=dev).requires_grad_(True)
y = ops.maxpool3x3s2(x)
dy = torch.randn_like(y)
def bwd():
    x.grad = None
    y.backward(dy, retain_graph=True)
for _ in range(5): bwd()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): bwd()
e.record(); torch.cuda.synchronize(); print("maxpool bwd ms", s.elapsed_time(e) / 20)
