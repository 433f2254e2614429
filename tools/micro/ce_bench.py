"""Time of the fused upsample+CE forward / backward at the bench shape (4x19x128x256 -> 1024x2048)."""
import sys, torch
sys.path.insert(0, ".")
from dcfp_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
z = (torch.randn(4, 19, 128, 256, generator=g) * 2).to(dev).requires_grad_(True)
lab = torch.randint(0, 19, (4, 1024, 2048), generator=g)
lab[torch.rand(4, 1024, 2048, generator=g) < 0.05] = 255
lab = lab.to(dev)
def run():
    z.grad = None
    l = ops.upsample_cross_entropy(z, lab, (1024, 2048), True, 255)
    l.backward()
run(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize(); print("upsample+CE fwd+bwd ms", s.elapsed_time(e) / 10)
