"""Full-size accuracy of the fused upsample+CE (4x19x128x256 -> 1024x2048) against fp64 torch on the CPU."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from dcfp_amd import ops
g = torch.Generator().manual_seed(0)
z = torch.randn(4, 19, 128, 256, generator=g) * 2
lab = torch.randint(0, 19, (4, 1024, 2048), generator=g)
lab[torch.rand(4, 1024, 2048, generator=g) < 0.05] = 255
zr = z.double().requires_grad_(True)
loss = F.cross_entropy(F.interpolate(zr, size=(1024, 2048), mode="bilinear", align_corners=True), lab, ignore_index=255)
loss.backward()
zg = z.cuda().requires_grad_(True)
lg = ops.upsample_cross_entropy(zg, lab.cuda(), (1024, 2048), True, 255)
lg.backward()
print("loss", lg.item(), loss.item(), "rel grad err", ((zg.grad.double().cpu() - zr.grad).norm() / zr.grad.norm()).item(),
      "max abs", (zg.grad.double().cpu() - zr.grad).abs().max().item(), "grad max", zr.grad.abs().max().item())
