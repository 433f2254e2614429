import torch, sys
sys.path.insert(0, ".")
from dcfp_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(4, 3, 1024, 2048, device=dev); w = torch.randn(64, 3, 3, 3, device=dev) * 0.2
y = ops.conv2d_fwd(x, w, None, 2, 1, 1); dy = torch.randn_like(y)
def bench(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
print("stem fwd   %.3f ms" % bench(lambda: ops.conv2d_fwd(x, w, None, 2, 1, 1)))
print("stem fwd+stats %.3f ms" % bench(lambda: ops.conv2d_fwd(x, w, None, 2, 1, 1, want_stats=True)))
print("stem wgrad %.3f ms" % bench(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), 2, 1, 1)))
