"""Two gloo ranks on one GPU: a conv-BN-ReLU-conv head through run_sequential, SyncBN with / without the
DataParallel wrapper, against the full batch (diagnostic)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(rank):
    sys.path.insert(0, ROOT)
    import torch, torch.nn as nn, torch.distributed as dist
    from dcfp_amd import ops
    from dcfp_amd.networks import _exec
    from dcfp_amd.engine import DataParallel
    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29563", RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(4, 32, 17, 33, generator=g).to(dev)
    T = torch.randn(4, 8, 17, 33, generator=g).to(dev)

    class Head(nn.Module):
        def __init__(self):
            super().__init__()
            self.seq = nn.Sequential(nn.Conv2d(32, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                     nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                     nn.Conv2d(64, 8, 1, bias=True))
        def forward(self, x):
            return _exec.run_sequential(self.seq, x)

    def run(mode):
        torch.manual_seed(7)
        m = Head().to(dev).train()
        if mode == "full":
            x, t = X, T
        else:
            m = nn.SyncBatchNorm.convert_sync_batchnorm(m)
            x, t = X[2 * rank:2 * rank + 2].contiguous(), T[2 * rank:2 * rank + 2].contiguous()
        model = DataParallel(m) if mode == "dp" else m
        x = x.clone().requires_grad_(True)
        y = model(x)
        loss = (y * t).sum() / y.numel()
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        if mode == "sync":
            for v in grads.values():
                dist.all_reduce(v); v.div_(2)
        return grads, x.grad.detach().clone(), float(loss)
    full = run("full")
    out = {"rank": rank}
    def rel(a, b): return float((a - b).norm() / b.norm())
    for mode in ("sync", "dp"):
        r = run(mode)
        out[mode] = {k: "%.1e" % rel(r[0][k], full[0][k]) for k in full[0]}
        out[mode]["dx"] = "%.1e" % rel(r[1] / 2, full[1][2 * rank:2 * rank + 2])
        out[mode]["loss"] = [r[2], full[2]]
    print("DIAG " + json.dumps(out), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)]) for r in range(2)]
        sys.exit(max(p.wait() for p in ps))
