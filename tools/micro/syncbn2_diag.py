"""Two gloo ranks on one GPU: SyncBN forward/backward at op level against the full batch (diagnostic)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(rank):
    sys.path.insert(0, ROOT)
    import torch, torch.distributed as dist
    from dcfp_amd import ops
    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    g = torch.Generator().manual_seed(3)
    C = 64
    X = torch.randn(4, C, 24, 32, generator=g).to(dev)
    DY = torch.randn(4, C, 24, 32, generator=g).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev); beta = torch.randn(C, generator=g).to(dev)
    # 1. raw async all_reduce on a device view
    s = torch.full((2, C), float(rank + 1), device=dev)
    both = s[0].as_strided((2 * C,), (1,), s[0].storage_offset())
    w = dist.all_reduce(both, async_op=True); w.wait()
    torch.cuda.synchronize()
    r1 = float((s - 3.0).abs().max())
    # 2. sync_bn_bwd_sums
    s = torch.full((2, C), float(rank + 1), device=dev)
    a, b, w = ops.sync_bn_bwd_sums(s[0], s[1], dist.group.WORLD, async_op=True); w.wait()
    torch.cuda.synchronize()
    r2 = float((a - 3.0).abs().max()), float((b - 3.0).abs().max())
    # 3. BN forward/backward op level
    def run(x, dy, sync):
        x = x.clone().requires_grad_(True); ga = gamma.clone().requires_grad_(True); be = beta.clone().requires_grad_(True)
        rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
        y = ops.BatchNormActFn.apply(x, ga, be, rm, rv, None, True, True, 0.1, 1e-5, sync, None, None)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach(), x.grad, ga.grad, be.grad
    full = run(X, DY, False)
    part = run(X[2 * rank:2 * rank + 2].contiguous(), DY[2 * rank:2 * rank + 2].contiguous(), True)
    def rel(a, b): return float((a - b).norm() / b.norm())
    out = {"rank": rank, "raw": r1, "sums": r2, "y": rel(part[0], full[0][2 * rank:2 * rank + 2]),
           "dx": rel(part[1], full[1][2 * rank:2 * rank + 2])}
    gs = [part[2].clone(), part[3].clone()]
    for t in gs: dist.all_reduce(t)
    out["dgamma"] = rel(gs[0], full[2]); out["dbeta"] = rel(gs[1], full[3])
    print("DIAG " + json.dumps(out), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)]) for r in range(2)]
        sys.exit(max(p.wait() for p in ps))
