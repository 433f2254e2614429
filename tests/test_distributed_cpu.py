"""CPU, world_size 2 over gloo: the host side of the data-parallel path — Engine rendezvous,
loss all-reduce (utils/pyt_utils.py:34-40), the SyncBN statistics exchange of
dcfp_amd.ops (one all_gather forward, one all_reduce backward) and gradient averaging giving
rank-identical BN-gamma gradients (what makes the EIC score identical on every rank)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    import argparse
    from dcfp_amd.engine import Engine
    from dcfp_amd import ops
    sys.argv = ["x"]
    eng = Engine(custom_parser=argparse.ArgumentParser(), backend="gloo")
    assert eng.distributed and eng.world_size == world and eng.local_rank == rank
    # loss all-reduce
    t = eng.all_reduce_tensor(torch.tensor(float(rank + 1)))
    assert abs(t.item() - 1.5) < 1e-7
    # SyncBN forward statistics: each rank holds a shard with a different batch size
    g = torch.Generator().manual_seed(3)
    full = torch.randn(5, 7, 6, 4, generator=g) * 2 + 1
    shard = full[:2] if rank == 0 else full[2:]
    mean = shard.mean((0, 2, 3)); var = shard.var((0, 2, 3), unbiased=False)
    gm, gv, cnt = ops.sync_bn_stats(mean, var, shard.numel() // 7, dist.group.WORLD)
    assert float(cnt) == full.numel() // 7
    assert torch.allclose(gm, full.mean((0, 2, 3)), atol=1e-6)
    assert torch.allclose(gv, full.var((0, 2, 3), unbiased=False), atol=1e-5)
    # backward sums
    s1, s2 = ops.sync_bn_bwd_sums(torch.full((7,), float(rank + 1)), torch.arange(7.) * (rank + 1), dist.group.WORLD)
    assert torch.allclose(s1, torch.full((7,), 3.0)) and torch.allclose(s2, torch.arange(7.) * 3)
    # DDP-style gradient averaging -> rank-identical BN-gamma grads -> rank-identical EIC (oracle formula)
    from oracle import scoring
    gamma = torch.linspace(-1, 1, 16)
    grad = torch.cos(torch.arange(16.) + rank)
    dist.all_reduce(grad); grad /= world
    eic = scoring.eic_step(gamma.numpy(), grad.numpy(), 0, 0.999)
    gathered = [torch.zeros(16) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(eic))
    assert torch.equal(gathered[0], gathered[1])
    _bn_shards_match_pooled_oracle(rank, world)
    _arena_reducer_mean_of_shard_means(rank, world)
    # max-over-ranks timing reduction used by bench.py
    tt = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.item() == world - 1
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def _bn_shards_match_pooled_oracle(rank, world):
    """Full BatchNorm(+ReLU) forward AND backward on unequal shards, every cross-rank quantity through the
    product's exchange (ops.sync_bn_stats / ops.sync_bn_bwd_sums, the code engine.py:65's SyncBatchNorm
    maps to), against F.batch_norm + autograd on the pooled batch in one process."""
    import torch.nn.functional as F
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(11)
    Cc, eps, mom = 5, 1e-5, 0.1
    full = torch.randn(6, Cc, 4, 3, generator=g) * 1.7 + 0.4
    dy_full = torch.randn(6, Cc, 4, 3, generator=g)
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g) * 0.3
    sl = slice(0, 2) if rank == 0 else slice(2, 6)
    x, dy = full[sl], dy_full[sl]
    rm, rv, nbt = torch.zeros(Cc), torch.ones(Cc), torch.tensor(0)
    mean = x.mean((0, 2, 3)); var = x.var((0, 2, 3), unbiased=False)
    gm, gv, cnt = ops.sync_bn_stats(mean, var, x.numel() // Cc, dist.group.WORLD, running=(rm, rv, mom, nbt))
    istd = torch.rsqrt(gv + eps)
    bc = lambda v: v.view(1, Cc, 1, 1)
    y = torch.relu((x - bc(gm)) * bc(istd) * bc(gamma) + bc(beta))
    gg = dy * (y > 0)
    s1 = gg.sum((0, 2, 3)); s2 = (gg * (x - bc(gm))).sum((0, 2, 3))
    dgamma_local, dbeta_local = s2 * istd, s1.clone()
    r1, r2, work = ops.sync_bn_bwd_sums(s1.clone(), s2.clone(), dist.group.WORLD, async_op=True)
    work.wait()
    M = float(cnt)
    dx = bc(gamma * istd) * (gg - bc(r1) / M - (x - bc(gm)) * bc(istd * istd * r2) / M)
    # oracle: one process, the pooled batch
    xf = full.clone().requires_grad_(True); gf = gamma.clone().requires_grad_(True); bf = beta.clone().requires_grad_(True)
    orm, orv = torch.zeros(Cc), torch.ones(Cc)
    yf = torch.relu(F.batch_norm(xf, orm, orv, gf, bf, True, mom, eps))
    (yf * dy_full).sum().backward()
    assert torch.allclose(y, yf[sl].detach(), atol=1e-5)
    assert torch.allclose(dx, xf.grad[sl], atol=2e-5)
    assert torch.allclose(rm, orm, atol=1e-6) and torch.allclose(rv, orv, atol=1e-5) and int(nbt) == 1
    # gamma / beta gradients stay per-rank sums; the gradient exchange adds them up (SUM here, /world in DDP)
    dist.all_reduce(dgamma_local); dist.all_reduce(dbeta_local)
    assert torch.allclose(dgamma_local, gf.grad, atol=2e-5) and torch.allclose(dbeta_local, bf.grad, atol=2e-5)


def _arena_reducer_mean_of_shard_means(rank, world):
    """arena.GradReducer: gradients written straight into the flat arena during backward, averaged over
    the ranks by 3 chunked all-reduces issued as their ranges complete.  Loss semantics of the reference
    (train.py:259-268 + DDP): each rank's CE is a mean over ITS OWN valid pixels, gradients are averaged
    over ranks - so the oracle is one process with loss = mean over shards of the per-shard CE."""
    import torch.nn.functional as F
    from dcfp_amd.arena import ParamArena, GradReducer, grad_target, grad_commit
    g = torch.Generator().manual_seed(3)
    shapes = [(4, 3, 1, 1), (4,), (300,), (17, 5)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ps]
    feats = torch.randn(6, 3, 5, 5, generator=g)
    labels = torch.randint(0, 4, (6, 5, 5), generator=g)
    labels[0, :3] = 255; labels[4, 2:] = 255                     # different valid counts per shard
    shards = [slice(0, 2), slice(2, 6)]
    arena = ParamArena(ps)
    red = GradReducer(arena, None, 3)
    assert len(red.bounds) == 3 and red.bounds[0][2] == 0 and red.bounds[-1][3] == arena.total
    # arena identity is a matter of the parameter SET: the optimizer walks the parameters by param group, the wrapper in
    # module order; a different order must find the same arena (and keep its reducer), a different set must not
    # silently re-home parameters that have a gradient exchange attached
    assert ParamArena.of(list(reversed(ps))) is arena and arena.reducer is red and arena.covers([ps[2], ps[0], ps[3], ps[1]])
    assert not arena.covers(ps[:3]) and not arena.covers(ps + [ps[0]])
    try:
        ParamArena.of(ps[:2])
        raise AssertionError("a partial parameter list replaced an arena that has a reducer")
    except RuntimeError as e:
        assert "gradient reducer" in str(e)
    assert all(p._dcfp_slot.arena is arena for p in ps)

    def loss_fn(w, b, extra, unused, xs, ys):
        return F.cross_entropy(F.conv2d(xs, w, b), ys, ignore_index=255) + 1e-3 * (extra ** 2).sum() + 0.0 * unused.sum()

    class Direct(torch.autograd.Function):                       # stands in for the HIP Functions' direct write
        @staticmethod
        def forward(ctx, x, *params):
            ctx.params = params
            with torch.enable_grad():
                leaves = [p.detach().clone().requires_grad_(True) for p in params]
                ctx.inner = (leaves, loss_fn(*leaves, feats[shards[rank]], labels[shards[rank]]))
            return x * 0 + ctx.inner[1].detach()

        @staticmethod
        def backward(ctx, gout):
            leaves, inner = ctx.inner
            grads = torch.autograd.grad(inner, leaves[:3])       # `unused` (index 3) never gets a gradient
            outs = []
            for p, gr in zip(reversed(ctx.params[:3]), reversed(grads)):
                t, tok = grad_target(p)
                t.copy_(gr * gout)
                outs.append(grad_commit(p, t, tok))
            return (None,) + tuple(reversed(outs)) + (None,)
    arena.zero_grad()
    # a backward that died half-way (exception in a later node) must not poison the next step
    red._active, red.works, red.launched = True, [], 1           # what such a backward leaves behind
    red.begin_step()
    assert not red._active and red.launched == 0
    out = Direct.apply(torch.ones((), requires_grad=True), *ps)
    out.backward()
    assert red.launched == 3                                     # the unused range is completed with zeros
    total = sum(loss_fn(*ref, feats[sh], labels[sh]) for sh in shards) / world
    total.backward()
    for p, r in zip(ps[:3], ref[:3]):
        assert torch.allclose(p.grad, r.grad, atol=1e-6), (p.shape, (p.grad - r.grad).abs().max())
    assert ps[3].grad is None and float(arena.grad_views[3].abs().sum()) == 0.0
    # in-place zero_grad keeps the views attached and zeroes the arena with one fill
    arena.zero_grad(set_to_none=False)
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in ps)
    out = Direct.apply(torch.ones((), requires_grad=True), *ps)
    out.backward()
    for p, r in zip(ps[:3], ref[:3]):
        assert torch.allclose(p.grad, r.grad, atol=1e-6)


def test_gloo_world2():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) and ret.get(1)
