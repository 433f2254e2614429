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
    # max-over-ranks timing reduction used by bench.py
    tt = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.item() == world - 1
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_gloo_world2():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) and ret.get(1)
