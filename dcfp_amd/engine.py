"""Distributed runtime — the hot-path half of engine.py:17-133: one process per GPU,
torch.distributed over RCCL/xGMI (backend 'nccl' on ROCm), SyncBN conversion + gradient
all-reduce wrapper, loss all-reduce.  Data loaders (engine.py:73-114) are outside the hot
path: the benchmark feeds synthetic tensors."""
import argparse
import os

import torch
import torch.distributed as dist

from . import syncbn_p2p
from .utils.pyt_utils import all_reduce_tensor


class DataParallel(torch.nn.Module):
    """What torch's DistributedDataParallel does for the reference (engine.py:66-68), on the arenas of
    dcfp_amd/arena.py: at construction rank 0's parameters and buffers are broadcast (ONE collective for
    the parameter arena, one per buffer dtype); in backward the kernels write gradients straight into the
    gradient arena and arena.GradReducer averages it over the ranks in a few large all-reduces issued as
    soon as their range is complete; `loss.backward()` returns with averaged gradients in place.
    `.module` is the wrapped model, as with DDP."""

    def __init__(self, module, group=None, n_chunks=3):
        super().__init__()
        from .arena import ParamArena, GradReducer
        self.module = module
        params = [p for p in module.parameters() if p.requires_grad]
        self.arena = ParamArena.of(params)
        self.group = group
        with torch.no_grad():
            dist.broadcast(self.arena.flat_param, 0, group=group)
            by_dtype = {}
            for b in module.buffers():
                by_dtype.setdefault(b.dtype, []).append(b)
            for dt, bufs in by_dtype.items():
                flat = torch.cat([b.reshape(-1) for b in bufs])
                dist.broadcast(flat, 0, group=group)
                off = 0
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view(b.shape))
                    off += b.numel()
        # the broadcasts wrote through the flat tensors: neither a Parameter's version counter nor ops.WEIGHT_EPOCH
        # moved, so permuted-weight copies / folded-BN caches of a forward that ran before wrapping would be stale
        from . import ops
        ops.WEIGHT_EPOCH[0] += 1
        self.reducer = GradReducer(self.arena, group, n_chunks)
        self._params = params

    def forward(self, *args, **kwargs):
        if torch.is_grad_enabled():
            for p in self._params:          # e.g. an optimizer built over another parameter list re-homed them
                slot = getattr(p, "_dcfp_slot", None)
                if slot is None or slot.arena is not self.arena:
                    raise RuntimeError("DataParallel: a parameter left the gradient arena of this wrapper - its "
                                       "gradient would not be exchanged between the ranks")
            self.reducer.begin_step()
        return self.module(*args, **kwargs)


class Engine(object):
    def __init__(self, custom_parser=None, backend=None):
        self.distributed = False
        self.devices = None
        self.local_rank = 0
        self.world_size = 1
        self.parser = custom_parser if custom_parser is not None else argparse.ArgumentParser()
        assert isinstance(self.parser, argparse.ArgumentParser)
        self.inject_default_parser()
        self.args, _ = self.parser.parse_known_args()
        if "WORLD_SIZE" in os.environ and getattr(self.args, "ddp", True):
            self.distributed = int(os.environ["WORLD_SIZE"]) > 1
        if self.distributed:
            # torchrun exports LOCAL_RANK; torch.distributed.launch passes --local_rank
            self.local_rank = int(os.environ.get("LOCAL_RANK", self.args.local_rank))
            self.world_size = int(os.environ["WORLD_SIZE"])
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, init_method="env://")
            self.devices = list(range(self.world_size))
        else:
            self.devices = [0]

    def inject_default_parser(self):
        p = self.parser
        have = {a.dest for a in p._actions}
        if "devices" not in have:
            p.add_argument("-d", "--devices", default="", help="set data parallel training")
        if "continue_fpath" not in have:
            p.add_argument("-c", "--continue", type=str, metavar="FILE", dest="continue_fpath",
                           help="continue from one certain checkpoint")
        if "local_rank" not in have:
            p.add_argument("--local_rank", default=0, type=int, help="process rank on node")

    def data_parallel(self, model, n_chunks=3, torch_ddp=False, bucket_cap_mb=128):
        """SyncBatchNorm conversion + gradient-averaging wrapper (engine.py:63-71).  The BN layers stay
        parameter holders; their SyncBN semantics are executed by dcfp_amd.ops.  The wrapper is
        dcfp_amd.engine.DataParallel: parameters / gradients in flat arenas, the gradient exchange as
        `n_chunks` contiguous all-reduces over RCCL overlapped with backward (xGMI rings are per-link bound
        and the step is compute-bound, so few ~87 MB messages beat many small buckets).
        torch_ddp=True wraps in torch's DistributedDataParallel instead (set DCFP_ARENA_DIRECT=0 with it:
        its hooks need the gradients to pass through autograd)."""
        if self.distributed:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            if torch_ddp:
                from . import arena
                if arena.DIRECT:
                    raise RuntimeError("Engine.data_parallel(torch_ddp=True) needs DCFP_ARENA_DIRECT=0: with direct "
                                       "arena writes the gradients bypass torch DDP's hooks")
                ids = [self.local_rank] if torch.cuda.is_available() else None
                return torch.nn.parallel.DistributedDataParallel(
                    model, device_ids=ids, output_device=self.local_rank if ids else None,
                    bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
            model = DataParallel(model, n_chunks=n_chunks)
            if syncbn_p2p.wanted() and torch.cuda.is_available():
                # opt-in (DCFP_SYNCBN_P2P=1): the 230 per-layer SyncBN collectives become single peer-to-peer kernels
                if dist.get_rank() == 0:
                    import sys
                    print("dcfp_amd: DCFP_SYNCBN_P2P=1 - SyncBN statistics through the peer-to-peer exchange kernel.  It is "
                          "bit-identical to the collectives between processes SHARING one GPU (tests/test_syncbn_p2p_gpu.py, "
                          "tests/test_ddp2_gpu.py) but has never run across xGMI: its flag / data ordering over the fabric is "
                          "argued, not observed.  A give-up poisons the statistics with NaN and stops the job at that step "
                          "(syncbn_p2p.check_all()); compare `comm.syncbn_exposed_ms` with the default RCCL path before relying "
                          "on it.", file=sys.stderr, flush=True)
                syncbn_p2p.enable(dist.group.WORLD, torch.device("cuda", torch.cuda.current_device()))
        return model

    def all_reduce_tensor(self, tensor, norm=True):
        if self.distributed:
            return all_reduce_tensor(tensor, world_size=self.world_size, norm=norm)
        return torch.mean(tensor)

    def __enter__(self):
        return self

    def __exit__(self, type, value, tb):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return False
