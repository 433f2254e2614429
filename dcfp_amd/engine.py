"""Distributed runtime — the hot-path half of engine.py:17-133: one process per GPU,
torch.distributed over RCCL/xGMI (backend 'nccl' on ROCm), SyncBN conversion + gradient
all-reduce wrapper, loss all-reduce.  Data loaders (engine.py:73-114) are outside the hot
path: the benchmark feeds synthetic tensors."""
import argparse
import os

import torch
import torch.distributed as dist

from .utils.pyt_utils import all_reduce_tensor


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

    def data_parallel(self, model, bucket_cap_mb=128):
        """SyncBatchNorm conversion + DDP (engine.py:63-71).  The BN layers stay parameter
        holders; their SyncBN semantics are executed by dcfp_amd.ops.BatchNormActFn.  Buckets
        are large (default 128 MB): xGMI rings are per-link bound and the step is compute-bound,
        so few big all-reduces overlapped with backward beat many small ones."""
        if self.distributed:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            ids = [self.local_rank] if torch.cuda.is_available() else None
            model = torch.nn.parallel.DistributedDataParallel(
                model, device_ids=ids, output_device=self.local_rank if ids else None,
                bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
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
