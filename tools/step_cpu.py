#!/usr/bin/env python
"""Host-side cost of one training step: how long Python takes to ENQUEUE forward and backward
against how long the GPU takes to run them (the step is GPU-bound only while enqueue < run)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from dcfp_amd import optimizer as opt, pruners
    device = torch.device("cuda:0")
    torch.manual_seed(12345)
    seg = bench.build_model("resnet101", device)
    model = seg
    if "--ddp" in sys.argv:      # SyncBN + DDP over RCCL at world size 1, as bench.py --force-ddp
        import argparse
        import torch.distributed as dist
        from dcfp_amd.engine import Engine
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ["DCFP_FORCE_SYNCBN"] = "1"
        dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
        engine = Engine(custom_parser=argparse.ArgumentParser())
        engine.distributed = True
        model = engine.data_parallel(seg)
    optimizer = opt.build_optimizer(bench._OptArgs, seg)
    pruning = pruners.dcfp_pruning(seg, 0.999)
    images, labels = bench.synthetic_batch(4, 1024, 2048, 12345, device)
    rows = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        optimizer.zero_grad()
        opt.adjust_learning_rate(optimizer, 0.01, it, 4000, 0.9, -1)
        loss = model(images, labels, deepsup=True)
        t1 = time.perf_counter()
        loss["loss"].item()
        t2 = time.perf_counter()
        loss["loss"].backward()
        pruning.step(seg)
        optimizer.step()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    for r in rows[2:]:
        print("fwd enqueue %.1f ms | wait for loss %.1f ms | bwd+eic+sgd enqueue %.1f ms | drain %.1f ms | step %.1f ms"
              % tuple(1e3 * v for v in r))


if __name__ == "__main__":
    main()
