#!/usr/bin/env python
"""Which ATen (non-HIP-library) kernels still run inside one training step, by input shape.
Everything heavy is meant to go through libdcfp_hip.so; this lists what is left."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1024,2048")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    from dcfp_amd import optimizer as opt, pruners
    device = torch.device("cuda:0")
    H, W = [int(v) for v in args.size.split(",")]
    torch.manual_seed(12345)
    model = bench.build_model("resnet101", device)
    optimizer = opt.build_optimizer(bench._OptArgs, model)
    pruning = pruners.dcfp_pruning(model, 0.999)
    images, labels = bench.synthetic_batch(args.batch, H, W, 12345, device)

    def step(it):
        optimizer.zero_grad()
        opt.adjust_learning_rate(optimizer, 0.01, it, 4000, 0.9, -1)
        loss = model(images, labels, deepsup=True)
        loss["loss"].item()
        loss["loss"].backward()
        pruning.step(model)
        optimizer.step()

    step(0); step(1)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(2)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(
        sort_by="self_cuda_time_total", row_limit=args.top, max_name_column_width=40,
        max_shapes_column_width=70))


if __name__ == "__main__":
    main()
