#!/usr/bin/env python
"""Per-SHAPE conv timing inside one instrumented training step (bench.py aggregates by kernel
instance only): which layers of a (pruned) model sit furthest from the MFMA roofline."""
import argparse
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channel-cfg", default=None)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    from dcfp_amd import optimizer as opt, pruners, ops, _lib
    device = torch.device("cuda:0")
    torch.manual_seed(12345)
    model = bench.build_model("resnet101", device, args.channel_cfg)
    optimizer = opt.build_optimizer(bench._OptArgs, model)
    pruning = pruners.dcfp_pruning(model, 0.999)
    images, labels = bench.synthetic_batch(4, 1024, 2048, 12345, device)

    def step(it):
        optimizer.zero_grad()
        loss = model(images, labels, deepsup=True)
        loss["loss"].item()
        loss["loss"].backward()
        pruning.step(model)
        optimizer.step()

    step(0); step(1)
    ops.profile_start()
    step(2)
    recs = ops.profile_stop()
    which = {"conv_fwd": _lib.CONV_FWD, "conv_dgrad": _lib.CONV_DGRAD, "conv_wgrad": _lib.CONV_WGRAD, "conv_dgrad_red": _lib.CONV_DGRAD}
    agg = defaultdict(lambda: [0.0, 0.0, 0, ""])
    for kind, key, work, ms in recs:
        if kind not in which:
            continue
        d = key
        k = (kind, d.Cin, d.Cout, d.KH, d.H, d.W, d.stride, d.dil)
        a = agg[k]
        a[0] += work; a[1] += ms; a[2] += 1; a[3] = ops.conv_kernel_name(d, which[kind])
    tot = sum(a[1] for a in agg.values())
    print(f"conv total {tot:.1f} ms")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        kind, cin, cout, kh, h, w, s, dil = k
        print(f"{kind:10s} {cin:5d}->{cout:5d} {kh}x{kh} s{s} d{dil:<2d} @{h}x{w}  x{a[2]:<3d} {a[1]:7.2f} ms "
              f"{a[0] / a[1] / 1e9:6.1f} TF  {a[3]}")


if __name__ == "__main__":
    main()
