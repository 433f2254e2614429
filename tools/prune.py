#!/usr/bin/env python
"""Offline pruning driver — same CLI and control flow as the reference's prune.py:24-124:
raise global_percent from --start_global_percent by --step_global_percent until
FLOPs(pruned)/FLOPs(full) <= 1 - prune_ratio, writing pruned.pth + channel_cfg.pth.
CPU, single process, like the reference; no forward pass is needed (static graph + static
FLOPs counter)."""
import argparse
import copy
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dcfp_amd import networks  # noqa: E402
from dcfp_amd.pruners import init_pruned_model  # noqa: E402
from dcfp_amd.pruners.dcfp_pruner import DCFPPruner  # noqa: E402
from dcfp_amd.utils.flops_counter import get_model_complexity_info  # noqa: E402
from dcfp_amd.utils.pyt_utils import load_model  # noqa: E402


def str2bool(v):
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_parser():
    p = argparse.ArgumentParser(description="DCFP")
    p.add_argument("--save-path", type=str, default="./ckpt")
    p.add_argument("--model-path", type=str, default="")
    p.add_argument("--score-path", type=str, default="")
    p.add_argument("--prune-ratio", type=float, default=0.6)
    p.add_argument("--start_global_percent", type=float, default=0.5)
    p.add_argument("--step_global_percent", type=float, default=0.02)
    p.add_argument("--model", type=str, default="deeplabv3")
    p.add_argument("--backbone", type=str, default="resnet50")
    p.add_argument("--backbone-para", type=str, default="{}")
    p.add_argument("--model-para", type=str, default="{}")
    p.add_argument("--align-corner", type=str2bool, default="True")
    p.add_argument("--dataset", type=str, default="CS")
    return p


def get_num_classes(dataset):
    for prefix, n in (("CS", 19), ("CTX", 59), ("ADE", 150), ("COCO", 171)):
        if dataset.startswith(prefix):
            return n
    raise ValueError(dataset)


def build(args, deepsup):
    return getattr(networks, args.model).Seg_Model(
        backbone=args.backbone, backbone_para=json.loads(args.backbone_para),
        model_para=json.loads(args.model_para), num_classes=get_num_classes(args.dataset),
        align_corner=args.align_corner, criterion=None, deepsup=deepsup)


def main(argv=None):
    args = get_parser().parse_args(argv)
    os.makedirs(args.save_path, exist_ok=True)
    flops, params = get_model_complexity_info(build(args, False), (3, 512, 512), print_per_layer_stat=False)
    flops = float(flops.split(" GFLOPs")[0])
    seg_model = build(args, True)
    load_model(seg_model, args.model_path)
    global_percent = args.start_global_percent
    while True:
        pruner = DCFPPruner(global_percent=global_percent, layer_keep=0.02, score_file=args.score_path)
        sub_model, channel_cfg = pruner.prune_model(copy.deepcopy(seg_model), except_start_keys=["conv_deepsup"])
        torch.save(sub_model.state_dict(), os.path.join(args.save_path, "pruned.pth"))
        torch.save(channel_cfg, os.path.join(args.save_path, "channel_cfg.pth"))
        seg_model2 = build(args, False)
        init_pruned_model(seg_model2, torch.load(os.path.join(args.save_path, "channel_cfg.pth"), weights_only=False))
        load_model(seg_model2, os.path.join(args.save_path, "pruned.pth"))
        flops2, params2 = get_model_complexity_info(seg_model2, (3, 512, 512), print_per_layer_stat=False)
        flops2 = float(flops2.split(" GFLOPs")[0])
        print("global_percent: {}, flops_ratio: {}".format(global_percent, flops2 / flops))
        if flops2 / flops <= (1 - args.prune_ratio):
            print("Finish!")
            print("flops: {}, params: {}".format(flops, params))
            print("flops2: {}, params2: {}".format(flops2, params2))
            break
        global_percent = global_percent + args.step_global_percent
        if global_percent >= 1.0:
            break
    return global_percent


if __name__ == "__main__":
    main()
