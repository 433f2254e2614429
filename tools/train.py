#!/usr/bin/env python
"""Training + scoring driver — the loop of the reference's train.py:140-293 (same flags for
the model / optimiser / loss / pruning options) over SYNTHETIC Cityscapes-shaped batches (the
dataset loaders of the reference are outside the hot path: SURVEY.md §2 row 19).
One process per GPU: `python -m torch.distributed.run --nproc-per-node N tools/train.py ...`."""
import argparse
import json
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

from dcfp_amd import networks, pruners  # noqa: E402
from dcfp_amd.engine import Engine  # noqa: E402
from dcfp_amd.loss.criterion import build_criterions  # noqa: E402
from dcfp_amd.optimizer import adjust_learning_rate, build_optimizer  # noqa: E402
from dcfp_amd.utils.pyt_utils import load_model  # noqa: E402


def str2bool(v):
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_parser():
    p = argparse.ArgumentParser(description="DCFP")
    p.add_argument("--start-iters", type=int, default=0)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--num-steps", type=int, default=40000)
    p.add_argument("--random-seed", type=int, default=12345)
    p.add_argument("--ddp", type=str2bool, default="True")
    p.add_argument("--save-pred-every", type=int, default=10000)
    p.add_argument("--save-steps", type=int, default=0)
    p.add_argument("--snapshot-dir", type=str, default="ckpt")
    p.add_argument("--batch-size", type=int, default=8, help="global batch (split over ranks)")
    p.add_argument("--ignore-label", type=int, default=255)
    p.add_argument("--input-size", type=str, default="769,769")
    p.add_argument("--num-classes", type=int, default=19)
    p.add_argument("--model", type=str, default="deeplabv3")
    p.add_argument("--backbone", type=str, default="resnet50")
    p.add_argument("--backbone-para", type=str, default='{"pretrained": false}')
    p.add_argument("--model-para", type=str, default="{}")
    p.add_argument("--align-corner", type=str2bool, default="True")
    p.add_argument("--no-decay", type=str, default=None)
    p.add_argument("--optim", type=str, default="sgd")
    p.add_argument("--learning-rate", type=float, default=1e-2)
    p.add_argument("--weight-decay", type=float, default=0.0005)
    p.add_argument("--power", type=float, default=0.9)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--betas", type=str, default="0.9,0.999")
    p.add_argument("--warmup", type=int, default=-1)
    p.add_argument("--deepsup", type=str2bool, default="True")
    p.add_argument("--loss-type", type=str, default="ce")
    p.add_argument("--loss-para", type=str, default="{}")
    p.add_argument("--prune-type", type=str, default=None)
    p.add_argument("--channel-cfg", type=str, default=None)
    p.add_argument("--log-time", type=str2bool, default="False",
                   help="synchronise after every iteration and print its device time (batch synthesis excluded)")
    return p


class SyntheticDataset:
    """Cityscapes-shaped tensors: images N(0,1), labels uniform in [0,num_classes) with 5 % ignore."""

    def __init__(self, num_classes, ignore_label, size, seed):
        self.num_classes, self.ignore_label, self.class_weights = num_classes, ignore_label, None
        self.size, self.gen = size, torch.Generator().manual_seed(seed)

    def batch(self, n, device):
        h, w = self.size
        images = torch.randn(n, 3, h, w, generator=self.gen)
        labels = torch.randint(0, self.num_classes, (n, h, w), generator=self.gen)
        labels[torch.rand(n, h, w, generator=self.gen) < 0.05] = self.ignore_label
        return images.to(device, non_blocking=True), labels.to(device, non_blocking=True)


def main(argv=None):
    parser = get_parser()
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    with Engine(custom_parser=parser) as engine:
        args = parser.parse_args()
        main_flag = (not engine.distributed) or engine.local_rank == 0
        if main_flag:
            os.makedirs(args.snapshot_dir, exist_ok=True)
        args.save_steps = min(args.save_steps or args.num_steps, args.num_steps)
        seed = args.random_seed + (engine.local_rank if engine.distributed else 0)
        torch.manual_seed(seed)
        h, w = map(int, args.input_size.split(","))
        dataset = SyntheticDataset(args.num_classes, args.ignore_label, (h, w), seed)
        criterion = build_criterions(args.loss_type, dataset, json.loads(args.loss_para))
        seg_model = getattr(networks, args.model).Seg_Model(
            backbone=args.backbone, backbone_para=json.loads(args.backbone_para),
            model_para=json.loads(args.model_para), num_classes=dataset.num_classes,
            align_corner=args.align_corner, criterion=criterion, deepsup=args.deepsup)
        if args.channel_cfg is not None:
            channel_cfg = torch.load(args.channel_cfg, weights_only=False)
            pruners.init_pruned_model(seg_model, channel_cfg)
            if main_flag:
                torch.save(channel_cfg, osp.join(args.snapshot_dir, "channel_cfg.pth"))
        if args.resume:
            load_model(seg_model, args.resume)
        device = torch.device("cuda", engine.local_rank)
        seg_model.to(device)
        optimizer = build_optimizer(args, seg_model)
        optimizer.zero_grad()
        train_pruning = pruners.dcfp_pruning(seg_model, 0.999) if args.prune_type == "dcfp" else None
        model = engine.data_parallel(seg_model)
        model.train()
        per_rank = max(1, args.batch_size // engine.world_size)
        for it in range(args.start_iters, args.num_steps):
            images, labels = dataset.batch(per_rank, device)
            if args.log_time:
                torch.cuda.synchronize()
                t_it = time.perf_counter()
            if "gsrl" in args.loss_type:   # fine-tune stage: {'ori', 'weight'} labels (datasets/Base.py:73-89)
                labels = {"ori": labels, "weight": 1.0 + (labels % 3 == 0).float()}
            optimizer.zero_grad()
            lr = adjust_learning_rate(optimizer, args.learning_rate, it, args.num_steps, args.power, args.warmup)
            loss = model(images, labels, deepsup=args.deepsup)
            if not bool(loss["loss"] == loss["loss"]):     # (train.py:260-261 asserts this - a host sync per step, as there)
                from dcfp_amd import syncbn_p2p
                syncbn_p2p.check_all()                     # a SyncBN exchange that gave up on a peer poisons with NaN: say so
                from dcfp_amd import ops as _ops
                _ops.check_fused_status()                  # ... and so does a fused BatchNorm backward that gave up
                raise AssertionError("loss is NaN")
            reduce_loss = engine.all_reduce_tensor(loss["loss"])
            loss["loss"].backward()
            if train_pruning is not None:
                train_pruning.step(seg_model)
            optimizer.step()
            if args.log_time:
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t_it) * 1e3
                if main_flag:
                    print("  step %.1f ms  %.2f images/s/GPU" % (ms, per_rank / ms * 1e3), flush=True)
            if main_flag:
                print("Iters%d/%d lr=%.2e loss=%.4f" % (it + 1, args.num_steps, lr, reduce_loss.item()), flush=True)
                done = it + 1
                if done >= args.save_steps and ((args.num_steps - done) % args.save_pred_every == 0 or done >= args.num_steps):
                    torch.save(seg_model.state_dict(), osp.join(args.snapshot_dir, "CS_scenes_%d.pth" % done))
        if main_flag and train_pruning is not None:
            train_pruning.export_eic(osp.join(args.snapshot_dir, "score.pth"))
        from dcfp_amd import syncbn_p2p, ops as _ops
        _ops.check_fused_status()
        syncbn_p2p.finish()            # DCFP_SYNCBN_P2P=1: no exchange may have given up on a peer


if __name__ == "__main__":
    main()
