#!/bin/bash
# BASELINE.json config 5 end to end on one GPU, synthetic data: a few scoring iterations of
# DeepLabv3-R101 at 1024x2048 -> score.pth -> DCFPPruner masks (prune.py, on the host CPU as in the
# reference) -> slim model -> GSRL fine-tune iterations with per-step timing.
#   tools/pipeline_cfg5.sh OUTDIR [STEPS]
set -e
out=${1:-gpurun_out/pipe}; steps=${2:-6}
bp='{"pretrained": false, "os": 8, "mg_unit": [1, 2, 4], "inplanes": 128}'
common=(--model deeplabv3 --backbone resnet101 --backbone-para "$bp" --input-size 1024,2048 --batch-size 4 --ddp false --log-time true)
python tools/train.py "${common[@]}" --num-steps $steps --prune-type dcfp --snapshot-dir $out/train
python tools/prune.py --model deeplabv3 --backbone resnet101 --backbone-para "$bp" --save-path $out/prune \
    --model-path $out/train/CS_scenes_$steps.pth --score-path $out/train/score.pth --prune-ratio 0.6
python tools/train.py "${common[@]}" --num-steps $steps --loss-type gsrl --channel-cfg $out/prune/channel_cfg.pth \
    --resume $out/prune/pruned.pth --snapshot-dir $out/finetune --learning-rate 1e-3
python bench.py --channel-cfg $out/prune/channel_cfg.pth --no-cpu-baseline --steps 3 --warmup 1 > $out/bench_pruned.json
rm -f $out/train/*.pth $out/prune/pruned.pth $out/finetune/CS_scenes_*.pth   # keep the scratch dir small
