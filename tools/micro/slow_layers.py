import sys, os, argparse, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from dcfp_amd import ops, _lib, optimizer as opt, pruners
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
m = bench.build_model("resnet101", dev, None)
o = opt.build_optimizer(bench._OptArgs, m); o.zero_grad()
tp = pruners.dcfp_pruning(m, 0.999)
x, y = bench.synthetic_batch(4, 1024, 2048, 12345, dev)
def step():
    o.zero_grad()
    l = m(x, y, deepsup=True)["loss"]; l.backward(); tp.step(m); o.step()
for _ in range(2): step()
torch.cuda.synchronize()
ops.profile_start(); step(); recs = ops.profile_stop()
which = {"conv_fwd": _lib.CONV_FWD, "conv_dgrad": _lib.CONV_DGRAD, "conv_wgrad": _lib.CONV_WGRAD, "conv_dgrad_red": _lib.CONV_DGRAD}
agg = {}
for kind, key, work, ms in recs:
    if kind in which:
        name = ops.conv_kernel_name(key, which[kind])
        k = (kind, key.Cin, key.Cout, key.H, key.W, key.KH, key.stride, key.dil, name)
        a = agg.setdefault(k, [0.0, 0.0, 0]); a[0] += work; a[1] += ms; a[2] += 1
tot = 0
if "--all" in sys.argv:      # every (pass, shape) entry of the step with the MFMA work it really issues
    desc = {}
    for kind, key, work, ms in recs:
        if kind in which:
            desc[(kind, key.Cin, key.Cout, key.H, key.W, key.KH, key.stride, key.dil, ops.conv_kernel_name(key, which[kind]))] = \
                ops.conv_executed_fraction(key, which[kind])
    all_ms = 0.0
    for k, (w, ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        all_ms += ms
        print("%-14s Cin%4d Cout%4d %4dx%4d k%d s%d d%2d  n=%2d  %7.3f ms  %6.1f TF nominal  %6.1f TF executed  %s" % (
            k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], n, ms, w / (ms * 1e-3) / 1e12, w * desc[k] / (ms * 1e-3) / 1e12, k[8]))
    print("total %.1f ms" % all_ms)
    sys.exit(0)
for k, (w, ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = w / (ms * 1e-3) / 1e12
    if tf < 125 and not k[-1].startswith("winograd"):
        tot += ms
        print("%-10s Cin%4d Cout%4d %4dx%4d k%d s%d d%2d  n=%2d  %6.3f ms  %6.1f TF  %s" % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], n, ms, tf, k[8]))
print("total below 125 TF (non-Winograd): %.2f ms" % tot)
