"""The data-parallel code path on ONE MI355X: a world-size-1 RCCL group with the SyncBN exchange forced on
(DCFP_FORCE_SYNCBN=1) and Engine.data_parallel's wrapper (gradient arena + chunked all-reduce) must give
the SAME BITS as the plain single-process path - loss, every parameter gradient, the EIC vector, the BN
running statistics - on DeepLabv3-R50 2x3x129x257 and on BASELINE config 3 itself, DeepLabv3-R101 4x3x1024x2048
(engine.py:63-68, train.py:259-268).  At world size 1
every collective is the identity, so any difference is a bug in the exchange plumbing (pooled-statistics
kernel, device-side count, asynchronous sums, arena views), which is exactly what N > 1 runs on top of.
Runs in a child process: it creates a process group and flips a process-wide switch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child():
    sys.path.insert(0, ROOT)
    import argparse
    import torch
    import torch.distributed as dist
    from dcfp_amd import networks, pruners, optimizer as opt
    from dcfp_amd.engine import Engine, DataParallel
    from dcfp_amd.loss.criterion import build_criterions
    from oracle import fill

    class DS:
        ignore_label = 255; num_classes = 19; class_weights = None

    class A:
        # two param groups (optimizer.py:18-33): the optimizer walks the parameters in decay / no-decay order, the
        # data-parallel wrapper in module order - they must still share ONE arena, or the exchange is lost
        no_decay = "bn"; optim = "sgd"; momentum = 0.9; learning_rate = 1e-3; weight_decay = 5e-4
    dev = torch.device("cuda:0")
    bb = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}
    full = os.environ.get("DCFP_DDP_CASE") == "config3"
    if full:     # BASELINE config 3: DeepLabv3-R101, 4x3x1024x2048 - the 256 x 256 LDS-DMA tiles, the fused Winograd kernels,
        # split-K weight gradients writing into the gradient arena, 2048-channel SyncBN rows (bench.py's inputs)
        backbone = "resnet101"
        g = torch.Generator().manual_seed(12345)
        x = torch.randn(4, 3, 1024, 2048, generator=g).to(dev)
        lab = torch.randint(0, 19, (4, 1024, 2048), generator=g)
        lab[torch.rand(lab.shape, generator=g) < 0.05] = 255
        lab = lab.to(dev)
    else:
        backbone = "resnet50"
        x = fill.closed_form_input(2, 129, 257).to(dev)
        lab = fill.closed_form_labels(2, 129, 257).to(dev)

    def run(ddp):
        torch.manual_seed(12345)
        m = networks.deeplabv3.Seg_Model(backbone=backbone, backbone_para=dict(bb), num_classes=19, align_corner=True,
                                         criterion=build_criterions("ce", DS(), {"ds_weight": 0.4}), deepsup=True)
        if not full:
            m.load_state_dict(fill.closed_form_state(m.state_dict()))
        m.conv_deepsup[3].p = 0.0
        m = m.to(dev).train()
        optimizer = opt.build_optimizer(A, m)
        optimizer.zero_grad()
        tp = pruners.dcfp_pruning(m, 0.999)
        info = {}
        if ddp:
            sys.argv = ["x"]
            eng = Engine(custom_parser=argparse.ArgumentParser())
            eng.distributed = True
            model = eng.data_parallel(m)
            assert isinstance(model, DataParallel)
            assert any(isinstance(mod, torch.nn.SyncBatchNorm) for mod in m.modules())
        else:
            model = m
        losses = []
        for it in range(2):                       # second step: after an optimizer update, cached Wp / tables reused
            optimizer.zero_grad()
            loss = model(x, lab, deepsup=True)["loss"]
            red = eng.all_reduce_tensor(loss) if ddp else loss
            losses.append(red.item())
            loss.backward()
            tp.step(m)
            if it == 0:
                grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
                eic = torch.cat([tp.get_eic()["eic"][n].reshape(-1) for n in tp._names]).clone()
            optimizer.step()
        torch.cuda.synchronize()
        if ddp:
            info["launched"] = model.reducer.launched
            info["chunks"] = len(model.reducer.bounds)
            info["one_arena"] = optimizer.arena() is model.arena and model.arena.reducer is model.reducer
        info["groups"] = [len(g["params"]) for g in optimizer.param_groups]
        bufs = {k: v.detach().clone() for k, v in m.state_dict().items()}
        info["table_rebuilds"] = optimizer.table_rebuilds
        info["peak_GiB"] = torch.cuda.max_memory_allocated() / 2**30
        del model, m, optimizer, tp, loss, red
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return losses, grads, eic, bufs, info

    plain = run(False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", DCFP_FORCE_SYNCBN="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    ddp = run(True)
    dist.destroy_process_group()
    out = {"loss": [plain[0], ddp[0]], "loss_equal": plain[0] == ddp[0],
           "grad_diff": [k for k in plain[1] if not torch.equal(plain[1][k], ddp[1][k])],
           "eic_equal": bool(torch.equal(plain[2], ddp[2])),
           "state_diff": [k for k in plain[3] if not torch.equal(plain[3][k], ddp[3][k])],
           "n_params": len(plain[1]), "info": [plain[4], ddp[4]],
           "nbt": int(ddp[3]["backbone.bn1.num_batches_tracked"])}
    print("DDP_RESULT " + json.dumps(out))


@pytest.mark.parametrize("case", ["r50_129x257", "config3"])
def test_syncbn_ddp_path_bit_identical_to_plain(cuda, case):
    # DCFP_FANIN_BN_SUMS=2: the plain run too takes bn3's sums from the fan-in epilogue, as the SyncBN path does (by
    # default a run without SyncBN leaves them to the fused BatchNorm backward: another summation order for bn3).  Every
    # other BatchNorm backward is the fused kernel in the plain run and reduce + exchange + apply in the SyncBN run -
    # the bit-identity below holds the two against each other at model level.
    env = dict(os.environ, DCFP_DDP_CASE=case, DCFP_FANIN_BN_SUMS="2")
    env.pop("DCFP_FORCE_SYNCBN", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("DDP_RESULT ")][-1][len("DDP_RESULT "):])
    assert rec["loss_equal"], rec["loss"]
    assert rec["grad_diff"] == [], rec["grad_diff"][:8]
    assert rec["eic_equal"]
    assert rec["state_diff"] == [], rec["state_diff"][:8]          # weights after 2 SGD steps, running stats, counters
    assert rec["n_params"] > 150 and rec["nbt"] == 2
    assert rec["info"][1]["launched"] == rec["info"][1]["chunks"] == 3      # the exchange really ran, in 3 all-reduces
    assert rec["info"][1]["one_arena"] and all(n > 0 for n in rec["info"][1]["groups"]) and len(rec["info"][1]["groups"]) == 2
    assert rec["info"][0]["table_rebuilds"] == 2 and rec["info"][1]["table_rebuilds"] == 2   # one pointer table per group, built once


if __name__ == "__main__" and "--child" in sys.argv:
    _child()
