"""The peer-to-peer SyncBatchNorm exchange (dcfp_amd/csrc/syncbn_p2p.hip, dcfp_amd/syncbn_p2p.py) with TWO (and FOUR) real
processes on one MI355X: each rank's mailbox is mapped into the other through hipIpc, exactly as across GPUs
(engine.py:65 of the reference: nn.SyncBatchNorm's per-layer all_gather / all_reduce).  Checked against what the
gloo collectives deliver for the same rows: gather bit-exact, rank-order sum bit-exact, the pooled statistics
bit-identical to dcfp_syncbn_combine_f32 and to ops.syncbn_combine_reference, running statistics included; 400
back-to-back exchanges with the ranks deliberately skewed (slot reuse); the side-stream form; and the exit
condition: a rank whose peer never shows up gets NaN outputs and a status word, not a hang."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT = "29561"


def _child(rank, world, out_path):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    import torch.distributed as dist
    from dcfp_amd import _lib, ops, syncbn_p2p
    from dcfp_amd._lib import BnRunning

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(PORT) + world), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    g = dist.group.WORLD
    if os.environ.get("P2P_TEST_SETUP_FAILURE"):      # ONE rank cannot allocate: every rank must raise, none may hang
        if rank == 1:
            os.environ["DCFP_P2P_MEM"] = "9"
        try:
            syncbn_p2p.enable(g, dev)
            raised = None
        except RuntimeError as e:
            raised = str(e)
        dist.barrier()
        dist.destroy_process_group()
        with open(out_path + ".%d" % rank, "w") as f:
            json.dump({"raised": raised}, f)
        return
    px = syncbn_p2p.enable(g, dev, max_channels=2048)
    rec = {"world": px.world, "cap": px.cap}

    def gathered(local):            # the rows as a host-side collective delivers them
        rows = [torch.empty_like(local.cpu()) for _ in range(world)]
        dist.all_gather(rows, local.cpu())
        return torch.stack(rows).to(dev)

    gen = torch.Generator().manual_seed(100 + rank)
    ok = True
    for Cc in (64, 65, 257, 2048):
        n = 2 * Cc + 1
        local = torch.randn(n, generator=gen).to(dev)
        local[Cc:2 * Cc].abs_()                                   # variances
        local[2 * Cc] = float(1000 + 37 * rank)                   # unequal pixel counts
        rows = gathered(local)
        out0 = torch.empty(world * n, device=dev)
        px.exchange(local, out0, 0)
        chk = {"gather": bool(torch.equal(out0.view(world, n), rows))}
        # backward form: 2C floats, rank-order sum
        both = local[:2 * Cc].contiguous()
        out1 = torch.empty(2 * Cc, device=dev)
        px.exchange(both, out1, 1)
        want = rows[0, :2 * Cc].clone()
        for r in range(1, world):
            want = want + rows[r, :2 * Cc]
        chk["sum"] = bool(torch.equal(out1, want))
        # forward form: pooled statistics + running statistics
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        run = BnRunning(rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 0)
        out2 = torch.empty(n, device=dev)
        px.exchange(local, out2, 2, C.byref(run))
        gm, gv, tot = ops.syncbn_combine_reference(rows, Cc)
        chk["pooled_vs_host"] = bool(torch.equal(out2[:Cc], gm) and torch.equal(out2[Cc:2 * Cc], gv) and torch.equal(out2[2 * Cc:], tot))
        rm2, rv2 = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        nbt2 = torch.zeros(1, dtype=torch.int64, device=dev)
        run2 = BnRunning(rm2.data_ptr(), rv2.data_ptr(), nbt2.data_ptr(), 0.1, 0)
        ref = torch.empty(n, device=dev)
        _lib.check(_lib.lib().dcfp_syncbn_combine_f32(C.c_void_p(rows.data_ptr()), world, Cc, C.c_void_p(ref.data_ptr()),
                                                      C.c_void_p(ref[Cc:].data_ptr()), C.c_void_p(ref[2 * Cc:].data_ptr()),
                                                      C.byref(run2), None), "combine")
        torch.cuda.synchronize()
        chk["pooled_vs_kernel"] = bool(torch.equal(out2, ref))
        chk["running"] = bool(torch.equal(rm, rm2) and torch.equal(rv, rv2) and int(nbt) == 1 and int(nbt2) == 1)
        if not chk["running"]:
            chk["running_detail"] = [float((rm - rm2).abs().max()), float((rv - rv2).abs().max()), int(nbt), int(nbt2),
                                     float(rm.abs().max()), float(rm2.abs().max())]
        ok &= all(v is True for k, v in chk.items() if k != "running_detail")
        rec["C%d" % Cc] = chk
    rec["basic_ok"] = bool(ok)

    # 400 exchanges back to back, the ranks skewed against each other (host sleeps on alternating ranks): every slot
    # is reused 100 times; the sums must all be right
    Cc = 512
    outs, wants = [], []
    base = torch.arange(2 * Cc, device=dev, dtype=torch.float32)
    for k in range(400):
        if k % 50 == 7 + 11 * rank:
            time.sleep(0.05)
        local = base * float(rank + 1) + float(k)
        out = torch.empty(2 * Cc, device=dev)
        if k % 3 == 0:
            px.exchange_async(local, out, 1).wait()
        else:
            px.exchange(local, out, 1)
        outs.append(out)
        w = base * 1.0 + float(k)
        for r in range(1, world):
            w = w + (base * float(r + 1) + float(k))
        wants.append(w)
    torch.cuda.synchronize()
    rec["stress_ok"] = all(bool(torch.equal(a, b)) for a, b in zip(outs, wants))
    px.check()                                                     # no exchange timed out so far
    rec["exchanges"] = px.exchanges
    dist.barrier()

    # exit condition: rank 0 starts an exchange its peer never joins
    if rank == 0:
        px.spin = 20000
        lone = torch.ones(8, device=dev)
        out = torch.zeros(8, device=dev)
        t0 = time.time()
        px.exchange(lone, out, 1)
        torch.cuda.synchronize()
        rec["timeout_s"] = time.time() - t0
        rec["timeout_nan"] = bool(torch.isnan(out).all())
        try:
            px.check()
            rec["timeout_raised"] = False
        except RuntimeError:
            rec["timeout_raised"] = True
    dist.barrier()
    syncbn_p2p.disable()
    dist.destroy_process_group()
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(rec, f)


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_processes_sharing_one_gpu(cuda, tmp_path, world):
    """world = 4: four mailboxes mapped into each other, four-term rank-order sums (the box allows 6 GPU processes)."""
    out = str(tmp_path / "p2p.json")
    env = dict(os.environ)
    env.pop("DCFP_SYNCBN_P2P", None)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(world), out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, logs[r][-3000:])
    recs = [json.load(open(out + ".%d" % r)) for r in range(world)]
    print("P2P", json.dumps(recs))
    for rec in recs:
        assert rec["world"] == world and rec["basic_ok"] and rec["stress_ok"], rec
        assert rec["exchanges"] == 4 * 3 + 400
    assert recs[0]["timeout_nan"] and recs[0]["timeout_raised"] and recs[0]["timeout_s"] < 30.0, recs[0]


def test_p2p_setup_failure_on_one_rank_raises_on_all(cuda, tmp_path):
    out = str(tmp_path / "p2pf.json")
    env = dict(os.environ, P2P_TEST_SETUP_FAILURE="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), "2", out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=200)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, logs[r][-3000:])
    recs = [json.load(open(out + ".%d" % r)) for r in range(2)]
    for rec in recs:
        assert rec["raised"] and "rank 1" in rec["raised"] and "p2p_alloc" in rec["raised"], recs


if __name__ == "__main__" and "--child" in sys.argv:
    i = sys.argv.index("--child")
    _child(int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[i + 3])
