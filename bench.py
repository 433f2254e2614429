#!/usr/bin/env python
"""bench.py — training images/s of DeepLabv3-R101 (+ASPP, os=8) at 4x3x1024x2048 per GPU on
N MI355X (BASELINE.json metric; config (3) at N=1, config (4) at N>1).

A step is the reference's iteration (train.py:255-270): zero_grad -> poly LR -> forward (CE +
0.4*deep-supervision CE, fused with the 8x bilinear upsample) -> loss all-reduce/.item() ->
backward (DDP bucketed gradient all-reduce over RCCL, SyncBN statistics exchange) ->
dcfp_pruning.step (EIC) -> SGD step.  Synthetic data (images N(0,1), labels uniform 0..18 with
5 % ignore=255, seed 12345+rank), default-init weights.  One process per GPU; launched for N>1
as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.

Prints ONE JSON line with the throughput, a `roofline` object for the dominant conv kernel
(algorithmic FLOPs / live HIP-event time of its launches in one instrumented step) and, at
N=1, a `cpu_baseline` object (the CPU oracle timed on this host's cores on a bounded sample).
"""
import argparse
import subprocess
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, no TF32 on gfx950
PEAK_BF16_MFMA = 2.5e15    # same guide: dense bf16 MFMA
PEAK_HBM = 8.0e12


class _DS:
    ignore_label = 255
    num_classes = 19
    class_weights = None


class _OptArgs:
    no_decay = None
    optim = "sgd"
    momentum = 0.9
    learning_rate = 0.01
    weight_decay = 5e-4


def synthetic_batch(n, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n, 3, h, w, generator=g)
    labels = torch.randint(0, 19, (n, h, w), generator=g)
    labels[torch.rand(n, h, w, generator=g) < 0.05] = 255
    return images.to(device), labels.to(device)


def build_model(backbone, device, channel_cfg=None):
    from dcfp_amd import networks
    from dcfp_amd.loss.criterion import build_criterions
    crit = build_criterions("ce", _DS(), {"ds_weight": 0.4})
    bb = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}
    model = networks.deeplabv3.Seg_Model(backbone=backbone, backbone_para=bb, model_para={}, num_classes=19,
                                         align_corner=True, criterion=crit, deepsup=True)
    if channel_cfg:   # slim model of a prune.py run (pruners/channel_pruner.py:29-74)
        from dcfp_amd import pruners
        pruners.init_pruned_model(model, torch.load(channel_cfg, weights_only=False))
    return model.to(device).train()


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota
    (a 1-GPU box exposes all host cores but grants a 16-CPU share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cap = int(os.environ.get("DCFP_CPU_THREADS", "16"))
    return max(1, min(n, cap))


def host_ram_gib():
    try:
        return os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        return None


def cpu_baseline(backbone, n, h, w, full_hw, reps=3):
    """The CPU oracle (a port of the reference's CPU path) on a bounded sample: the same model
    at batch `n` (ASPP image-pool BN needs >= 2) and h x w pixels; images/s are scaled by the
    pixel ratio to the full 1024x2048 workload (`extrapolated` says whether that ratio is not 1)."""
    from dcfp_amd import networks
    from oracle import model as omodel
    from oracle.train_step import CpuTrainer
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(12345)
    bb = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}
    m = networks.deeplabv3.Seg_Model(backbone=backbone, backbone_para=bb, num_classes=19, align_corner=True,
                                     deepsup=True)
    cfg = omodel.Cfg("deeplabv3", backbone, align_corner=True)
    tr = CpuTrainer(m.state_dict(), cfg, lr=0.01)
    del m
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(n, 3, h, w, generator=g)
    lab = torch.randint(0, 19, (n, h, w), generator=g)
    lab[torch.rand(n, h, w, generator=g) < 0.05] = 255
    mask = torch.ones(n, 512)
    times = []
    for i in range(reps + 1):               # first step untimed: oneDNN primitive creation, allocator warm-up
        t0 = time.perf_counter()
        tr.step(x, lab, dropout_mask=mask)
        if i:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    scale = (h * w) / float(full_hw[0] * full_hw[1])
    ram = host_ram_gib()
    return {"value": (n / dt) * scale, "unit": "images/s", "cores": cores, "kind": "port",
            "extrapolated": scale != 1.0, "host_ram_GiB": None if ram is None else round(ram, 1),
            "sample": f"median of {reps} step(s) (after 1 warm-up step) of the CPU oracle (oracle/train_step.py), "
                      f"DeepLabv3-{backbone} batch {n} at {h}x{w} ({dt:.1f} s per step)" +
                      (f", images/s scaled by the pixel ratio {scale:.4f} to {full_hw[0]}x{full_hw[1]}" if scale != 1.0 else
                       ", the full resolution: per-image rate of a batch-2 step, nothing scaled (BASELINE.md section 3)"),
            "step_s": times}


def roofline_from_profile(recs, images_per_step, step_s):
    """Aggregate one instrumented step by kernel instance; the dominant one (by time) is reported."""
    from dcfp_amd import ops, _lib
    which = {"conv_fwd": _lib.CONV_FWD, "conv_dgrad": _lib.CONV_DGRAD, "conv_wgrad": _lib.CONV_WGRAD,
             "conv_dgrad_red": _lib.CONV_DGRAD}
    agg = {}
    conv_flops = conv_ms = bn_bytes = bn_ms = 0.0
    for kind, key, work, ms in recs:
        if kind in which:
            name = ops.conv_kernel_name(key, which[kind])
            if kind == "conv_dgrad_red":      # the fan-in launches that also reduce the previous block's bn3 gradient
                name += " + bn3 sums epilogue"
            a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
            a[0] += work; a[1] += ms; a[2] += 1
            a[3] += work * ops.conv_executed_fraction(key, which[kind])     # MFMA work really issued
            conv_flops += work; conv_ms += ms
        else:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += work; a[1] += ms; a[2] += 1
            bn_bytes += work; bn_ms += ms
    convs = {k: v for k, v in agg.items() if "kernel" in k}
    # dominant = the single-kernel conv entry with the most time (its rocprofv3 row must agree with the live
    # average); the Winograd ops are three or four kernels under one event pair - they are listed in `detail`
    # with the MFMA work their GEMMs issue, and profiles/ holds their components' rows
    single = {k: v for k, v in convs.items() if not k.startswith("winograd")} or convs
    dom = max(single, key=lambda k: single[k][1])
    w, ms, cnt, w_exec = convs[dom]
    # HBM-side bytes per launch from the committed PMC profile (cannot be collected inside this
    # process): corrected FETCH_SIZE + WRITE_SIZE of the same kernel instance on its dominant shape
    traffic = traffic_note = None
    try:
        with open(os.path.join(ROOT, "profiles", "r03_conv_traffic.json")) as f:
            tk = json.load(f)["kernels"].get(dom)
        if tk:
            traffic = tk["fetch_bytes_corrected"] + tk["write_bytes"]
            traffic_note = ("bytes/launch beyond L2 (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, "
                            "profiles/r03_conv_traffic_pmc.txt): " + tk["note"] +
                            f"; algorithmic bytes of that launch {tk['algorithmic_bytes'] / 1e6:.0f} MB")
    except Exception:
        pass
    # the split kernels issue 6 bf16 MFMA products per fp32 product: their ceiling is the dense
    # bf16 MFMA peak / 6 in fp32-equivalent FLOP/s
    split = dom.startswith(("igemm3_kernel", "wgrad3_kernel"))
    peak = PEAK_BF16_MFMA / 6.0 if split else PEAK_F32_MFMA
    wino = dom.startswith("winograd")
    if wino:     # a Winograd op (two transform passes + batched GEMM): the MFMA work issued, not the direct-conv count
        w_nominal, w = w, w_exec
    roof = {"bound": "mfma", "kernel": dom, "achieved": w / (ms * 1e-3) / 1e12, "peak": peak / 1e12,
            "unit": "TFLOP/s", "frac": w / (ms * 1e-3) / peak,
            # `achieved` counts the nominal multiply-adds of the convolutions (padded taps included - SURVEY 8d);
            # executed_frac counts only the MFMAs issued (K-steps of kernel rows lying wholly in the padding are
            # skipped for the ASPP / layer4 dilations): that is the matrix cores' real utilisation
            "executed_frac": w_exec / (ms * 1e-3) / peak, "traffic": traffic,
            "traffic_note": traffic_note,
            "launches_per_step": cnt, "avg_launch_ms": ms / cnt, "flop_per_launch": w / cnt,
            **({"nominal_TFLOP/s": w_nominal / (ms * 1e-3) / 1e12,
                "note": "Winograd F(2x2,3x3): achieved = MFMA FLOPs issued by the batched GEMM / time of the whole op "
                        "(input transform + GEMM + output transform); nominal = direct-conv FLOP count / same time"}
               if wino else {}),
            "dtype": "f32 as 3 bf16 planes (6 x v_mfma_f32_32x32x16_bf16 per product)" if split
                     else "f32 (v_mfma_f32_32x32x2_f32)"}
    # the whole family of the dominant kernel (every launch whose entry name starts with the same kernel: for the 1x1 convs
    # also the fan-in launches that carry the bn3 sums epilogue and are listed as an entry of their own) - the figure that
    # stays like for like from round to round
    fam_key = dom.split("<")[0].split(" ")[0]
    fam = [v for k, v in convs.items() if k.startswith(fam_key)]
    fw, fms, fcnt = sum(v[0] for v in fam), sum(v[1] for v in fam), sum(v[2] for v in fam)
    roof.update({"family": fam_key, "family_frac": fw / (fms * 1e-3) / peak, "family_TFLOP/s": fw / (fms * 1e-3) / 1e12,
                 "family_launches_per_step": fcnt, "family_ms_per_step": fms})
    others = {k: {"TFLOP/s": v[0] / (v[1] * 1e-3) / 1e12, "executed_TFLOP/s": v[3] / (v[1] * 1e-3) / 1e12,
                  "ms_per_step": v[1], "launches": v[2]}
              for k, v in sorted(convs.items(), key=lambda kv: -kv[1][1])}
    hbm = {k: {"GB/s": v[0] / (v[1] * 1e-3) / 1e9, "ms_per_step": v[1], "launches": v[2]}
           for k, v in agg.items() if "kernel" not in k}
    extra = {"conv_ms_per_step": conv_ms, "conv_TFLOP/s_over_all_convs": conv_flops / (conv_ms * 1e-3) / 1e12,
             "bn_ms_per_step": bn_ms, "bn_GB/s": bn_bytes / (bn_ms * 1e-3) / 1e9 if bn_ms else None,
             "step_conv_roofline_frac": conv_flops / step_s / PEAK_F32_MFMA,
             "conv_kernels": others, "hbm_kernels": hbm}
    return roof, extra


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tail(path, n=30):
    try:
        with open(path, "r", errors="replace") as f:
            return "".join(f.readlines()[-n:])
    except OSError:
        return ""


def launch_ranks(n, argv, worker=None, device_count=None, poll_s=0.2, timeout_s=480.0):
    """`python bench.py --gpus N` without a launcher: start N CHILD worker processes, one per GPU
    (scripts/cs/pretrain.sh:31 + engine.py:38-46: one process per device, env:// rendezvous), wait
    for them and forward rank 0's JSON line.  Runs before anything in this process touches the GPU
    (torch.cuda.device_count() does not initialise it), never exec()s, and fails non-zero when
    fewer than N devices are visible instead of silently measuring one.
    The wait has a DEADLINE (`timeout_s`, --rank-timeout): a rank stuck in the RCCL rendezvous or in a collective must
    not hang the caller.  On a failure or at the deadline exactly the children started here are stopped, and the
    parent says which ranks were still alive and prints the tail of every rank's stderr (RCCL warnings included:
    the children run with NCCL_DEBUG=WARN unless the caller set it)."""
    have = torch.cuda.device_count() if device_count is None else device_count
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) visible; refusing to report a "
              f"{have}-GPU number as an {n}-GPU result", file=sys.stderr)
        return 2
    worker = worker or [sys.executable, os.path.abspath(__file__)]
    port = _free_port()
    import tempfile
    import threading
    logdir = tempfile.mkdtemp(prefix="dcfp_bench_ranks_")
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")
        errs.append(open(os.path.join(logdir, f"rank{r}.stderr"), "w"))
        procs.append(subprocess.Popen(worker + list(argv), env=env, text=True, stderr=errs[r],
                                      stdout=subprocess.PIPE if r == 0 else errs[r]))
    # drain rank 0's stdout on a thread so a chatty worker cannot fill the pipe
    lines = []
    t = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    t.start()
    rc = 0
    pending = set(range(n))
    deadline = time.monotonic() + timeout_s
    while pending and rc == 0:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
        if pending and rc == 0 and time.monotonic() > deadline:
            rc = 124
            print(f"bench.py: deadline of {timeout_s:.0f} s passed with rank(s) {sorted(pending)} of {n} still running "
                  f"(rendezvous on 127.0.0.1:{port}); stopping them", file=sys.stderr)
        time.sleep(poll_s)
    for r in pending:          # a rank failed or the deadline passed: stop exactly the children started here
        procs[r].terminate()
    for r in pending:
        try:
            procs[r].wait(timeout=30)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    t.join(timeout=30)
    for f in errs:
        f.close()
    for r in range(n):          # the workers' stderr: everything when all went well is just forwarded, tails on failure
        text = _tail(os.path.join(logdir, f"rank{r}.stderr"), 40 if rc else 10 ** 6)
        if text:
            print(f"---- rank {r} stderr{' (tail)' if rc else ''} ----\n{text}", file=sys.stderr, end="")
    if rc:
        return rc
    recs = [l for l in lines if l.startswith("{")]
    if not recs:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    rec = json.loads(recs[-1])
    if rec.get("n_gpus") != n or rec.get("rccl_ranks") != n:
        print(f"bench.py: asked for {n} ranks, rank 0 reports n_gpus={rec.get('n_gpus')} "
              f"rccl_ranks={rec.get('rccl_ranks')}", file=sys.stderr)
        return 1
    print(recs[-1].rstrip("\n"))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--size", default="1024,2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample", default="auto",
                    help="n,h,w of the CPU baseline sample; auto = batch 2 at the full resolution when host RAM >= 150 GiB, "
                         "else 2,512,1024 scaled by the pixel ratio")
    ap.add_argument("--channel-cfg", default=None,
                    help="time the slim model described by this channel_cfg.pth instead (not the headline config)")
    ap.add_argument("--alt-legs", action="store_true",
                    help="also re-time the step in child processes with DCFP_CONV_WINOGRAD=0 (direct kernels only) and "
                         "with the opt-in DCFP_CONV_MATH=bf16x3 (off-contract: narrower multiplicands); off by default")
    ap.add_argument("--no-alt", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rank-timeout", type=float, default=480.0,
                    help="seconds the --gpus N parent waits for its ranks before it stops them and reports")
    ap.add_argument("--force-ddp", action="store_true",
                    help="wrap in SyncBN+DDP and run the collectives even at world size 1 (rehearsal)")
    ap.add_argument("--syncbn-p2p", action="store_true",
                    help="data-parallel runs: SyncBN statistics through the peer-to-peer exchange kernel "
                         "(dcfp_amd/syncbn_p2p.py) instead of one RCCL collective per layer; off by default")
    args, _ = ap.parse_known_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.rank_timeout))   # parent: no GPU call before this point
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} contradicts WORLD_SIZE={world} set by the launcher")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON record: keep the real stdout aside and point fd 1 at stderr, so that
    # library chatter (RCCL prints its version banner to stdout when the communicator is created) cannot get in
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the HIP path: no GPU visible")
    from dcfp_amd import _lib
    _lib.lib()  # fail loudly if libdcfp_hip.so is missing
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://")
    elif args.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["DCFP_FORCE_SYNCBN"] = "1"
        dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
    H, W = [int(v) for v in args.size.split(",")]
    if args.syncbn_p2p:
        os.environ["DCFP_SYNCBN_P2P"] = "1"

    from dcfp_amd import optimizer as opt, pruners, ops, syncbn_p2p
    from dcfp_amd.engine import Engine
    torch.manual_seed(12345 + rank)           # train.py:166-171
    seg_model = build_model(args.backbone, device, args.channel_cfg)
    optimizer = opt.build_optimizer(_OptArgs, seg_model)
    optimizer.zero_grad()
    train_pruning = pruners.dcfp_pruning(seg_model, 0.999)
    engine = Engine(custom_parser=argparse.ArgumentParser())
    if args.force_ddp and world == 1:
        engine.distributed = True
    ddp = world > 1 or args.force_ddp
    model = engine.data_parallel(seg_model) if ddp else seg_model
    images, labels = synthetic_batch(args.batch, H, W, 12345 + rank, device)
    max_iter = 4000

    def step(it):
        optimizer.zero_grad()                 # train.py:256, verbatim (set_to_none under torch 2.x)
        opt.adjust_learning_rate(optimizer, 0.01, it, max_iter, 0.9, -1)
        loss = model(images, labels, deepsup=True)
        reduce_loss = engine.all_reduce_tensor(loss["loss"]) if ddp else loss["loss"]
        val = reduce_loss.item()              # the reference syncs here every iteration (train.py:263)
        # the peer-to-peer SyncBN exchange poisons its outputs with NaN when it gives up on a peer: ask it first, so that
        # the message names the exchange and the rank (no-op unless --syncbn-p2p; one 4-byte read beside the sync above)
        syncbn_p2p.check_all()
        if val != val:
            ops.check_fused_status()          # a fused BatchNorm backward that gave up poisons gradients with NaN: say so
            raise RuntimeError("loss is NaN")
        loss["loss"].backward()
        train_pruning.step(seg_model)
        optimizer.step()
        return val

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        step(it); it += 1
    fence()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for k in range(args.steps):
        last = step(it); it += 1
        marks[k + 1].record()              # per-step device time (the headline is the wall clock around all K steps)
    fence()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    global_batch = args.batch * world
    value = global_batch * args.steps / dt
    step_s = dt / args.steps

    roof, extra, comm = None, None, None
    reducer = getattr(model, "reducer", None) if ddp else None
    if not args.no_roofline:
        if reducer is not None:
            reducer.timing = True
        ops.profile_start()
        step(it); it += 1
        recs = ops.profile_stop()
        if reducer is not None:
            reducer.timing = False
            # data-parallel step: what the collectives cost where they are exposed.  The 115 forward SyncBN all-gathers sit
            # on the compute stream (HIP-event time of each); the gradient all-reduces are overlapped with backward - what
            # is left of them is the wait at the end of backward; `alone` = the same all-reduces with nothing to overlap.
            kinds = ("syncbn_allgather", "syncbn_p2p_fwd")
            sync = [ms for (kind, _, _, ms) in recs if kind in kinds]
            # backward: the exchange of [sum g, sum g (x - mean)] runs beside the weight gradient enqueued behind it; what
            # the compute stream still waits for (event pair around each wait) and the synchronous ones (no weight gradient
            # to hide behind: p2p kernel on the compute stream)
            bwd_wait = [ms for (kind, _, _, ms) in recs if kind in ("syncbn_bwd_wait", "syncbn_p2p_bwd")]
            exposed = reducer.exposed_ms()
            alone = reducer.alone_ms()
            comm = {"syncbn_exposed_ms": sum(sync), "syncbn_allgathers": len(sync),
                    "syncbn_bwd_exposed_ms": sum(bwd_wait), "syncbn_bwd_exchanges": len(bwd_wait),
                    "syncbn_exchange": "p2p kernel" if any(r[0] == "syncbn_p2p_fwd" for r in recs) else "rccl",
                    "grad_allreduce_ms": alone, "grad_allreduce_exposed_ms": exposed,
                    "allreduce_overlapped_frac": (1.0 - exposed / alone) if (alone and exposed is not None) else None}
            recs = [r for r in recs if r[0] not in kinds + ("syncbn_bwd_wait", "syncbn_p2p_bwd")]
        if rank == 0:
            roof, extra = roofline_from_profile(recs, args.batch, step_s)
    fence()
    peak_mem = torch.cuda.max_memory_allocated() / 2**30
    sgd_rebuilds = getattr(optimizer, "table_rebuilds", None)   # pointer-table uploads (2 = built once per group)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.cpu_sample == "auto":
            # batch 2 per image; at the FULL resolution when the host has the memory for it
            # (R101 batch 2 at 1024x2048 keeps ~40 GB of activations in fp32), else at 512x1024 scaled by the pixel ratio
            ram = host_ram_gib() or 0.0
            full = ram >= 150.0 and (H, W) == (1024, 2048)
            n, h, w, reps = (2, H, W, 1) if full else (2, 512, 1024, 3)
        else:
            (n, h, w), reps = [int(v) for v in args.cpu_sample.split(",")], 3
        cpu = cpu_baseline(args.backbone, n, h, w, (H, W), reps)

    # Reported beside the headline, never as it: the same step with the opt-in 3-way bf16 split
    # conv kernels (fp32-grade products on the bf16 matrix cores; DESIGN.md "bf16x3").  The library
    # reads the switch once per process, so the leg runs in a child after this process let go of
    # its device memory.
    alt = direct = None
    if (rank == 0 and world == 1 and args.alt_legs and not args.force_ddp and not args.channel_cfg
            and os.environ.get("DCFP_CONV_MATH", "") == "" and os.environ.get("DCFP_CONV_WINOGRAD", "") == ""):
        del model, seg_model, optimizer, train_pruning, images, labels
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--backbone", args.backbone, "--batch", str(args.batch), "--size", args.size,
               "--no-cpu-baseline", "--no-roofline"]

        def leg(env_extra, what):
            try:
                r = subprocess.run(cmd, env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
                rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                return {"what": what, "value": rec["value"], "unit": "images/s", "ms_per_step": rec["ms_per_step"],
                        "final_loss": rec["final_loss"]}
            except Exception as exc:  # the legs are informational; the headline above stands on its own
                return {"what": what, "error": repr(exc)[:200]}
        # the same step on the direct conv kernels only (no Winograd): what the headline would be without the
        # algebraic restructuring of the wide 3x3 convs; same fp32 multiplicands either way
        direct = leg({"DCFP_CONV_WINOGRAD": "0"},
                     "DCFP_CONV_WINOGRAD=0: every conv on the direct implicit-GEMM kernels (round-2 kernels of DESIGN 3a)")
        alt = leg({"DCFP_CONV_MATH": "bf16x3"},
                  "bf16x3: fp32 operands split into 3 bf16 planes, 6 bf16 MFMA products per fp32 product, fp32 "
                  "accumulate (opt-in DCFP_CONV_MATH=bf16x3; not the headline)")
        alt["math"] = "bf16x3"

    if rank == 0:
        out = {"metric": "training images/sec at 1024x2048 DeepLabv3-R101", "value": value, "unit": "images/s",
               "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "median_ms_per_step": median_ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if os.environ.get("DCFP_CONV_MATH", "") in ("", "f32") else "f32 via bf16x3 split",
               "data": "synthetic",
               "config": {"workload": ("PRUNED (" + args.channel_cfg + ") " if args.channel_cfg else "") +
                                      f"DeepLabv3-{args.backbone}+ASPP os8, {args.batch}x3x{H}x{W} per GPU, "
                                      "CE+0.4*deepsup CE (fused upsample), " +
                                      ("" if os.environ.get("DCFP_CONV_WINOGRAD", "") == "0" else
                                       "3x3 stride-1 convs with >= 256 channels as fp32 Winograd F(2x2,3x3) "
                                       "(fwd / dgrad / wgrad), all other convs direct implicit GEMM, ") +
                                      ("SyncBN + gradient all-reduce over RCCL, " if ddp else "") +
                                      "EIC step, SGD m0.9 wd5e-4",
                          "global_batch": global_batch, "parallelism": f"dp{world}"},
               "final_loss": last, "peak_mem_GiB": peak_mem,
               "grad_allreduce_launches_per_step": getattr(getattr(model, "reducer", None), "launched", None) if ddp else None,
               "sgd_table_rebuilds": sgd_rebuilds, "comm": comm,
               "conv_roofline_images_per_s_per_gpu_at_100pct": 11.97 if (H, W, args.backbone, args.channel_cfg) == (1024, 2048, "resnet101", None) else None,
               "roofline": roof, "cpu_baseline": cpu, "direct_conv": direct, "alt_math": alt, "detail": extra}
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    ops.check_fused_status()           # a fused BatchNorm backward that gave up is an error, not a number
    if dist.is_initialized():
        from dcfp_amd import syncbn_p2p
        syncbn_p2p.finish()            # an exchange that gave up on a peer is an error, not a number
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
