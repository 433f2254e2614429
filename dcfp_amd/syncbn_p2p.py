"""Peer-to-peer SyncBatchNorm exchange: host side of csrc/syncbn_p2p.hip.

nn.SyncBatchNorm (engine.py:65 of the reference) costs one all-gather per BatchNorm layer forward and one
all-reduce backward - 230 latency-bound collectives per DeepLabv3-R101 step.  `SyncBnP2P` replaces each of them
with ONE single-workgroup kernel that writes this rank's row straight into every peer's mailbox over xGMI, waits
for the peers' rows in its own mailbox and reduces them in rank order (same numbers as the collective path, bit
for bit; identical on every rank).

Set-up (once per process group): allocate the mailbox (fine-grained device memory), export its IPC handle, exchange
the handles over the host-side process group, map every peer's mailbox, barrier.  After that an exchange is one
kernel launch on the caller's stream; the sequence number advances by one per exchange on every rank.

Opt-in: DCFP_SYNCBN_P2P=1 (Engine.data_parallel turns it on for the world group).  Off by default until it has
been timed against RCCL on an 8-GPU node; with it off nothing here runs.
DCFP_P2P_MEM=0|1|2 picks the allocation kind (fine-grained / uncached / plain), DCFP_P2P_SPIN the polling rounds
before the kernel gives up (default 200 000 000 of ~0.55 us each, about two minutes).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib

_ACTIVE = {}      # process group -> SyncBnP2P


def wanted():
    return os.environ.get("DCFP_SYNCBN_P2P", "0") not in ("0", "")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _StreamWork:
    """What sync_bn_bwd_sums hands back for an exchange launched on the side stream: wait() makes the
    current stream wait for it (the interface of the c10d Work the RCCL path returns)."""

    def __init__(self, event, held):
        self.event = event
        self.held = held      # the exchange's input and output: alive (and out of the allocator's hands) until wait()

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        self.held = None
        return True


class SyncBnP2P:
    def __init__(self, group, device, max_channels=2048):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        if self.world > 8:
            raise RuntimeError("SyncBnP2P: at most 8 ranks (one xGMI node)")
        self.cap = (2 * int(max_channels) + 1 + 31) // 32 * 32
        self.spin = int(os.environ.get("DCFP_P2P_SPIN", "200000000"))
        L = _lib.lib()
        nbytes = L.dcfp_syncbn_p2p_mailbox_bytes(self.world, self.cap)
        if nbytes == 0:
            raise RuntimeError("SyncBnP2P: bad mailbox geometry")
        # Set-up fails on ALL ranks or on none: every rank reaches both object all-gathers whatever happened to it
        # locally, and the verdict is the AND of everybody's (a rank raising alone would leave the others in a barrier).
        self.local, self._mapped, err, raw = None, [], None, None
        self.boxes = (C.c_void_p * self.world)()
        with torch.cuda.device(self.device):
            try:
                ptr = C.c_void_p()
                _lib.check(L.dcfp_p2p_alloc(nbytes, int(os.environ.get("DCFP_P2P_MEM", "0")), C.byref(ptr)), "p2p_alloc")
                self.local = ptr
                handle = C.create_string_buffer(64)
                _lib.check(L.dcfp_p2p_export(self.local, handle), "p2p_export (hipIpcGetMemHandle)")
                raw = handle.raw
            except Exception as e:       # noqa: BLE001 - reported through the all-gather below
                err = "rank %d: %s" % (self.rank, e)
            rows = [None] * self.world
            dist.all_gather_object(rows, (raw, os.getpid(), err), group=group)
            if err is None and all(r[2] is None for r in rows):
                try:
                    for r, (peer_raw, pid, _) in enumerate(rows):
                        if r == self.rank:
                            self.boxes[r] = self.local
                            continue
                        if pid == os.getpid():
                            raise RuntimeError("two ranks in one process")
                        ptr = C.c_void_p()
                        _lib.check(L.dcfp_p2p_import(C.create_string_buffer(peer_raw, 64), C.byref(ptr)),
                                   "p2p_import (hipIpcOpenMemHandle) of rank %d" % r)
                        self.boxes[r] = ptr
                        self._mapped.append(ptr)
                except Exception as e:   # noqa: BLE001
                    err = "rank %d: %s" % (self.rank, e)
            errs = [None] * self.world
            dist.all_gather_object(errs, err if err is not None else next((r[2] for r in rows if r[2]), None),
                                   group=group)
            bad = [e for e in errs if e]
            if bad:
                for ptr in self._mapped:
                    L.dcfp_p2p_unmap(ptr)
                self._mapped = []
                dist.barrier(group=group)      # everyone has unmapped before anyone frees
                if self.local is not None:
                    L.dcfp_p2p_free(self.local)
                    self.local = None
                raise RuntimeError("SyncBnP2P set-up failed (%s)" % "; ".join(sorted(set(bad))))
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.seq = 0
        self.exchanges = 0
        self._side = None
        dist.barrier(group=group)          # nobody posts before everyone has mapped everyone

    # -- one exchange on the current stream
    def exchange(self, local, out, mode, run=None):
        n = local.numel()
        if not (local.is_cuda and local.is_contiguous() and local.dtype == torch.float32 and out.is_contiguous()):
            raise RuntimeError("SyncBnP2P.exchange: contiguous fp32 device tensors only")
        if n > self.cap:
            raise RuntimeError("SyncBnP2P: %d floats exceed the mailbox capacity %d" % (n, self.cap))
        if out.numel() != (self.world * n if mode == 0 else n):
            raise RuntimeError("SyncBnP2P.exchange: wrong output size")
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        # the sequence number advances only for an exchange that was really launched: a launch error on one rank must
        # not shift its numbering against the peers' for the rest of the run (it is fatal for the group anyway: raised)
        _lib.check(_lib.lib().dcfp_syncbn_p2p_exchange_f32(self.boxes, self.world, self.rank, self.seq + 1, self.cap,
                                                           _p(local), n, mode, _p(out), run, self.spin,
                                                           _p(self.status), stream), "syncbn_p2p_exchange")
        self.seq += 1
        self.exchanges += 1
        return out

    def exchange_async(self, local, out, mode):
        """The exchange on a side stream behind everything the current stream has queued; returns a work object whose
        wait() orders the current stream behind it.  Kernels enqueued in between overlap the wait for the peers."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.exchange(local, out, mode)
            ev = torch.cuda.Event()
            ev.record(self._side)
        # `local` and `out` came from the CURRENT stream's allocator pool but are read / written on the side stream: without
        # this the caching allocator may hand `local`'s block to the next current-stream allocation (the weight gradient
        # enqueued in between, by design) before the exchange kernel has read it.  c10d does the same for its collectives.
        local.record_stream(self._side)
        out.record_stream(self._side)
        return _StreamWork(ev, (local, out))

    def check(self):
        """Raise if any exchange timed out (blocking read of the status word: once per step, next to the step's own
        host synchronisation - engine.DataParallel.check_exchange(), bench.py and tools/train.py call it there, so that a
        dead peer stops the job at the first step instead of after ~230 two-minute time-outs per step)."""
        s = int(self.status.item())
        if s != 0:
            raise RuntimeError("SyncBnP2P: exchange %d gave up waiting for a peer (rank %d of %d; its outputs were "
                               "poisoned with NaN)" % (s, self.rank, self.world))

    def close(self):
        if self.local is None:
            return
        torch.cuda.synchronize(self.device)
        L = _lib.lib()
        try:
            dist.barrier(group=self.group)     # every rank's kernels are done with every mailbox
        except Exception:
            pass
        for ptr in self._mapped:
            L.dcfp_p2p_unmap(ptr)
        self._mapped = []
        try:
            dist.barrier(group=self.group)     # ... and has unmapped before any owner frees
        except Exception:
            pass
        L.dcfp_p2p_free(self.local)
        self.local = None


def enable(group, device, max_channels=2048):
    """Create (once) the exchange object of `group`; sync_bn_stats / sync_bn_bwd_sums of ops.py then use it."""
    px = _ACTIVE.get(group)
    if px is None:
        px = SyncBnP2P(group, device, max_channels)
        _ACTIVE[group] = px
    return px


def for_group(group):
    return _ACTIVE.get(group) if _ACTIVE else None


def disable(group=None):
    for g in ([group] if group is not None else list(_ACTIVE)):
        px = _ACTIVE.pop(g, None)
        if px is not None:
            px.close()


def check_all():
    """Once per step: raise on the first exchange any active group gave up on (no-op without the peer-to-peer exchange)."""
    for px in list(_ACTIVE.values()):
        px.check()


def finish():
    """End of a run: raise if any exchange gave up on a peer, then release every mailbox (bench.py, tools/train.py)."""
    try:
        for px in list(_ACTIVE.values()):
            px.check()
    finally:
        disable()
