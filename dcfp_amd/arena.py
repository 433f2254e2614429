"""Parameter / gradient / momentum arenas and the gradient reducer of the data-parallel step.

Reference: the step either side of backward — `build_optimizer` + `optimizer.step()`
(optimizer.py:12-49), `optimizer.zero_grad()` (train.py:256), `dcfp_pruning.step` reading every
scored BN's `weight.grad` (pruners/dcfp_pruner.py:15-20) and DDP's bucketed gradient all-reduce
(engine.py:65-68).  The reference leaves all of that to ~470 separately allocated tensors.

Here every parameter of a model is a VIEW into one flat fp32 buffer, every gradient a view into a
second one and every momentum buffer a view into a third (65 M floats = 260 MB each for
DeepLabv3-R101, 256-byte aligned slots).  Consequences:

  * addresses never change, so the pointer tables of the multi-tensor HIP kernels (SGD, EIC) are
    built ONCE instead of every step (`zero_grad(set_to_none=True)` used to hand out new gradient
    tensors each iteration);
  * the backward kernels (wgrad split-K reduce, BN dgamma/dbeta finalize, bias gradient) write their
    result straight into the gradient view - no per-parameter allocation, no AccumulateGrad copy;
  * `zero_grad` costs nothing (views are detached, the first write of the next backward overwrites)
    or ONE memset of the arena for the in-place flavour;
  * under data parallelism the gradient exchange is an all-reduce of a few contiguous arena ranges,
    issued on RCCL's stream as soon as the last gradient of a range has been written (ranges are cut
    in parameter order = reverse backward order) and therefore overlapped with the rest of backward.
    BN-gamma gradients live in the same ranges: after the exchange the EIC update is rank-identical
    (the "importance-score all-reduce").

Host logic only (torch supplies memory, streams and collectives); it also runs on CPU tensors so the
exchange is covered by world-size-2 gloo tests.
"""
import os

import torch
import torch.distributed as dist

ALIGN = 64            # floats: every slot starts on a 256-byte boundary
DIRECT = os.environ.get("DCFP_ARENA_DIRECT", "1") not in ("0",)   # =0: gradients go through autograd (torch DDP)


class GradSlot:
    """What a Parameter carries (`p._dcfp_slot`) so that a backward kernel can find its gradient view."""
    __slots__ = ("arena", "index")

    def __init__(self, arena, index):
        self.arena, self.index = arena, index


class ParamArena:
    def __init__(self, params):
        params = [p for p in dict.fromkeys(params)]          # unique, order kept
        if not params:
            raise ValueError("ParamArena: no parameters")
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError("ParamArena: parameters must be fp32 tensors on one device")
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_mom = None
        self.grad_views = []
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(params, self.offsets)):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                gv = self.flat_grad[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    gv.copy_(p.grad)
                    p.grad = gv
                self.grad_views.append(gv)
                p._dcfp_slot = GradSlot(self, i)
        self.reducer = None
        self.epoch = 0           # bumped when the set of live gradients may have changed

    # ------------------------------------------------------------------ lookup / validity
    @staticmethod
    def of(params):
        """The arena that already holds exactly these parameters - in ANY order: the optimizer walks them by param
        group (decay / no-decay, optimizer.py:18-33), the data-parallel wrapper in module order - and is still valid;
        otherwise a new one.  An arena that carries a gradient reducer is never replaced silently: its parameters
        would lose the exchange (ranks stepping on local gradients without an error)."""
        params = [p for p in dict.fromkeys(params)]
        slot = getattr(params[0], "_dcfp_slot", None) if params else None
        if slot is not None and slot.arena.covers(params):
            return slot.arena
        for p in params:
            old = getattr(p, "_dcfp_slot", None)
            if old is not None and old.arena.reducer is not None and old.arena.live(old.index):
                raise RuntimeError(
                    "ParamArena.of: these parameters belong to an arena with a gradient reducer (data-parallel "
                    "wrapper) but are not exactly its parameter set; build the optimizer over the same trainable "
                    "parameters as Engine.data_parallel, or wrap the model after changing them")
        return ParamArena(params)

    def covers(self, params):
        """True when `params` is exactly this arena's parameter set (order-independent) and every one of them still
        lives in it (not after model.to(...) / a load by assignment)."""
        if len(params) != len(self.params):
            return False
        base = self.flat_param.data_ptr()
        seen = 0
        for p in params:
            slot = getattr(p, "_dcfp_slot", None)
            if slot is None or slot.arena is not self or self.params[slot.index] is not p:
                return False
            if p.data_ptr() != base + 4 * self.offsets[slot.index]:
                return False
            seen += 1
        return seen == len(self.params) and len(set(map(id, params))) == seen

    def live(self, index):
        p = self.params[index]
        return p.data_ptr() == self.flat_param.data_ptr() + 4 * self.offsets[index]

    def momentum(self):
        if self.flat_mom is None:
            self.flat_mom = torch.zeros_like(self.flat_param)
        return self.flat_mom

    def momentum_view(self, index):
        o, p = self.offsets[index], self.params[index]
        return self.momentum()[o:o + p.numel()].view(p.shape)

    # ------------------------------------------------------------------ gradients
    def zero_grad(self, set_to_none=True):
        """set_to_none: detach every view (free: the next backward's first write overwrites).
        Otherwise ONE fill of the flat buffer, gradients stay attached (torch 1.10's default, train.py:256)."""
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.flat_grad.zero_()
            for p, gv in zip(self.params, self.grad_views):
                p.grad = gv
        self.epoch += 1

    def grad_target(self, param):
        """Where a backward kernel should write the gradient of `param`: (tensor, token).
        token 1: the arena view itself, first gradient since zero_grad(set_to_none=True);
        token 2: a temporary that grad_commit adds onto the attached view (in-place zero_grad, several
                 backward passes);  token 0: a fresh tensor handed back to autograd."""
        slot = getattr(param, "_dcfp_slot", None)
        if slot is not None and DIRECT and slot.arena is self and self.live(slot.index):
            view = self.grad_views[slot.index]
            if param.grad is None:
                return view, 1
            if param.grad.data_ptr() == view.data_ptr():
                return torch.empty_like(view), 2
        return torch.empty_like(param, memory_format=torch.contiguous_format), 0

    def grad_commit(self, param, out, token, add=None):
        """After the kernel filled `out`: returns what the autograd Function returns for this input
        (None when the gradient already sits in param.grad)."""
        if token == 0:
            return out
        idx = param._dcfp_slot.index
        view = self.grad_views[idx]
        if token == 1:
            param.grad = view
        else:
            if self.reducer is not None:
                self.reducer.check_not_in_flight(idx)
            if add is not None:
                add(view, out)
            else:
                view.add_(out)
        if self.reducer is not None:
            self.reducer.mark_ready(idx)
        return None


def grad_target(param):
    """Module-level helpers for the autograd Functions (direct write when `param` lives in an arena)."""
    slot = getattr(param, "_dcfp_slot", None) if param is not None else None
    if slot is None:
        return torch.empty_like(param, memory_format=torch.contiguous_format), 0
    return slot.arena.grad_target(param)


def grad_commit(param, out, token, add=None):
    if token == 0:
        return out
    return param._dcfp_slot.arena.grad_commit(param, out, token, add)


class GradReducer:
    """Averages the gradient arena over the ranks of `group` in `n_chunks` contiguous all-reduces,
    each issued (asynchronously, on the collective's own stream) the moment every gradient of its range
    has been written during backward; a callback queued on the autograd engine waits for them at the end
    of backward, so `loss.backward()` returns with averaged gradients in place, as DDP does
    (engine.py:65-68).  Chunks are cut in parameter order: the LAST chunk (heads, ASPP, layer4) completes
    first and its exchange overlaps the backward of everything before it.  xGMI rings are per-link
    bound, so few large messages (~87 MB each for R101) beat many small buckets."""

    def __init__(self, arena, group=None, n_chunks=3):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group)
        n = len(arena.params)
        n_chunks = max(1, min(n_chunks, n))
        cuts = [0]
        for c in range(1, n_chunks):
            tgt = arena.total * c / n_chunks
            i = next((k for k in range(n) if arena.offsets[k] >= tgt), n)
            i = min(max(i, cuts[-1] + 1), n - (n_chunks - c))
            cuts.append(i)
        cuts.append(n)
        self.bounds = []                 # (first param index, last+1, flat start, flat end)
        for a, b in zip(cuts[:-1], cuts[1:]):
            self.bounds.append((a, b, arena.offsets[a], arena.total if b == n else arena.offsets[b]))
        self.chunk_of = [0] * n
        for c, (a, b, _, _) in enumerate(self.bounds):
            for i in range(a, b):
                self.chunk_of[i] = c
        # ncclAvg exists in RCCL; gloo (CPU tests) has no AVG: SUM then one scale of the range
        backend = dist.get_backend(group)
        self.use_avg = backend == "nccl"
        self._active = False
        self.launched = 0                # all-reduces issued in the last backward (bench / tests)
        self.timing = False              # bench.py: record how long finalize() keeps the compute stream waiting
        arena.reducer = self

    def begin_step(self):
        """Start of a training step (DataParallel.forward): drop whatever a backward that raised left behind, and make
        a step that never reaches the exchange visible (`launched` stays 0)."""
        if self._active:
            for w, _ in getattr(self, "works", []):
                w.wait()
        self._active = False
        self.works = []
        self.launched = 0

    def check_not_in_flight(self, index):
        """A second gradient for a parameter whose range is already being all-reduced (a parameter used twice in one
        graph) would race with the collective and miss the average."""
        if self._active and self.pending[self.chunk_of[index]] == -1:
            raise RuntimeError("GradReducer: a second gradient arrived for a parameter whose range is already being "
                               "exchanged (parameter used twice in one backward); use n_chunks=1 ranges cut after "
                               "its last use, or torch_ddp=True")

    def _begin(self):
        self._active = True
        self.pending = [b - a for (a, b, _, _) in self.bounds]
        self.ready = [False] * len(self.arena.params)
        self.works = []
        self.launched = 0
        torch.autograd.Variable._execution_engine.queue_callback(self.finalize)

    def mark_ready(self, index):
        if not self._active:
            self._begin()
        if self.ready[index]:
            return
        self.ready[index] = True
        c = self.chunk_of[index]
        self.pending[c] -= 1
        if self.pending[c] == 0:
            self._launch(c)

    def _launch(self, c):
        _, _, f0, f1 = self.bounds[c]
        buf = self.arena.flat_grad[f0:f1]
        if self.use_avg:
            w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append((w, buf))
        self.pending[c] = -1
        self.launched += 1

    def finalize(self):
        """End of backward: ranges with gradients that never arrived (a head that did not take part in the
        loss) are completed with zeros so that every rank issues the same collectives, then all exchanges
        are waited for (stream-ordered: the compute stream waits, the host does not)."""
        if not self._active:
            return
        for c, (a, b, _, _) in enumerate(self.bounds):
            if self.pending[c] > 0:
                for i in range(a, b):
                    if not self.ready[i]:
                        self.arena.grad_views[i].zero_()
                self._launch(c)
        timed = self.timing and torch.cuda.is_available() and self.arena.flat_grad.is_cuda
        if timed:       # how long the compute stream really waits for the exchanges at the end of backward
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w, buf in self.works:
            w.wait()
            if not self.use_avg:
                buf.div_(self.world)
        if timed:
            e1.record()
            self.exposed_events = (e0, e1)
        self.works = []
        self._active = False

    def exposed_ms(self):
        """Compute-stream time spent waiting for the gradient exchanges at the end of the last backward run with
        `timing` set (the part of the all-reduces that backward did not hide)."""
        ev = getattr(self, "exposed_events", None)
        if ev is None:
            return None
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1])

    def alone_ms(self, reps=3):
        """The same all-reduces issued back to back with nothing to overlap (on a scratch copy of the ranges)."""
        if not self.arena.flat_grad.is_cuda:
            return None
        scratch = torch.zeros_like(self.arena.flat_grad)
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        best = None
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for (_, _, f0, f1) in self.bounds:
                dist.all_reduce(scratch[f0:f1], op=op, group=self.group)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        return best


def reduce_now(arena, group=None):
    """Synchronous flavour for code that filled gradients outside our autograd Functions."""
    red = arena.reducer or GradReducer(arena, group)
    red._active = True
    red.pending = [1] * len(red.bounds)
    red.ready = [True] * len(arena.params)
    red.works = []
    red.launched = 0
    for c in range(len(red.bounds)):
        red._launch(c)
    red.finalize()
