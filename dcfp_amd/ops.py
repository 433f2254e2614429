"""Tensor-level wrappers and autograd Functions over the C-ABI (include/dcfp_hip.h).

PyTorch supplies device memory, streams and autograd bookkeeping; every FLOP and byte
moved on the hot path below happens in libdcfp_hip.so.  Inputs must be CUDA (HIP) fp32
tensors — there is deliberately no CPU / eager fallback.
"""
import ctypes as C
import os
import weakref

import torch
import torch.distributed as dist

from . import _lib, arena, syncbn_p2p
from ._lib import ConvDesc, BnRunning, check

_ws_cache = {}
# Bumped whenever parameters are rewritten through raw pointers (FusedSGD.step, load_state_dict):
# caches derived from weights (folded eval-mode BN, permuted conv weights) key on it because
# torch's per-tensor `_version` does not see those writes.
WEIGHT_EPOCH = [0]

# ---- optional per-launch timing (bench.py's roofline leg): when a list is installed, every
# conv / BN C-ABI call is bracketed by events on the stream it is launched on.
_PROFILE = None


def profile_start():
    global _PROFILE
    _PROFILE = []


def profile_stop():
    """Returns [(kind, key, algorithmic_work, milliseconds)] and disables timing."""
    global _PROFILE
    recs, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    return [(k, key, work, s.elapsed_time(e)) for (k, key, work, s, e) in (recs or [])]


def _timed(kind, key, work, fn):
    if _PROFILE is None:
        return fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    _PROFILE.append((kind, key, work, s, e))
    return r


def conv_kernel_name(d, which):
    buf = C.create_string_buffer(64)
    n = _lib.lib().dcfp_conv2d_kernel_name(C.byref(d), which, buf, 64)
    return buf.value.decode() if n > 0 else "?"


def conv_executed_fraction(d, which):
    """Share of the nominal multiply-adds the dispatched kernel issues (< 1 where dead kernel rows are skipped)."""
    return float(_lib.lib().dcfp_conv2d_executed_fraction(C.byref(d), which))


def _conv_flops(d):
    return 2.0 * d.N * d.Cout * d.Hout * d.Wout * d.Cin * d.KH * d.KW


def _require(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"dcfp_amd: {name} must be a CUDA/HIP tensor (no CPU fallback exists)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"dcfp_amd: {name} must be float32, got {t.dtype}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _workspace(tag, nbytes, device):
    """Grow-only scratch per (tag, device); reuse is safe because launches are stream-ordered."""
    key = (tag, device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _batch_strided(t):
    """Return (tensor, batch_stride) for a [N,C,H,W] tensor whose images are dense
    (a channel slice of a wider tensor qualifies); otherwise a contiguous copy."""
    N, Cc, H, W = t.shape
    st = t.stride()
    if st[3] == 1 and st[2] == W and st[1] == H * W and (N == 1 or st[0] >= Cc * H * W):
        return t, (st[0] if N > 1 else Cc * H * W)
    t = t.contiguous()
    return t, Cc * H * W


def _pitch_of(t):
    """Row pitch of a [N,C,H,W] tensor whose rows are dense but stride(2) > W (a [..., :W] view of a buffer with
    padded rows, channels H*pitch and images C*H*pitch apart); 0 for anything else (dense included)."""
    N, Cc, H, W = t.shape
    st = t.stride()
    if st[3] == 1 and st[2] > W and st[2] % 4 == 0 and st[1] == H * st[2] and (N == 1 or st[0] == Cc * H * st[2]):
        return st[2]
    return 0


_PITCH_BUFFERS = {}


def pitched_buffer(shape, pitch, key, device):
    """[N,C,H,W] view (rows `pitch` floats apart) of a persistent zero-initialised buffer: the kernels write the
    W live floats of a row only, so the tail stays zero for as long as the buffer lives (no per-step fill)."""
    k = (key, tuple(shape), pitch, str(device))
    buf = _PITCH_BUFFERS.get(k)
    if buf is None:
        buf = new_pitched(shape, pitch, device)
        _PITCH_BUFFERS[k] = buf
    return buf


class PitchLease:
    """Held by the autograd node that reads a module's persistent pitched buffer in its backward.  The buffer is busy
    while a lease on it is alive: `release()` (end of that backward) or the death of the node - a grad-enabled forward
    whose graph is dropped without backward, an exception between forward and backward - frees it for the next forward."""
    __slots__ = ("slot", "__weakref__")

    def __init__(self, slot):
        self.slot = slot
        slot["lease"] = weakref.ref(self)

    def release(self):
        if self.slot.get("lease") is not None and self.slot["lease"]() is self:
            self.slot["lease"] = None


def _pitch_busy(slot):
    ref = slot.get("lease")
    return ref is not None and ref() is not None


def owner_pitched(owner, shape, pitch, device, track=True):
    """The persistent row-pitched buffer of a module's output (it lives from the forward to the backward of the
    conv that reads it): (view, lease).  While a graph holds a lease on the buffer, a second forward gets a fresh
    buffer instead (lease None)."""
    slot = getattr(owner, "_dcfp_pitch", None) if owner is not None else None
    if slot is not None and not _pitch_busy(slot) and tuple(slot["y"].shape) == tuple(shape) \
            and _pitch_of(slot["y"]) == pitch and slot["y"].device == device:
        view = slot["y"]
    else:
        view = new_pitched(shape, pitch, device)
        if owner is not None and (slot is None or not _pitch_busy(slot)):
            slot = {"y": view, "lease": None}
            owner._dcfp_pitch = slot
        else:
            slot = None
    return view, (PitchLease(slot) if (slot is not None and track) else None)


def new_pitched(shape, pitch, device):
    """Fresh zeroed row-pitched [N,C,H,W] view.  The allocation starts with a margin of zeros (the contract of
    DcfpConvDesc.x_pitch: the pitch - W floats in front of the first element are readable zeros)."""
    N, Cc, H, W = shape
    lead = 64                                            # floats; keeps the view 256-byte aligned
    flat = torch.zeros(lead + N * Cc * H * pitch, dtype=torch.float32, device=device)
    return flat[lead:].view(N, Cc, H, pitch)[..., :W]


# The 3x3 convs whose taps shift columns by a non-multiple of 4 (dilation 1 / 2) read their shifted operand from a
# row-pitched tensor with a zero tail instead of handling image borders with 4-byte copies (include/dcfp_hip.h
# DcfpConvDesc.x_pitch).  DCFP_PITCHED=0: dense operands everywhere.
PITCHED = os.environ.get("DCFP_PITCHED", "1") not in ("0",)


def conv_pitch(xshape, wshape, stride, pad, dil):
    """Row pitch to use for the x / dy operands of this conv (0: keep them dense)."""
    if not PITCHED or wshape[2] != 3 or stride != 1 or pad != dil or ((pad | dil) & 3) == 0:
        return 0
    W = xshape[3]
    pitch = W + (pad + 3) // 4 * 4
    d = _desc(xshape, wshape, stride, pad, dil, pitch, pitch)
    return pitch if _lib.lib().dcfp_conv2d_pitch_supported(C.byref(d)) else 0


def conv_out_size(size, k, stride, pad, dil):
    return (size + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _desc(xshape, wshape, stride, pad, dil, x_pitch=0, dy_pitch=0):
    N, Cin, H, W = xshape
    Cout, Cin2, KH, KW = wshape
    if Cin2 != Cin:
        raise RuntimeError(f"conv2d: weight expects {Cin2} input channels, got {Cin} (groups unsupported)")
    d = ConvDesc(N, Cin, H, W, Cout, KH, KW, stride, pad, dil,
                 conv_out_size(H, KH, stride, pad, dil), conv_out_size(W, KW, stride, pad, dil),
                 x_pitch, dy_pitch)
    return d


# ------------------------------------------------------------------ conv2d
# The Bottleneck convs emit the batch statistics of their output for the BatchNorm behind them (no
# 4 B/element stats pass): BN -6 ms/step, conv epilogues +1 ms/step since the LDS-DMA kernels merge the
# per-lane partials through LDS instead of 640 swizzles per wave and tile.  DCFP_FUSED_BN_STATS=0: off.
FUSE_BN_STATS = os.environ.get("DCFP_FUSED_BN_STATS", "1") not in ("0",)
# The residual BatchNorm of a Bottleneck keeps its ReLU mask as one bit per element for the backward
# (instead of two re-reads of the 4-byte block output); DCFP_BN_RELU_BITMASK=0 switches it off.
BN_RELU_BITMASK = os.environ.get("DCFP_BN_RELU_BITMASK", "1") not in ("0",)
# =0: the SyncBN backward exchange of the two per-channel sums is waited for before anything else is enqueued (A/B)
SYNCBN_ASYNC = os.environ.get("DCFP_SYNCBN_ASYNC", "1") not in ("0",)


_WP_OWNERS = weakref.WeakSet()     # weight tensors that carry permuted copies (w._dcfp_wp)
_WP_TABLE = {"version": 0, "built": -1, "dev": None, "n": 0, "blocks": 0, "entries": []}


def _wp_buffer(w, which, d, nbytes, variant=""):
    """Persistent buffer for the permuted weight copy Wp of (weight tensor, pass, shape) and whether the
    library may skip rebuilding it (`wp_valid` of include/dcfp_hip.h).  Validity = same storage, same torch
    version counter and same WEIGHT_EPOCH (raw-pointer writes by FusedSGD).  In training every weight changes
    once per step: FusedSGD.step() then calls refresh_wp(), ONE multi-tensor launch that rebuilds every
    registered copy (forward and dgrad layouts of all convs) and re-validates them, instead of a permute
    launch inside each of the 233 conv calls of a step."""
    # (the copy's layout depends on the kernel the shape routes to, which depends on the operand pitches and,
    # for the inference call with its fused epilogue, on the variant: separate entries)
    key = (which, d.N, d.H, d.W, d.stride, d.pad, d.dil, d.x_pitch, d.dy_pitch, variant)
    tag = (w.data_ptr(), w._version, WEIGHT_EPOCH[0], torch.cuda.current_stream().cuda_stream)
    cache = getattr(w, "_dcfp_wp", None)
    if cache is None:
        cache = {}
        try:
            w._dcfp_wp = cache
            _WP_OWNERS.add(w)
        except Exception:        # not an attribute-bearing tensor: no caching
            return _workspace("conv", nbytes, w.device), 0
    ent = cache.get(key)
    if ent is not None and ent[1].numel() >= nbytes:
        if ent[0] == tag:
            return ent[1], 1
        cache[key] = (tag, ent[1], ent[2])
        return ent[1], 0
    if len(cache) > 8:           # shapes keep changing (multi-scale evaluation): do not hoard buffers
        cache.clear()
    buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=w.device)
    cache[key] = (tag, buf, ConvDesc(*[getattr(d, f[0]) for f in ConvDesc._fields_]))
    _WP_TABLE["version"] += 1
    return buf, 0


def _conv_workspace(w, which, d, variant=""):
    """(workspace, wp_valid) of a forward / dgrad call: the conv's own persistent buffer (the permuted weight copy of a
    direct kernel, or the transformed filters of a fused Winograd conv), or - where the library says the workspace is
    pure scratch (three-pass Winograd: transformed operands, hundreds of MB) - one buffer shared by all.  variant: a
    call that does not run what dcfp_conv2d_wp_layout describes for (d, which) keeps a copy of its own, rebuilt on demand."""
    L = _lib.lib()
    nbytes = L.dcfp_conv2d_workspace_bytes(C.byref(d), which)
    if L.dcfp_conv2d_workspace_is_scratch(C.byref(d), which):
        return _workspace("conv_scratch", nbytes, w.device), 0
    return _wp_buffer(w, which, d, nbytes, variant)


def refresh_wp():
    """After an in-place weight update: rebuild every registered Wp copy with one launch and mark them valid
    for the current WEIGHT_EPOCH.  The device table of (weights, copy, layout) records is rebuilt only when
    the set of registered copies changed (first steps of a run)."""
    T = _WP_TABLE
    L = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    if T["built"] != T["version"]:
        ents, recs, first = [], [], 0
        for w in list(_WP_OWNERS):
            if not w.is_cuda or not w.is_contiguous():
                continue
            for key, (tag, buf, desc) in list(w._dcfp_wp.items()):
                if key[-1]:          # inference-variant copies rebuild on demand (their layout is the call's own)
                    continue
                e = _lib.WpEntry()
                if L.dcfp_conv2d_wp_layout(C.byref(desc), key[0], C.byref(e)) != 0:
                    continue
                e.w, e.wp, e.first_block = w.data_ptr(), buf.data_ptr(), first
                first += e.n_blocks
                recs.append(e); ents.append((weakref.ref(w), key))
        T["entries"], T["n"], T["blocks"] = ents, len(recs), first
        if recs:
            arr = (_lib.WpEntry * len(recs))(*recs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            T["dev"] = host.to(torch.device("cuda", torch.cuda.current_device()))
        T["built"] = T["version"]
        T["ptrs"] = [e.w for e in recs]
    if not T["n"]:
        return
    live = []
    for (wr, key), ptr in zip(T["entries"], T["ptrs"]):
        w = wr()
        if w is None or w.data_ptr() != ptr or key not in w._dcfp_wp:     # a weight went away / moved: rebuild next time
            T["version"] += 1
            return
        live.append((w, key))
    check(L.dcfp_conv2d_permute_weights_multi_f32(C.c_void_p(T["dev"].data_ptr()), T["n"], T["blocks"],
                                                  C.c_void_p(stream)), "permute_weights_multi")
    for w, key in live:
        _, buf, desc = w._dcfp_wp[key]
        w._dcfp_wp[key] = ((w.data_ptr(), w._version, WEIGHT_EPOCH[0], stream), buf, desc)


def _bn_run(running_mean, running_var, momentum, nbt):
    """DcfpBnRunning for the statistics kernels (None: nothing to update)."""
    upd = running_mean is not None and momentum is not None
    if not upd and nbt is None:
        return None
    r = BnRunning()
    r.running_mean = running_mean.data_ptr() if upd else None
    r.running_var = running_var.data_ptr() if upd else None
    r.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
    r.momentum = float(momentum) if upd else 0.0
    return r


def _rp(run):
    return C.byref(run) if run is not None else None


FANIN_RED_USED = [0]     # BatchNorm backwards that took their sums from a fan-in epilogue (tests / bench)
FANIN_BN_SUMS = os.environ.get("DCFP_FANIN_BN_SUMS", "1") not in ("0",)   # =0: every BatchNorm backward runs its own reduce kernel (A/B)
# =2: the fan-in epilogue reduces bn3's gradient even where the fused BatchNorm backward applies (round-3 behaviour).  By
# default (1) it does so only under SyncBN: the fused backward reads dy and x once anyway (12 B/element, what the
# apply-only pass behind the epilogue sums costs), so the epilogue's extra reads would buy nothing there
FANIN_BN_SUMS_ALWAYS = os.environ.get("DCFP_FANIN_BN_SUMS", "1") == "2"
KEEP_XFORM = os.environ.get("DCFP_KEEP_XFORM", "1") not in ("0",)   # =0: weight gradients transform x again (A/B)


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, dil=1, want_stats=False, bn_run=None, out=None, keep=None):
    """y = conv2d(x, w).  With want_stats the result is (y, stats): stats = (mean, biased var) of y
    per output channel over (N, H, W) - what the BatchNorm that follows needs - taken from partials
    the conv epilogue emits, or None where the library has no fused statistics for the shape; bn_run
    (a BnRunning) lets the kernel that finalises them also update that BatchNorm's running statistics.
    out: optional [N, Cout, Ho, Wo] destination, may be a channel slice of a wider tensor (ASPP concat)."""
    _require(x, "x"); _require(w, "weight")
    xp = _pitch_of(x)
    if not xp:
        x = x.contiguous()
    w = w if w.is_contiguous() else w.contiguous()
    d = _desc(x.shape, w.shape, stride, pad, dil, xp, 0)
    yns = 0
    if out is None:
        y = torch.empty((d.N, d.Cout, d.Hout, d.Wout), dtype=torch.float32, device=x.device)
    else:
        y = out
        st = y.stride()
        if tuple(y.shape) != (d.N, d.Cout, d.Hout, d.Wout) or st[3] != 1 or st[2] != d.Wout or st[1] != d.Hout * d.Wout:
            raise RuntimeError("conv2d_fwd: out must be [N,Cout,Ho,Wo] with dense images (a channel slice is fine)")
        yns = st[0]
    if bias is not None:
        _require(bias, "bias"); bias = bias.contiguous()
    L = _lib.lib()
    # (a 3x3 conv WITH bias never runs as Winograd: where the same descriptor without bias does, the kept buffer holds
    #  transformed filters, not the direct kernel's permuted copy - the biased call gets a copy of its own)
    variant = "bias" if (bias is not None and d.KH == 3 and conv_kernel_name(d, _lib.CONV_FWD).startswith("winograd")) else ""
    ws, valid = _conv_workspace(w, _lib.CONV_FWD, d, variant)
    if keep is not None and KEEP_XFORM and bias is None:
        # Winograd conv whose weight gradient is Winograd too: leave the transformed input in a buffer that `keep`
        # (a dict living in the autograd context) holds until the backward pass (conv2d_wgrad(..., xform=))
        xb = L.dcfp_conv2d_xform_bytes(C.byref(d))
        if xb > 0:
            xf = torch.empty(xb, dtype=torch.uint8, device=x.device)
            slots = L.dcfp_conv2d_fwd_stat_slots(C.byref(d), _p(y), yns) if want_stats else 0
            part = _workspace("bn_stat_partials", slots * d.Cout * 2 * 4, x.device) if slots > 0 else None
            _timed("conv_fwd", d, _conv_flops(d), lambda: check(
                L.dcfp_conv2d_fwd_keep_f32_nchw(C.byref(d), _p(x), _p(w), _p(y), yns, _p(xf), xb, _p(part), _p(ws),
                                                ws.numel(), _stream()), "conv2d_fwd_keep"))
            keep["xform"] = xf
            if part is not None:
                mv = torch.empty((2, d.Cout), dtype=torch.float32, device=x.device)
                check(L.dcfp_bn_stats_from_partials_f32(_p(part), slots, 128, d.Cout, _p(mv[0]), _p(mv[1]),
                                                        _rp(bn_run), _stream()), "bn_stats_from_partials")
                return y, (mv[0], mv[1], bn_run is not None)
            return (y, None) if want_stats else y
    if want_stats and bias is None:
        slots = L.dcfp_conv2d_fwd_stat_slots(C.byref(d), _p(y), yns)
        if slots > 0:
            part = _workspace("bn_stat_partials", slots * d.Cout * 2 * 4, x.device)
            _timed("conv_fwd", d, _conv_flops(d), lambda: check(
                L.dcfp_conv2d_fwd_stats_f32_nchw(C.byref(d), _p(x), _p(w), _p(y), yns, _p(part), _p(ws),
                                                 ws.numel(), valid, _stream()), "conv2d_fwd_stats"))
            mv = torch.empty((2, d.Cout), dtype=torch.float32, device=x.device)
            check(L.dcfp_bn_stats_from_partials_f32(_p(part), slots, 128, d.Cout, _p(mv[0]), _p(mv[1]),
                                                    _rp(bn_run), _stream()), "bn_stats_from_partials")
            return y, (mv[0], mv[1], bn_run is not None)
    _timed("conv_fwd", d, _conv_flops(d), lambda: check(
        L.dcfp_conv2d_fwd_f32_nchw(C.byref(d), _p(x), _p(w), _p(bias), _p(y), yns, _p(ws), ws.numel(), valid,
                                   _stream()), "conv2d_fwd"))
    return (y, None) if want_stats else y


def conv2d_dgrad(dy, w, xshape, stride, pad, dil, out=None, accumulate=False):
    """dx = conv_transpose(dy, w); with accumulate, `out` (+)= the result in place."""
    _require(dy, "dy")
    w = w if w.is_contiguous() else w.contiguous()
    dp = _pitch_of(dy)
    d = _desc(xshape, w.shape, stride, pad, dil, 0, dp)
    if dp:
        ns = dy.stride(0)
    else:
        dy, ns = _batch_strided(dy)
    if out is not None:
        if tuple(out.shape) != tuple(xshape) or not out.is_contiguous():
            raise RuntimeError("conv2d_dgrad: out must be a contiguous tensor of the input shape")
        dx = out
    else:
        dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
        accumulate = False
    L = _lib.lib()
    ws, valid = _conv_workspace(w, _lib.CONV_DGRAD, d)
    _timed("conv_dgrad", d, _conv_flops(d), lambda: check(
        L.dcfp_conv2d_dgrad_f32_nchw(C.byref(d), _p(dy), ns, _p(w), _p(dx), int(bool(accumulate)),
                                     _p(ws), ws.numel(), valid, _stream()), "conv2d_dgrad"))
    return dx


MASKED_FANIN = os.environ.get("DCFP_MASKED_FANIN", "1") not in ("0",)   # =0: the residual gradient is written and re-read (A/B)


def conv2d_dgrad_fanin_ok(dyshape, w, xshape):
    """True where conv2d_dgrad_fanin is available for this 1x1 stride-1 conv."""
    d = _desc(xshape, w.shape, 1, 0, 1)
    return MASKED_FANIN and bool(_lib.lib().dcfp_conv2d_dgrad_fanin_supported(C.byref(d)))


def conv2d_dgrad_fanin(dy, w, xshape, fan_src, fan_mask):
    """dx = dgrad(dy, w) + fan_src * mask (1-bit ReLU mask of bn_apply_relu_mask): the gradient fan-in of a residual
    block without the residual branch's gradient ever being written."""
    _require(dy, "dy"); _require(fan_src, "fan_src")
    w = w if w.is_contiguous() else w.contiguous()
    d = _desc(xshape, w.shape, 1, 0, 1)
    dy, ns = _batch_strided(dy)
    if tuple(fan_src.shape) != tuple(xshape) or not fan_src.is_contiguous():
        raise RuntimeError("conv2d_dgrad_fanin: fan_src must be a contiguous tensor of the input shape")
    dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
    ws, valid = _conv_workspace(w, _lib.CONV_DGRAD, d)
    _timed("conv_dgrad", d, _conv_flops(d), lambda: check(
        _lib.lib().dcfp_conv2d_dgrad_fanin_f32_nchw(C.byref(d), _p(dy), ns, _p(w), _p(dx), _p(fan_src), _p(fan_mask),
                                                    _p(ws), ws.numel(), valid, _stream()), "conv2d_dgrad_fanin"))
    return dx


def conv2d_dgrad_fanin_red_slots(w, xshape):
    """Slot count of the fan-in form that also emits the previous block's BatchNorm-backward sums (0: not available)."""
    if not (MASKED_FANIN and FANIN_BN_SUMS):
        return 0
    d = _desc(xshape, w.shape, 1, 0, 1)
    return int(_lib.lib().dcfp_conv2d_dgrad_fanin_red_slots(C.byref(d)))


def conv2d_dgrad_fanin_red(dy, w, xshape, fan_src, fan_mask, red_x, red_mask, red_mean, slots):
    """conv2d_dgrad_fanin whose result dx is ALSO reduced, in the epilogue, to the BatchNorm-backward partial sums of the
    residual block that produced this block's input: part[slot][C][2] = (sum g, sum g*(red_x - red_mean)), g = dx * the
    bit of red_mask.  Returns (dx, part); bn_bwd_sums_from_partials finishes them."""
    _require(dy, "dy"); _require(fan_src, "fan_src"); _require(red_x, "red_x")
    w = w if w.is_contiguous() else w.contiguous()
    d = _desc(xshape, w.shape, 1, 0, 1)
    dy, ns = _batch_strided(dy)
    if tuple(fan_src.shape) != tuple(xshape) or not fan_src.is_contiguous():
        raise RuntimeError("conv2d_dgrad_fanin_red: fan_src must be a contiguous tensor of the input shape")
    if tuple(red_x.shape) != tuple(xshape) or not red_x.is_contiguous() or red_mean.numel() != xshape[1]:
        raise RuntimeError("conv2d_dgrad_fanin_red: red_x / red_mean must match the input shape")
    dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
    part = torch.empty((slots, xshape[1], 2), dtype=torch.float32, device=dy.device)
    ws, valid = _conv_workspace(w, _lib.CONV_DGRAD, d)
    # (its own profile kind: the launch also does the previous block's BatchNorm reduction - bench.py lists it apart)
    _timed("conv_dgrad_red", d, _conv_flops(d), lambda: check(
        _lib.lib().dcfp_conv2d_dgrad_fanin_red_f32_nchw(C.byref(d), _p(dy), ns, _p(w), _p(dx), _p(fan_src), _p(fan_mask),
                                                        _p(red_x), _p(red_mask), _p(red_mean), _p(part), _p(ws),
                                                        ws.numel(), valid, _stream()), "conv2d_dgrad_fanin_red"))
    return dx, part


def bn_bwd_sums_from_partials(part, var, eps, dgamma=None, dbeta=None):
    """(sum_dy, sum_dy_xmu, dgamma) as bn_bwd_reduce returns them, from the partial sums of conv2d_dgrad_fanin_red."""
    slots, Cc, _ = part.shape
    s = torch.empty((2, Cc), dtype=torch.float32, device=part.device)
    if dgamma is None:
        dgamma = torch.empty(Cc, dtype=torch.float32, device=part.device)
    _timed("bn_bwd_reduce", None, 8.0 * part.numel() / 2, lambda: check(
        _lib.lib().dcfp_bn_bwd_sums_from_partials_f32(_p(part), slots, Cc, _p(var), float(eps), _p(s[0]), _p(s[1]),
                                                      _p(dgamma), _p(dbeta), _stream()), "bn_bwd_sums_from_partials"))
    return s[0], s[1], dgamma


def conv2d_wgrad(dy, x, wshape, stride, pad, dil, need_bias=False, dw=None, db=None, xform=None):
    """(dw, db); dw / db: optional destinations (the gradient arena's views).  xform: the transformed input the
    forward call left behind (conv2d_fwd(..., keep=)), taken instead of x where the library says so."""
    _require(dy, "dy"); _require(x, "x")
    xp, dp = _pitch_of(x), _pitch_of(dy)
    if not xp:
        x = x.contiguous()
    d = _desc(x.shape, wshape, stride, pad, dil, xp, dp)
    if dp:
        ns = dy.stride(0)
    else:
        dy, ns = _batch_strided(dy)
    L = _lib.lib()
    nbytes = L.dcfp_conv2d_workspace_bytes(C.byref(d), _lib.CONV_WGRAD)
    ws = _workspace("conv_scratch", nbytes, x.device)     # (shared with the forward / dgrad scratch: same stream)
    if dw is None:
        dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
    elif tuple(dw.shape) != tuple(wshape) or not dw.is_contiguous():
        raise RuntimeError("conv2d_wgrad: dw must be a contiguous tensor of the weight shape")
    if need_bias and db is None:
        db = torch.empty((wshape[0],), dtype=torch.float32, device=x.device)
    if xform is not None and not need_bias and L.dcfp_conv2d_xform_bytes(C.byref(d)) == xform.numel():
        _timed("conv_wgrad", d, _conv_flops(d), lambda: check(
            L.dcfp_conv2d_wgrad_kept_f32_nchw(C.byref(d), _p(dy), ns, _p(xform), xform.numel(), _p(dw), _p(ws), ws.numel(),
                                              _stream()), "conv2d_wgrad_kept"))
        return dw, None
    _timed("conv_wgrad", d, _conv_flops(d), lambda: check(
        L.dcfp_conv2d_wgrad_f32_nchw(C.byref(d), _p(dy), ns, _p(x), _p(dw), _p(db) if need_bias else None, _p(ws),
                                     ws.numel(), _stream()), "conv2d_wgrad"))
    return dw, (db if need_bias else None)


def add_into(dst, src):
    """dst += src through the library (gradient accumulation onto an attached arena view)."""
    check(_lib.lib().dcfp_add_f32(_p(dst), _p(src), _p(dst), dst.numel(), _stream()), "add")


def wgrad_into_param(dy, x, w, bias, stride, pad, dil, keep=None):
    """Weight (and bias) gradient of a conv written where the parameter's gradient lives: returns
    (dw, db) as the autograd Function should return them (None = already accumulated in .grad).
    keep: the dict the forward call filled (conv2d_fwd(..., keep=)); its buffer is released here."""
    tw, kw = arena.grad_target(w)
    tb, kb = arena.grad_target(bias) if bias is not None else (None, 0)
    xform = keep.pop("xform", None) if keep else None
    conv2d_wgrad(dy, x, tuple(w.shape), stride, pad, dil, need_bias=bias is not None, dw=tw, db=tb, xform=xform)
    dw = arena.grad_commit(w, tw, kw, add_into)
    db = arena.grad_commit(bias, tb, kb, add_into) if bias is not None else None
    return dw, db


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d forward/backward (reference: every conv in networks/backbone/resnet.py,
    networks/tools/aspp.py, networks/deeplabv3.py)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil):
        ctx.keep = {} if ctx.needs_input_grad[1] else None     # (a backward pass will ask for the weight gradient)
        y = conv2d_fwd(x, w, bias, stride, pad, dil, keep=ctx.keep)
        ctx.save_for_backward(x)
        ctx.params = (w, bias)           # the Parameter objects: their gradient slots are looked up on them
        ctx.cfg = (stride, pad, dil)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, bias = ctx.params
        stride, pad, dil = ctx.cfg
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(dy, w, tuple(x.shape), stride, pad, dil)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            dw, db = wgrad_into_param(dy, x, w, bias, stride, pad, dil, keep=ctx.keep)
        return dx, dw, db, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1):
    return Conv2dFn.apply(x, w, bias, stride, pad, dil)


# ------------------------------------------------------------- batch norm
def bn_stats(x, run=None):
    _require(x, "x")
    N, Cc, H, W = x.shape
    x, ns = _batch_strided(x)
    L = _lib.lib()
    ws = _workspace("bn", L.dcfp_bn_workspace_bytes(N, Cc, H * W), x.device)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    var = torch.empty(Cc, dtype=torch.float32, device=x.device)
    _timed("bn_stats", None, 4.0 * x.numel(), lambda: check(
        L.dcfp_bn_stats_f32(_p(x), ns, N, Cc, H * W, _p(mean), _p(var), _rp(run), _p(ws), ws.numel(), _stream()),
        "bn_stats"))
    return mean, var


def bn_apply(x, mean, var, gamma, beta, eps, residual=None, relu=False, out=None):
    N, Cc, H, W = x.shape
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    yns, ypitch = 0, 0
    if out is None:
        y = torch.empty_like(x)
    else:           # a channel slice of the ASPP concat tensor (aspp.py:77), or a row-pitched buffer
        y = out
        st = y.stride()
        ypitch = _pitch_of(y)
        if tuple(y.shape) != tuple(x.shape) or (not ypitch and (st[3] != 1 or st[2] != W or st[1] != H * W)):
            raise RuntimeError("bn_apply: out must have dense images or be a row-pitched buffer")
        yns = st[0]
    nbytes = (8.0 + (4.0 if residual is not None else 0.0)) * x.numel()
    _timed("bn_apply", None, nbytes, lambda: check(
        _lib.lib().dcfp_bn_apply_f32(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), float(eps),
                                     _p(residual), int(relu), _p(y), yns, N, Cc, H * W, W, ypitch, _stream()),
        "bn_apply"))
    return y


def bn_apply_relu_mask(x, mean, var, gamma, beta, eps, residual):
    """y = relu(BN(x) + residual) plus the ReLU mask as one bit per element (int64 words), or None where
    the library has no mask path for the shape (HW % 256 != 0)."""
    N, Cc, H, W = x.shape
    if (H * W) % 256 != 0:
        return None
    x = x.contiguous(); residual = residual.contiguous()
    y = torch.empty_like(x)
    mask = torch.empty(x.numel() // 64, dtype=torch.int64, device=x.device)
    rc = [0]

    def run():
        rc[0] = _lib.lib().dcfp_bn_apply_relu_mask_f32(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), float(eps),
                                                       _p(residual), _p(y), _p(mask), N, Cc, H * W, _stream())
    _timed("bn_apply", None, 12.125 * x.numel(), run)
    if rc[0] == _lib.E_UNSUPPORTED:
        return None
    check(rc[0], "bn_apply_relu_mask")
    return y, mask


def bn_bwd_reduce(dy, x, y, mean, var, gamma, beta, eps, relu, dgamma=None, dbeta=None):
    """relu: 0 none, 1 mask from y, 2 mask re-derived from x (forward had no residual), 3 `y` is the
    1-bit-per-element mask written by bn_apply_relu_mask.  Returns (sum_dy, sum_dy_xmu) adjacent in one
    buffer, and dgamma; dgamma / dbeta (optional C-float destinations: the parameters' gradient slots)
    receive sum_dy_xmu * istd and a copy of sum_dy."""
    N, Cc, H, W = x.shape
    dy, dns = _batch_strided(dy)
    L = _lib.lib()
    ws = _workspace("bn", L.dcfp_bn_workspace_bytes(N, Cc, H * W), x.device)
    s = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    s1, s2 = s[0], s[1]
    if dgamma is None:
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    _timed("bn_bwd_reduce", None, (8.0 + (4.0 if relu == 1 else 0.125 if relu == 3 else 0.0)) * x.numel(), lambda: check(
        L.dcfp_bn_bwd_reduce_f32(_p(dy), dns, _p(x), _p(y) if relu in (1, 3) else None, 0, _p(mean), _p(var),
                                 _p(gamma), _p(beta), float(eps), int(relu), N, Cc, H * W, _p(s1), _p(s2),
                                 _p(dgamma), _p(dbeta), _p(ws), ws.numel(), _stream()),
        "bn_bwd_reduce"))
    return s1, s2, dgamma


def bn_bwd_apply(dy, x, y, mean, var, gamma, beta, eps, s1, s2, count, relu, want_residual, dx_out=None):
    """dx_out: optional row-pitched destination (pitched_buffer) for dx."""
    N, Cc, H, W = x.shape
    dy, dns = _batch_strided(dy)
    dxp = 0
    if dx_out is not None:
        dxp = _pitch_of(dx_out)
        if tuple(dx_out.shape) != tuple(x.shape) or not dxp:
            raise RuntimeError("bn_bwd_apply: dx_out must be a row-pitched buffer of x's shape")
        dx = dx_out
    else:
        dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual else None
    nbytes = (12.0 + (4.0 if relu == 1 else 0.125 if relu == 3 else 0.0) + (4.0 if want_residual else 0.0)) * x.numel()
    count_dev = count if isinstance(count, torch.Tensor) else None   # SyncBN: global count on device
    count_host = 0.0 if count_dev is not None else float(count)
    _timed("bn_bwd_apply", None, nbytes, lambda: check(
        _lib.lib().dcfp_bn_bwd_apply_f32(_p(dy), dns, _p(x), _p(y) if relu in (1, 3) else None, 0, _p(mean),
                                         _p(var), _p(gamma), _p(beta), float(eps), _p(s1), _p(s2), count_host,
                                         _p(count_dev), int(relu), _p(dx), _p(dres), N, Cc, H * W, W, dxp,
                                         _stream()),
        "bn_bwd_apply"))
    return dx, dres


# ---- the fused BatchNorm backward (dcfp_bn_bwd_fused_f32): both stages in one launch, dy and x read once.
# DCFP_BN_FUSED=0 keeps the two-kernel path (which the data-parallel path always takes: its sums cross the ranks).
BN_BWD_FUSED = os.environ.get("DCFP_BN_FUSED", "1") != "0"
FUSED_BN_USED = [0]            # launches since import (tests read it)
_FUSED_SYNC = {}               # (device index, stream) -> [hand-off buffer, last epoch, status word]


def _fused_sync(nbytes, device):
    """The hand-off buffer of the fused kernel's blocks, its next epoch and the give-up status word: zero-filled when
    created or grown, epochs strictly increasing per buffer (the contract of dcfp_bn_bwd_fused_f32)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    st = _FUSED_SYNC.get(key)
    if st is None:
        st = [None, 0, torch.zeros(1, dtype=torch.int32, device=device)]
        _FUSED_SYNC[key] = st
    if st[0] is None or st[0].numel() < nbytes or st[1] >= 0xfffffff0:
        st[0] = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        st[1] = 0
    st[1] += 1
    return st[0], st[1], st[2]


def check_fused_status():
    """Once per step, next to the step's own host synchronisation: raise if a block of a fused BatchNorm backward gave
    up waiting for the other blocks of its channel (its outputs were poisoned with NaN)."""
    for (dev, _), st in _FUSED_SYNC.items():
        v = int(st[2].item())
        if v != 0:
            raise RuntimeError("dcfp_amd: fused BatchNorm backward (call %d on device %s) gave up waiting for the blocks of "
                               "a channel; its outputs are NaN.  DCFP_BN_FUSED=0 selects the two-kernel path" % (v, dev))


def bn_bwd_fused(dy, x, y, mean, var, gamma, beta, eps, count, relu, want_residual, dx_out=None, dgamma=None, dbeta=None):
    """bn_bwd_reduce + bn_bwd_apply in one launch (same bits).  Returns (s1, s2, dgamma, dx, dres), or None when the
    shape is not supported by the fused kernel (the caller then takes the two-kernel path)."""
    N, Cc, H, W = x.shape
    L = _lib.lib()
    need = L.dcfp_bn_bwd_fused_sync_bytes(N, Cc, H * W)
    if need == 0:
        return None
    dy, dns = _batch_strided(dy)
    dxp = 0
    if dx_out is not None:
        dxp = _pitch_of(dx_out)
        if tuple(dx_out.shape) != tuple(x.shape) or not dxp:
            raise RuntimeError("bn_bwd_fused: dx_out must be a row-pitched buffer of x's shape")
        dx = dx_out
    else:
        dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual else None
    s = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    if dgamma is None:
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    sync, epoch, status = _fused_sync(need, x.device)
    nbytes = (12.0 + (4.0 if relu == 1 else 0.125 if relu == 3 else 0.0) + (4.0 if want_residual else 0.0)) * x.numel()
    rc = [0]

    def run():
        rc[0] = L.dcfp_bn_bwd_fused_f32(_p(dy), dns, _p(x), _p(y) if relu in (1, 3) else None, 0, _p(mean), _p(var),
                                        _p(gamma), _p(beta), float(eps), float(count), int(relu), _p(dx), _p(dres),
                                        N, Cc, H * W, W, dxp, _p(s[0]), _p(s[1]), _p(dgamma), _p(dbeta), _p(sync),
                                        sync.numel(), epoch, 0, _p(status), _stream())
    _timed("bn_bwd_fused", None, nbytes, run)
    if rc[0] == _lib.E_UNSUPPORTED:
        return None
    check(rc[0], "bn_bwd_fused")
    FUSED_BN_USED[0] += 1
    return s[0], s[1], dgamma, dx, dres


def _sync_group(group):
    if group is False or not dist.is_available() or not dist.is_initialized():
        return None
    g = None if group is True else group
    if dist.get_world_size(g) <= 1 and not os.environ.get("DCFP_FORCE_SYNCBN"):
        return None   # (the env switch rehearses the exchange at world size 1)
    return g if g is not None else dist.group.WORLD


_COUNT_CACHE = {}


def _count_tensor(count, device):
    """Per-rank pixel count as a one-float device tensor, uploaded once per distinct value (a fresh
    torch.tensor(..., device=cuda) per BN layer is a blocking H2D copy: 115 pipeline drains per step)."""
    key = (float(count), str(device))
    t = _COUNT_CACHE.get(key)
    if t is None:
        t = torch.tensor([float(count)], device=device, dtype=torch.float32)
        _COUNT_CACHE[key] = t
    return t


def syncbn_combine_reference(allv, Cc):
    """The pooled statistics of `world` gathered rows (mean[C], var[C], count) - the ONE formula both the
    HIP kernel (syncbn_combine_kernel in csrc/bn.hip) and the host path below evaluate, in fp64 and in
    rank order:  tot = sum n_r;  m = (sum mean_r n_r)/tot;  v = (sum (var_r + (mean_r - m)^2) n_r)/tot."""
    a = allv.double()
    world = a.shape[0]
    n = a[:, 2 * Cc]
    tot = torch.zeros((), dtype=torch.float64, device=a.device)
    m = torch.zeros(Cc, dtype=torch.float64, device=a.device)
    for r in range(world):
        tot = tot + n[r]
        m = m + a[r, :Cc] * n[r]
    m = m / tot
    v = torch.zeros(Cc, dtype=torch.float64, device=a.device)
    for r in range(world):
        dlt = a[r, :Cc] - m
        v = v + (a[r, Cc:2 * Cc] + dlt * dlt) * n[r]
    return m.float(), (v / tot).float(), tot.float().reshape(1)


def sync_bn_stats(mean, var, count, group, run=None, running=None):
    """SyncBatchNorm forward exchange (engine.py:65): ONE all_gather of [mean, var, count]
    (2C+1 floats per rank), then the pooled mean / biased variance over all ranks' pixels
    (parallel-variance combination), identical on every rank; the total count stays on the device.
    run (device path) / running = (running_mean, running_var, momentum, nbt) (host path): the running
    statistics are updated from the POOLED statistics, as nn.SyncBatchNorm does."""
    Cc = mean.numel()
    world = dist.get_world_size(group)
    local = torch.cat([mean, var, _count_tensor(count, mean.device)])
    px = syncbn_p2p.for_group(group) if mean.is_cuda else None
    if px is not None:       # one launch: peer-to-peer gather + the same rank-order fp64 combination
        out = torch.empty(2 * Cc + 1, device=mean.device, dtype=torch.float32)
        _timed("syncbn_p2p_fwd", None, 4.0 * local.numel() * world, lambda: px.exchange(local, out, 2, _rp(run)))
        return out[:Cc], out[Cc:2 * Cc], out[2 * Cc:]
    allv = torch.empty(world, local.numel(), device=mean.device, dtype=mean.dtype)
    # (exposed on the compute stream: the normalisation that follows needs the pooled statistics; bench.py sums these)
    if mean.is_cuda:
        _timed("syncbn_allgather", None, 4.0 * local.numel() * world,
               lambda: dist.all_gather_into_tensor(allv, local.unsqueeze(0), group=group))
    else:
        dist.all_gather_into_tensor(allv, local.unsqueeze(0), group=group)
    if mean.is_cuda:
        out = torch.empty(2 * Cc + 1, device=mean.device, dtype=torch.float32)
        check(_lib.lib().dcfp_syncbn_combine_f32(_p(allv), world, Cc, _p(out), _p(out[Cc:]), _p(out[2 * Cc:]),
                                                 _rp(run), _stream()), "syncbn_combine")
        return out[:Cc], out[Cc:2 * Cc], out[2 * Cc:]
    # host tensors (the gloo tests): the same formula, evaluated by torch in fp64
    gmean, gvar, total = syncbn_combine_reference(allv, Cc)
    if running is not None:
        rm, rv, momentum, nbt = running
        if rm is not None and momentum is not None:
            n = float(total)
            rm.mul_(1.0 - momentum).add_(gmean, alpha=momentum)
            rv.mul_(1.0 - momentum).add_(gvar * (n / max(n - 1.0, 1.0)), alpha=momentum)
        if nbt is not None:
            nbt.add_(1)
    return gmean.contiguous(), gvar.contiguous(), total.contiguous()


def sync_bn_bwd_sums(s1, s2, group, async_op=False):
    """SyncBatchNorm backward exchange: ONE all_reduce(SUM) of [sum g, sum g*(x-mean)] (2C floats);
    bn_bwd_reduce returns the two rows adjacent in one buffer, which is then reduced in place.
    async_op: returns (s1, s2, work) with the exchange in flight on the collective's stream - the caller
    enqueues independent kernels (a weight gradient) and calls work.wait() before it reads the sums."""
    Cc = s1.numel()
    adjacent = (s1.is_contiguous() and s2.is_contiguous()
                and s1.untyped_storage().data_ptr() == s2.untyped_storage().data_ptr()
                and s2.storage_offset() == s1.storage_offset() + Cc)
    px = syncbn_p2p.for_group(group) if s1.is_cuda else None
    if px is not None:       # one launch: peer-to-peer gather + sum in rank order (identical on every rank)
        both = s1.as_strided((2 * Cc,), (1,), s1.storage_offset()) if adjacent else torch.cat([s1, s2])
        out = torch.empty_like(both)
        if async_op:
            work = px.exchange_async(both, out, 1)
            return out[:Cc], out[Cc:], work
        _timed("syncbn_p2p_bwd", None, 8.0 * Cc * px.world, lambda: px.exchange(both, out, 1))
        return out[:Cc], out[Cc:]
    if adjacent:
        both = s1.as_strided((2 * Cc,), (1,), s1.storage_offset())
        work = dist.all_reduce(both, group=group, async_op=async_op)
        return (s1, s2, work) if async_op else (s1, s2)
    both = torch.cat([s1, s2])
    work = dist.all_reduce(both, group=group, async_op=async_op)
    r1, r2 = both[:Cc], both[Cc:]
    return (r1, r2, work) if async_op else (r1.contiguous(), r2.contiguous())


def bn_forward_impl(x, gamma, beta, running_mean, running_var, residual, relu, training,
                    momentum, eps, sync, nbt=None, stats=None, want_mask=False, out=None):
    """Shared BN(+ReLU)(+residual) forward: returns (y, state) with state =
    (mean, var, count, group) for the backward.  `stats` = this rank's (mean, biased var, running-updated?)
    when the producing conv already emitted them (conv2d_fwd(..., want_stats=True)).  nbt: the module's
    num_batches_tracked buffer (incremented by the kernel that finalises the statistics)."""
    _require(x, "x")
    x = x.contiguous()
    N, Cc, H, W = x.shape
    count = float(N * H * W)
    group = _sync_group(sync) if training else None
    if training:
        run = _bn_run(running_mean, running_var, momentum, nbt)
        if stats is not None:
            mean, var, applied = stats
            if applied:
                run = None
        elif group is None:
            mean, var = bn_stats(x, run)
            run = None
        else:
            mean, var = bn_stats(x)
        if group is not None:
            mean, var, count = sync_bn_stats(mean, var, count, group, run)
        elif run is not None:     # statistics came from elsewhere without the bookkeeping
            if running_mean is not None and momentum is not None:
                check(_lib.lib().dcfp_bn_update_running_f32(
                    _p(mean), _p(var), Cc, float(momentum), float(count), None, _p(running_mean),
                    _p(running_var), _stream()), "bn_update_running")
            if nbt is not None:
                nbt.add_(1)
    else:
        mean, var = running_mean, running_var
    if want_mask and relu and residual is not None and BN_RELU_BITMASK and out is None:
        ym = bn_apply_relu_mask(x, mean, var, gamma, beta, eps, residual)
        if ym is not None:
            return ym[0], (mean, var, count, group, ym[1])
    y = bn_apply(x, mean, var, gamma, beta, eps, residual, relu, out=out)
    return y, (mean, var, count, group)


def bn_backward_reduce(dy, x, y, gamma, beta, state, relu, training, gparam=None, bparam=None, eps=1e-5, pre=None):
    """First half of the BN backward: the two per-channel sums, dgamma / dbeta written to the parameters'
    gradient slots, and (SyncBN) the exchange of the sums started asynchronously.  Returns a tuple for
    bn_backward_apply; independent kernels enqueued between the two overlap the exchange."""
    mean, var, count, group = state[:4]
    mask = state[4] if len(state) > 4 else None
    if relu and mask is not None:      # residual BN whose forward kept the ReLU mask as bits
        relu, y = 3, mask
    else:
        relu = (1 if y is not None else 2) if relu else 0
    gp = gparam if gparam is not None else gamma
    bp = bparam if bparam is not None else beta
    tg, kg = arena.grad_target(gp)
    tb, kb = arena.grad_target(bp)
    if pre is not None:     # the producer of dy already reduced it per 128 pixels (conv2d_dgrad_fanin_red): finish the sums
        s1, s2, _ = bn_bwd_sums_from_partials(pre, var, eps, dgamma=tg, dbeta=tb)
    else:
        s1, s2, _ = bn_bwd_reduce(dy, x, y, mean, var, gamma, beta, eps, relu, dgamma=tg, dbeta=tb)
    dgamma = arena.grad_commit(gp, tg, kg, add_into)
    dbeta = arena.grad_commit(bp, tb, kb, add_into)
    work = None
    if training and group is not None:
        s1, s2, work = sync_bn_bwd_sums(s1, s2, group, async_op=SYNCBN_ASYNC)[:3] if SYNCBN_ASYNC else \
            sync_bn_bwd_sums(s1, s2, group) + (None,)
    elif not training:   # running statistics are constants: dx = g * gamma * istd
        s1 = torch.zeros_like(s1); s2 = torch.zeros_like(s2)
    return (s1, s2, work, relu, y, dgamma, dbeta)


def bn_backward_apply(dy, x, gamma, beta, state, red, eps, want_res, dx_out=None):
    s1, s2, work, relu, y, dgamma, dbeta = red
    if work is not None:
        # (bench.py sums these: how long the compute stream stands still for a backward SyncBN exchange that the weight
        #  gradient enqueued in between did not hide - comm.syncbn_bwd_exposed_ms)
        _timed("syncbn_bwd_wait", None, 0.0, work.wait)
    mean, var, count = state[:3]
    dx, dres = bn_bwd_apply(dy, x, y, mean, var, gamma, beta, eps, s1, s2, count, relu, want_res, dx_out)
    return dx, dgamma, dbeta, dres


def bn_backward_impl(dy, x, y, gamma, beta, state, relu, training, eps, want_res, gparam=None, bparam=None,
                     between=None, dx_out=None, pre=None):
    """Shared backward: returns (dx, dgamma, dbeta, dres).  dgamma/dbeta are this rank's sums (None when
    they went straight into the gradient arena; the gradient all-reduce averages them); under SyncBN the
    sums entering dx are global.  `y` is only needed for the ReLU mask of a BN that had a residual input;
    otherwise the mask is re-derived from x inside the kernels (pass y=None).  between: optional callable
    run after the reduction (and the start of the SyncBN exchange) and before dx - independent work
    that hides the exchange."""
    if BN_BWD_FUSED and training and pre is None and state[3] is None and not isinstance(state[2], torch.Tensor):
        # no exchange between the two stages: one launch, dy and x read once
        mask = state[4] if len(state) > 4 else None
        if relu and mask is not None:
            frelu, fy = 3, mask
        else:
            frelu, fy = ((1 if y is not None else 2) if relu else 0), y
        gp = gparam if gparam is not None else gamma
        bp = bparam if bparam is not None else beta
        tg, kg = arena.grad_target(gp)
        tb, kb = arena.grad_target(bp)
        out = bn_bwd_fused(dy, x, fy, state[0], state[1], gamma, beta, eps, state[2], frelu, want_res, dx_out,
                           dgamma=tg, dbeta=tb)
        if out is not None:
            dgamma = arena.grad_commit(gp, tg, kg, add_into)
            dbeta = arena.grad_commit(bp, tb, kb, add_into)
            mid = between() if between is not None else None
            return (out[3], dgamma, dbeta, out[4]) + ((mid,) if between is not None else ())
    red = bn_backward_reduce(dy, x, y, gamma, beta, state, relu, training, gparam, bparam, eps, pre=pre)
    mid = between() if between is not None else None
    out = bn_backward_apply(dy, x, gamma, beta, state, red, eps, want_res, dx_out)
    return out + ((mid,) if between is not None else ())


class BatchNormActFn(torch.autograd.Function):
    """y = act(BN(x) [+ residual]) with batch statistics (training) or running statistics
    (eval).  Mirrors nn.BatchNorm2d + nn.ReLU(inplace) (+ the Bottleneck residual add,
    networks/backbone/resnet.py:41-56); `sync` reproduces nn.SyncBatchNorm (engine.py:65):
    statistics pooled over all ranks, gamma/beta gradients left local for the gradient
    all-reduce."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, relu, training,
                momentum, eps, sync, nbt, pitch_cfg):
        x = x.contiguous()
        # pitch_cfg = (owner module, y_pitch, dx_pitch): write y row-pitched for the 3x3 conv that reads it next /
        # write dx row-pitched for the 3x3 conv that produced x (their shifted operands: conv_pitch)
        out, ctx.pitch_slot, ctx.dx_pitch, ctx.dx_key = None, None, 0, "bn_dx"
        if pitch_cfg is not None:
            owner, y_pitch, ctx.dx_pitch = pitch_cfg
            # (the gradient handed to autograd is a view of a persistent buffer: one per owning module, so that a later
            #  BatchNorm backward of the same shape cannot overwrite a gradient autograd still holds)
            ctx.dx_key = ("bn_dx", id(owner))
            if y_pitch and residual is None:
                out, ctx.pitch_slot = owner_pitched(owner, tuple(x.shape), y_pitch, x.device,
                                                    track=torch.is_grad_enabled())
        y, state = bn_forward_impl(x, gamma, beta, running_mean, running_var, residual, relu,
                                   training, momentum, eps, sync, nbt=nbt, out=out)
        mean, var, count, group = state[:4]
        # y is saved only where the ReLU mask cannot be re-derived from x (residual input)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, mean, var,
                              count if isinstance(count, torch.Tensor) else None)
        ctx.params = (gamma, beta)
        ctx.cfg = (relu, training, eps, None if isinstance(count, torch.Tensor) else count, group,
                   residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, count_t = ctx.saved_tensors
        gamma, beta = ctx.params
        relu, training, eps, count, group, has_res = ctx.cfg
        state = (mean, var, count_t if count_t is not None else count, group)
        need_res = has_res and ctx.needs_input_grad[5]
        dx_out = pitched_buffer(tuple(x.shape), ctx.dx_pitch, ctx.dx_key, x.device) if ctx.dx_pitch else None
        dx, dgamma, dbeta, dres = bn_backward_impl(dy, x, y, gamma, beta, state, relu, training, eps, need_res,
                                                   dx_out=dx_out)
        if ctx.pitch_slot is not None:
            ctx.pitch_slot.release()
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None


class BottleneckFn(torch.autograd.Function):
    """One residual Bottleneck (networks/backbone/resnet.py:38-58) as a single autograd node:
    1x1 -> BN,ReLU -> 3x3(dil) -> BN,ReLU -> 1x1 -> BN (+residual) -> ReLU, with an optional
    1x1(stride)+BN downsample on the residual.  Same kernels as the op-level Functions; what the
    fusion buys is the backward: the block input receives two gradients (through conv1 and
    through the residual), and here conv1's dgrad ACCUMULATES into the residual gradient buffer
    instead of autograd adding two full tensors (33 full-size adds per step); every weight / gamma / beta
    gradient is written straight into the gradient arena; under SyncBN each weight gradient is enqueued
    between a BatchNorm's reduction and its dx so that it hides the exchange of the sums."""

    @staticmethod
    def forward(ctx, x, cfg, *tensors):
        # tensors: w1,g1,b1, w2,g2,b2, w3,g3,b3 [, wd,gd,bd]; cfg: dict of python values + BN buffers
        # (the block that produced x, if it left its bn3 record on the tensor: see the end of this function)
        ctx.prev_rec = getattr(x, "_dcfp_bn3_rec", None) if FANIN_BN_SUMS else None
        x = x.contiguous()
        has_ds = len(tensors) == 12
        w1, g1, b1, w2, g2, b2, w3, g3, b3 = tensors[:9]
        bnargs = cfg["bn"]   # per BN: (running_mean, running_var, training, momentum, eps, sync, nbt)
        stride, dil = cfg["stride"], cfg["dil"]
        fuse = FUSE_BN_STATS   # each conv hands the batch statistics of its output to the BatchNorm behind it

        def conv_bn(inp, wgt, a, g, b, st=1, pd=0, dl=1, relu=True, res=None, want_mask=False, out=None, keep=None):
            rm, rv, training, momentum, eps, sync, nbt = a
            stats = None
            if fuse and training:
                # the kernel finalising the statistics also does the running-stat bookkeeping, unless the
                # statistics still have to be pooled over the ranks first (SyncBN)
                run = _bn_run(rm, rv, momentum, nbt) if _sync_group(sync) is None else None
                c, stats = conv2d_fwd(inp, wgt, None, st, pd, dl, want_stats=True, bn_run=run, keep=keep)
            else:
                c = conv2d_fwd(inp, wgt, None, st, pd, dl, keep=keep)
            y, state = bn_forward_impl(c, g, b, rm, rv, res, relu, training, momentum, eps, sync, nbt=nbt,
                                       stats=stats, want_mask=want_mask, out=out)
            return c, y, state

        # conv2 with dilation 1 / 2: its input y1 (and, in backward, the gradient of its output) are kept
        # row-pitched with a zero tail, so the 3x3 kernels copy every shifted quad without border handling.
        # y1 lives until this block's backward: one persistent buffer per block (cfg["owner"]), guarded by a
        # busy flag (a second forward before the backward gets a fresh buffer).
        N_, _, H_, W_ = x.shape
        y1_shape = (N_, w1.shape[0], H_, W_)
        pitch = conv_pitch(y1_shape, tuple(w2.shape), stride, dil, dil)
        y1_out, ctx.pitch_slot = None, None
        if pitch:
            y1_out, ctx.pitch_slot = owner_pitched(cfg.get("owner"), y1_shape, pitch, x.device)
        ctx.pitch = pitch
        c1, y1, st1 = conv_bn(x, w1, bnargs[0], g1, b1, out=y1_out)
        # (conv2's Winograd input transform is kept for its weight gradient: 4x the size of y1, until this block's backward)
        ctx.keep2 = {} if ctx.needs_input_grad[5] else None      # inputs: x, cfg, w1, g1, b1, w2, ...
        c2, y2, st2 = conv_bn(y1, w2, bnargs[1], g2, b2, stride, dil, dil, keep=ctx.keep2)
        if has_ds:
            wd, gd, bd = tensors[9:]
            cd, res, std = conv_bn(x, wd, bnargs[3], gd, bd, stride, 0, 1, relu=False)
        else:
            cd, res, std = None, x, None
        c3, out, st3 = conv_bn(y2, w3, bnargs[2], g3, b3, res=res, want_mask=True)
        ctx.has_ds = has_ds
        ctx.cfg = (stride, dil, [a[2] for a in bnargs], [a[4] for a in bnargs])
        ctx.params = tensors
        # per-BN backward state: tensors go through save_for_backward (version checks), the rest stays python
        flat, meta = [], []
        for st in (st1, st2, st3, std):
            if st is None:
                meta.append(None)
                continue
            mean, var, count, group = st[:4]
            mask = st[4] if len(st) > 4 else None
            meta.append((len(flat), isinstance(count, torch.Tensor), None if isinstance(count, torch.Tensor) else count,
                         group, mask is not None))
            flat += [mean, var] + ([count] if isinstance(count, torch.Tensor) else []) + ([mask] if mask is not None else [])
        ctx.meta = meta
        # with the 1-bit mask the block output is not needed by its own backward
        keep_out = out if len(st3) <= 4 else None
        ctx.save_for_backward(x, c1, y1, c2, y2, c3, keep_out, cd, *flat)
        # What the NEXT block needs to reduce this block's bn3 gradient in its fan-in epilogue (the gradient of `out` is
        # that fan-in's result): bn3's input, the ReLU bit mask and the batch mean.  The record travels on the output
        # tensor; the consumer leaves the partial sums in it and this block's backward picks them up.
        ctx.rec = None
        # (only when a backward pass can follow: a training-mode forward under no_grad builds no graph, and the record on
        #  its output would keep bn3's input alive for nothing)
        fused_bn3 = BN_BWD_FUSED and st3[3] is None and not FANIN_BN_SUMS_ALWAYS     # bn3's backward will be ONE launch
        if FANIN_BN_SUMS and not fused_bn3 and len(st3) > 4 and any(ctx.needs_input_grad):
            ctx.rec = {"c3": c3, "mask": st3[4], "mean": st3[0]}
            out._dcfp_bn3_rec = ctx.rec
        return out

    @staticmethod
    def backward(ctx, dout):
        x, c1, y1, c2, y2, c3, out, cd = ctx.saved_tensors[:8]
        flat = ctx.saved_tensors[8:]
        tensors = ctx.params
        w1, g1, b1, w2, g2, b2, w3, g3, b3 = tensors[:9]
        stride, dil, training, eps = ctx.cfg

        def state(i):
            m = ctx.meta[i]
            if m is None:
                return None
            pos, count_is_t, count, group, has_mask = m
            mean, var = flat[pos], flat[pos + 1]
            pos += 2
            if count_is_t:
                count = flat[pos]; pos += 1
            st = (mean, var, count, group)
            return st + ((flat[pos],) if has_mask else ())
        st1, st2, st3, std = state(0), state(1), state(2), state(3)

        def wg(dy, inp, w, st=1, pd=0, dl=1, keep=None):
            return lambda: wgrad_into_param(dy, inp, w, None, st, pd, dl, keep=keep)[0]

        # bn3 (+residual, ReLU): gradient of conv3's output and of the residual branch.  Identity-shortcut blocks whose
        # forward kept the ReLU mask as bits do not materialise the residual gradient dout * mask: conv1's dgrad adds
        # it from (dout, mask) in its epilogue (conv2d_dgrad_fanin) - 4 B/element less written per block
        mask3 = st3[4] if len(st3) > 4 else None
        fanin = (not ctx.has_ds and mask3 is not None and ctx.needs_input_grad[0] and dout.is_contiguous()
                 and tuple(dout.shape) == tuple(x.shape) and conv2d_dgrad_fanin_ok(None, w1, tuple(x.shape)))
        # bn3's sums may already sit in this block's record, reduced per 128 pixels by the fan-in that produced dout - but
        # only if dout IS that fan-in's result (autograd adds other consumers' gradients into a new tensor, or in place
        # with a version bump: either way the partial sums would be of something else)
        pre, rec = None, getattr(ctx, "rec", None)
        if rec is not None:
            part, ptr, ver = rec.pop("part", None), rec.pop("dx_ptr", None), rec.pop("dx_ver", None)
            if part is not None and dout.data_ptr() == ptr and dout._version == ver and dout.is_contiguous():
                pre = part
                FANIN_RED_USED[0] += 1
            # the record's references must not outlive this backward: the node object (ctx) stays until the whole graph
            # is dropped, and 33 blocks' bn3 inputs held that long are 14 GB of peak memory
            rec.clear()
            ctx.rec = None
        d_c3, dg3, db3, d_res = bn_backward_impl(dout, c3, out, g3, b3, st3, True, training[2], eps[2], not fanin, pre=pre)
        d_y2 = conv2d_dgrad(d_c3, w3, tuple(y2.shape), 1, 0, 1)
        # conv3's weight gradient does not feed bn2: it runs between bn2's reduction and its dx
        dc2_out = pitched_buffer(tuple(c2.shape), ctx.pitch, "d_c2", c2.device) if ctx.pitch else None
        d_c2, dg2, db2, _, dw3 = bn_backward_impl(d_y2, c2, None, g2, b2, st2, True, training[1], eps[1], False,
                                                  between=wg(d_c3, y2, w3), dx_out=dc2_out)
        d_y1 = conv2d_dgrad(d_c2, w2, tuple(y1.shape), stride, dil, dil)
        d_c1, dg1, db1, _, dw2 = bn_backward_impl(d_y1, c1, None, g1, b1, st1, True, training[0], eps[0], False,
                                                  between=wg(d_c2, y1, w2, stride, dil, dil, keep=ctx.keep2))
        grads = [None, dg1, db1, dw2, dg2, db2, dw3, dg3, db3]
        if ctx.has_ds:
            wd, gd, bd = tensors[9:]
            d_cd, dgd, dbd, _, dw1 = bn_backward_impl(d_res, cd, None, gd, bd, std, False, training[3], eps[3], False,
                                                      between=wg(d_c1, x, w1))
            dwd, _ = wgrad_into_param(d_cd, x, wd, None, stride, 0, 1)
            grads += [dwd, dgd, dbd]
            dx = conv2d_dgrad(d_cd, wd, tuple(x.shape), stride, 0, 1) if ctx.needs_input_grad[0] else None
        else:
            dw1, _ = wgrad_into_param(d_c1, x, w1, None, 1, 0, 1)
            dx = d_res
        grads[0] = dw1
        prev = ctx.prev_rec
        slots = 0
        if fanin and prev is not None and tuple(prev["c3"].shape) == tuple(x.shape):
            slots = conv2d_dgrad_fanin_red_slots(w1, tuple(x.shape))
        if fanin and slots > 0:
            # dx is the gradient arriving at the previous block's output: reduce it for that block's bn3 on the way out
            dx, part = conv2d_dgrad_fanin_red(d_c1, w1, tuple(x.shape), dout, mask3, prev["c3"], prev["mask"],
                                              prev["mean"], slots)
            prev["part"], prev["dx_ptr"], prev["dx_ver"] = part, dx.data_ptr(), dx._version
            del part
        elif fanin:
            dx = conv2d_dgrad_fanin(d_c1, w1, tuple(x.shape), dout, mask3)
        elif ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(d_c1, w1, tuple(x.shape), 1, 0, 1, out=dx, accumulate=True)
        else:
            dx = None
        ctx.prev_rec = None
        if ctx.pitch_slot is not None:
            ctx.pitch_slot.release()            # y1's buffer may be reused by the next forward of this block
        return (dx, None) + tuple(grads)


def bottleneck(x, cfg, tensors):
    return BottleneckFn.apply(x, cfg, *tensors)


def batch_norm_act(x, gamma, beta, running_mean, running_var, residual=None, relu=False,
                   training=True, momentum=0.1, eps=1e-5, sync=False, nbt=None, pitch_cfg=None):
    return BatchNormActFn.apply(x, gamma, beta, running_mean, running_var, residual, relu,
                                training, momentum, eps, sync, nbt, pitch_cfg)


class AsppFn(torch.autograd.Function):
    """The five ASPP branches and their concatenation (networks/tools/aspp.py:70-77) as ONE autograd node:
    1x1 / three dilated 3x3 conv -> BN -> ReLU, and global mean -> 1x1 conv -> BN -> ReLU -> broadcast,
    each writing its channel slice of the 1280-channel tensor directly (no torch.cat: the BN kernels take a
    batch stride).  Backward: the branches read their slice of the incoming gradient in place and their
    data gradients ACCUMULATE into one buffer through the dgrad epilogue (the reference's autograd adds
    five 2048-channel tensors: 4 read-read-write passes over 1 GB each); each branch's weight gradient
    runs between the next branch's BN reduction and its dx (hides the SyncBN exchange)."""

    @staticmethod
    def forward(ctx, x, cfg, *tensors):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        widths = [t.shape[0] for t in tensors[0::3]]
        cat = torch.empty((N, sum(widths), H, W), dtype=torch.float32, device=x.device)
        offs = [sum(widths[:k]) for k in range(5)]
        states, cs, keeps = [], [], []
        for k in range(4):
            w, g, b = tensors[3 * k:3 * k + 3]
            pad, dil = cfg["convs"][k]
            rm, rv, training, momentum, eps, sync, nbt = cfg["bn"][k]
            stats = None
            kp = {} if ctx.needs_input_grad[2 + 3 * k] else None      # inputs: x, cfg, then (w, gamma, beta) per branch
            keeps.append(kp)
            if FUSE_BN_STATS and training:
                run = _bn_run(rm, rv, momentum, nbt) if _sync_group(sync) is None else None
                c, stats = conv2d_fwd(x, w, None, 1, pad, dil, want_stats=True, bn_run=run, keep=kp)
            else:
                c = conv2d_fwd(x, w, None, 1, pad, dil, keep=kp)
            _, st = bn_forward_impl(c, g, b, rm, rv, None, True, training, momentum, eps, sync, nbt=nbt, stats=stats,
                                    out=cat[:, offs[k]:offs[k] + widths[k]])
            cs.append(c); states.append(st)
        w5, g5, b5 = tensors[12:15]
        rm, rv, training, momentum, eps, sync, nbt = cfg["bn"][4]
        pooled = rowsum(x, 1.0 / (H * W))                           # AdaptiveAvgPool2d(1) (aspp.py:56)
        c5 = conv2d_fwd(pooled, w5, None, 1, 0, 1)
        y5, st5 = bn_forward_impl(c5, g5, b5, rm, rv, None, True, training, momentum, eps, sync, nbt=nbt)
        broadcast_hw(y5, H, W, out=cat[:, offs[4]:offs[4] + widths[4]])   # bilinear 1x1 -> HxW (aspp.py:76)
        states.append(st5)
        ctx.keeps = keeps        # the branches' kept Winograd input transforms (released by their weight gradients)
        ctx.cfg = (cfg["convs"], [a[2] for a in cfg["bn"]], [a[4] for a in cfg["bn"]], offs, widths)
        ctx.params = tensors
        flat, meta = [], []
        for st in states:
            mean, var, count, group = st[:4]
            is_t = isinstance(count, torch.Tensor)
            meta.append((len(flat), is_t, None if is_t else count, group))
            flat += [mean, var] + ([count] if is_t else [])
        ctx.meta = meta
        ctx.save_for_backward(x, pooled, c5, *cs, *flat)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        x, pooled, c5 = ctx.saved_tensors[:3]
        cs = ctx.saved_tensors[3:7]
        flat = ctx.saved_tensors[7:]
        tensors = ctx.params
        convs, training, eps, offs, widths = ctx.cfg
        N, Cin, H, W = x.shape

        def state(i):
            pos, is_t, count, group = ctx.meta[i]
            return (flat[pos], flat[pos + 1], flat[pos + 2] if is_t else count, group)
        need_dx = ctx.needs_input_grad[0]
        grads = [None] * 15
        dx = None
        pending = None            # the previous branch's weight gradient, run inside the next BN backward
        for k in range(4):
            w, g, b = tensors[3 * k:3 * k + 3]
            pad, dil = convs[k]
            dslice = dcat[:, offs[k]:offs[k] + widths[k]]
            res = bn_backward_impl(dslice, cs[k], None, g, b, state(k), True, training[k], eps[k], False,
                                   between=pending)
            d_c, grads[3 * k + 1], grads[3 * k + 2] = res[0], res[1], res[2]
            if pending is not None:
                grads[3 * (k - 1)] = res[4]
            if need_dx:
                dx = conv2d_dgrad(d_c, w, tuple(x.shape), 1, pad, dil, out=dx, accumulate=dx is not None)
            pending = (lambda d_c=d_c, w=w, pad=pad, dil=dil, kp=ctx.keeps[k]:
                       wgrad_into_param(d_c, x, w, None, 1, pad, dil, keep=kp)[0])
        # image-pooling branch: broadcast^T = sum over pixels, then BN / 1x1 conv on N x C x 1 x 1
        w5, g5, b5 = tensors[12:15]
        g_y5 = rowsum(dcat[:, offs[4]:offs[4] + widths[4]], 1.0)
        res = bn_backward_impl(g_y5, c5, None, g5, b5, state(4), True, training[4], eps[4], False, between=pending)
        d_c5, grads[13], grads[14], grads[9] = res[0], res[1], res[2], res[4]
        grads[12], _ = wgrad_into_param(d_c5, pooled, w5, None, 1, 0, 1)
        if need_dx:
            d_pooled = conv2d_dgrad(d_c5, w5, tuple(pooled.shape), 1, 0, 1)
            dx = broadcast_hw(d_pooled, H, W, 1.0 / (H * W), out=dx, accumulate=dx is not None)   # mean^T
        return (dx, None) + tuple(grads)


def aspp_branches(x, cfg, tensors):
    return AsppFn.apply(x, cfg, *tensors)


class ForkFn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (layer3's output feeds layer4 and the deep-supervision
    head, networks/deeplabv3.py:44-50): the two incoming gradients are summed by dcfp_add_f32."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)      # an unused tap (deepsup=False) sends None, not a zero tensor + a full add
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        return add(ga, gb)


def fork(x):
    return ForkFn.apply(x) if (torch.is_grad_enabled() and x.requires_grad) else (x, x)


# ---------------------------------------------------------------- pooling
class MaxPool3x3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (networks/backbone/resnet.py:100)."""

    @staticmethod
    def forward(ctx, x):
        _require(x, "x")
        x = x.contiguous()
        N, Cc, H, W = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, Cc, Ho, Wo), dtype=torch.int32, device=x.device)
        check(_lib.lib().dcfp_maxpool3x3s2_fwd_f32(_p(x), _p(y), _p(idx), N, Cc, H, W, Ho, Wo, _stream()),
              "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (N, Cc, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, Cc, H, W, Ho, Wo = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dy.device)
        check(_lib.lib().dcfp_maxpool3x3s2_bwd_f32(_p(dy), _p(idx), _p(dx), N, Cc, H, W, Ho, Wo, _stream()),
              "maxpool_bwd")
        return dx


def maxpool3x3s2(x):
    return MaxPool3x3s2Fn.apply(x)


def rowsum(x, scale):
    N, Cc, H, W = x.shape
    x, ns = _batch_strided(x)
    y = torch.empty((N, Cc, 1, 1), dtype=torch.float32, device=x.device)
    check(_lib.lib().dcfp_rowsum_f32(_p(x), ns, _p(y), float(scale), N, Cc, H * W, _stream()), "rowsum")
    return y


def broadcast_hw(v, H, W, scale=1.0, out=None, accumulate=False):
    """y[n,c,:,:] (+)= scale * v[n,c]; out: optional destination with dense images (a channel slice)."""
    N, Cc = v.shape[0], v.shape[1]
    v = v.contiguous()
    yns = 0
    if out is None:
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=v.device)
        accumulate = False
    else:
        y = out
        st = y.stride()
        if tuple(y.shape) != (N, Cc, H, W) or st[3] != 1 or st[2] != W or st[1] != H * W:
            raise RuntimeError("broadcast_hw: out must be [N,C,H,W] with dense images")
        yns = st[0]
    check(_lib.lib().dcfp_broadcast_hw_f32(_p(v), float(scale), _p(y), yns, int(bool(accumulate)), N, Cc, H * W,
                                           _stream()), "broadcast_hw")
    return y


class GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) (networks/tools/aspp.py:56)."""

    @staticmethod
    def forward(ctx, x):
        _require(x, "x")
        ctx.hw = (x.shape[2], x.shape[3])
        return rowsum(x, 1.0 / (x.shape[2] * x.shape[3]))

    @staticmethod
    def backward(ctx, dy):
        H, W = ctx.hw
        return broadcast_hw(dy, H, W, 1.0 / (H * W))


class BroadcastHWFn(torch.autograd.Function):
    """F.interpolate of a 1x1 map to HxW (networks/tools/aspp.py:76): a constant broadcast
    for either align_corners setting."""

    @staticmethod
    def forward(ctx, v, H, W):
        _require(v, "v")
        return broadcast_hw(v, H, W)

    @staticmethod
    def backward(ctx, dy):
        return rowsum(dy, 1.0), None, None


def global_avg_pool(x):
    return GlobalAvgPoolFn.apply(x)


def broadcast_to_hw(v, H, W):
    return BroadcastHWFn.apply(v, H, W)


class ChannelScaleFn(torch.autograd.Function):
    """nn.Dropout2d with a host-drawn per-(n,c) keep/scale mask (networks/deeplabv3.py:40)."""

    @staticmethod
    def forward(ctx, x, mask):
        _require(x, "x")
        x = x.contiguous()
        N, Cc, H, W = x.shape
        mask = mask.contiguous()
        y = torch.empty_like(x)
        check(_lib.lib().dcfp_channel_scale_f32(_p(x), _p(mask), _p(y), N, Cc, H * W, _stream()),
              "channel_scale")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cc, H, W = dy.shape
        dx = torch.empty_like(dy)
        check(_lib.lib().dcfp_channel_scale_f32(_p(dy), _p(mask), _p(dx), N, Cc, H * W, _stream()),
              "channel_scale")
        return dx, None


def dropout2d(x, p, training, fixed_mask=None):
    if not training or p == 0.0:
        return x
    if fixed_mask is not None:
        mask = fixed_mask.to(device=x.device, dtype=torch.float32)
    else:
        mask = torch.empty(x.shape[0], x.shape[1], device=x.device).bernoulli_(1.0 - p).div_(1.0 - p)
    return ChannelScaleFn.apply(x, mask)


def add(a, b):
    """out = a + b through the library (gradient fan-in)."""
    a = a.contiguous(); b = b.contiguous()
    out = torch.empty_like(a)
    check(_lib.lib().dcfp_add_f32(_p(a), _p(b), _p(out), a.numel(), _stream()), "add")
    return out


# --------------------------------------------- bilinear upsample (+) cross-entropy
class UpsampleBilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners) (networks/deeplabv3.py:47,50)."""

    @staticmethod
    def forward(ctx, x, H, W, align_corners):
        _require(x, "x")
        x = x.contiguous()
        N, Cc, h, w = x.shape
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
        check(_lib.lib().dcfp_upsample_bilinear_fwd_f32(_p(x), _p(y), N, Cc, h, w, H, W,
                                                        int(bool(align_corners)), _stream()),
              "upsample_fwd")
        ctx.cfg = (N, Cc, h, w, H, W, int(bool(align_corners)))
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, h, w, H, W, ac = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((N, Cc, h, w), dtype=torch.float32, device=dy.device)
        check(_lib.lib().dcfp_upsample_bilinear_bwd_f32(_p(dy), _p(dx), N, Cc, h, w, H, W, ac, _stream()),
              "upsample_bwd")
        return dx, None, None, None


def upsample_bilinear(x, size, align_corners):
    return UpsampleBilinearFn.apply(x, int(size[0]), int(size[1]), align_corners)


def upsample_ce_forward(logits, labels, size, align_corners, ignore_index, pixel_keep=None,
                        want_gt_prob=False):
    """Returns (out2=[loss_sum, valid_count], lse[N,H,W], gt_prob or None)."""
    _require(logits, "logits")
    logits = logits.contiguous()
    if labels.dtype != torch.int64 or not labels.is_cuda:
        raise RuntimeError("upsample_ce: labels must be a CUDA int64 tensor [N,H,W]")
    labels = labels.contiguous()
    N, Cc, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    if tuple(labels.shape) != (N, H, W):
        raise RuntimeError(f"upsample_ce: labels shape {tuple(labels.shape)} != {(N, H, W)}")
    L = _lib.lib()
    ws = _workspace("ce", L.dcfp_upsample_ce_workspace_bytes(N, H, W), logits.device)
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    lse = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    gtp = torch.empty((N, H, W), dtype=torch.float32, device=logits.device) if want_gt_prob else None
    if pixel_keep is not None:
        pixel_keep = pixel_keep.to(torch.uint8).contiguous()
    check(L.dcfp_upsample_ce_fwd_f32(_p(logits), _p(labels), _p(pixel_keep), int(ignore_index), N, Cc,
                                     h, w, H, W, int(bool(align_corners)), _p(lse), _p(gtp), _p(out2),
                                     _p(ws), ws.numel(), _stream()), "upsample_ce_fwd")
    return out2, lse, gtp


class UpsampleCEFn(torch.autograd.Function):
    """loss = CrossEntropyLoss(ignore_index, 'mean')(F.interpolate(logits, size), labels),
    fused (networks/deeplabv3.py:47,50 + loss/criterion.py:60)."""

    @staticmethod
    def forward(ctx, logits, labels, H, W, align_corners, ignore_index, pixel_keep):
        out2, lse, _ = upsample_ce_forward(logits, labels, (H, W), align_corners, ignore_index, pixel_keep)
        ctx.save_for_backward(logits, labels, lse, out2, pixel_keep)
        ctx.cfg = (H, W, int(bool(align_corners)), int(ignore_index))
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, gout):
        logits, labels, lse, out2, pixel_keep = ctx.saved_tensors
        H, W, ac, ignore = ctx.cfg
        logits = logits.contiguous()
        N, Cc, h, w = logits.shape
        scale = (gout / out2[1]).reshape(1).to(torch.float32).contiguous()
        dl = torch.empty_like(logits)
        check(_lib.lib().dcfp_upsample_ce_bwd_f32(_p(logits), _p(labels.contiguous()), _p(pixel_keep),
                                                  ignore, N, Cc, h, w, H, W, ac, _p(lse), _p(scale),
                                                  _p(dl), _stream()), "upsample_ce_bwd")
        return dl, None, None, None, None, None, None


def upsample_cross_entropy(logits, labels, size, align_corners, ignore_index=255, pixel_keep=None):
    if pixel_keep is not None:
        pixel_keep = pixel_keep.to(torch.uint8).contiguous()
    return UpsampleCEFn.apply(logits, labels, int(size[0]), int(size[1]), align_corners,
                              ignore_index, pixel_keep)


# ------------------------------------------------------------------ GSRL pieces
def upsample_margin(logits, size, align_corners):
    """p1 - p2 of softmax(F.interpolate(logits, size)) per pixel, [N,H,W]."""
    _require(logits, "logits")
    logits = logits.contiguous()
    N, Cc, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    check(_lib.lib().dcfp_upsample_margin_f32(_p(logits), N, Cc, h, w, H, W, int(bool(align_corners)),
                                              _p(out), _stream()), "upsample_margin")
    return out


def maxfilter2d(x, k):
    """F.max_pool2d(x[:,None], k, stride=1, padding=k//2)[:,0] for a [N,H,W] map."""
    _require(x, "x")
    x = x.contiguous()
    N, H, W = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().dcfp_maxfilter2d_s1_f32(_p(x), _p(y), N, H, W, int(k), _stream()), "maxfilter2d")
    return y


class UpsampleWCEFn(torch.autograd.Function):
    """Per-image (sum w*CE, sum w) of the bilinearly upsampled logits with per-pixel weights w
    (loss/criterion.py:94-99), fused like UpsampleCEFn."""

    @staticmethod
    def forward(ctx, logits, labels, pix_weight, H, W, align_corners, ignore_index):
        _require(logits, "logits"); _require(pix_weight, "pix_weight")
        logits = logits.contiguous(); labels = labels.contiguous(); pix_weight = pix_weight.contiguous()
        N, Cc, h, w = logits.shape
        L = _lib.lib()
        ws = _workspace("wce", L.dcfp_upsample_wce_workspace_bytes(N, H, W), logits.device)
        lse = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
        out = torch.empty((N, 2), dtype=torch.float32, device=logits.device)
        check(L.dcfp_upsample_wce_fwd_f32(_p(logits), _p(labels), _p(pix_weight), int(ignore_index), N, Cc,
                                          h, w, H, W, int(bool(align_corners)), _p(lse), _p(out), _p(ws),
                                          ws.numel(), _stream()), "upsample_wce_fwd")
        ctx.save_for_backward(logits, labels, pix_weight, lse)
        ctx.cfg = (H, W, int(bool(align_corners)), int(ignore_index))
        return out

    @staticmethod
    def backward(ctx, gout):
        logits, labels, pix_weight, lse = ctx.saved_tensors
        H, W, ac, ignore = ctx.cfg
        N, Cc, h, w = logits.shape
        gs = gout[:, 0].contiguous().to(torch.float32)
        dl = torch.empty_like(logits)
        check(_lib.lib().dcfp_upsample_wce_bwd_f32(_p(logits), _p(labels), _p(pix_weight), ignore, N, Cc, h, w,
                                                   H, W, ac, _p(lse), _p(gs), _p(dl), _stream()),
              "upsample_wce_bwd")
        return dl, None, None, None, None, None, None


def upsample_weighted_ce(logits, labels, pix_weight, size, align_corners, ignore_index=255):
    return UpsampleWCEFn.apply(logits, labels, pix_weight, int(size[0]), int(size[1]), align_corners,
                               ignore_index)


# ------------------------------------------------------------------ inference / evaluation
def conv2d_fused_infer(x, w, scale, shift, stride=1, pad=0, dil=1, residual=None, relu=False):
    """y = act(conv(x,w)*scale[co] + shift[co] (+ residual)): conv with an eval-mode BatchNorm
    folded into its epilogue (no autograd: inference only)."""
    _require(x, "x"); _require(w, "weight")
    x = x.contiguous(); w = w.contiguous()
    d = _desc(x.shape, w.shape, stride, pad, dil)
    y = torch.empty((d.N, d.Cout, d.Hout, d.Wout), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = residual.contiguous()
        if tuple(residual.shape) != tuple(y.shape):
            raise RuntimeError("conv2d_fused_infer: residual shape mismatch")
    L = _lib.lib()
    if L.dcfp_conv2d_workspace_is_scratch(C.byref(d), _lib.CONV_FWD):
        ws, valid = _workspace("conv_scratch", L.dcfp_conv2d_workspace_bytes(C.byref(d), _lib.CONV_FWD), x.device), 0
    else:
        ws, valid = _wp_buffer(w, _lib.CONV_FWD, d, L.dcfp_conv2d_workspace_bytes(C.byref(d), _lib.CONV_FWD), "fused")
    check(L.dcfp_conv2d_fwd_fused_f32_nchw(C.byref(d), _p(x), _p(w), _p(scale.contiguous()),
                                           _p(shift.contiguous()), _p(residual), int(bool(relu)), _p(y),
                                           _p(ws), ws.numel(), valid, _stream()), "conv2d_fwd_fused")
    return y


def upsample_argmax(logits, size, align_corners):
    """argmax over classes of F.interpolate(logits, size) per pixel -> int32 [N,H,W]."""
    _require(logits, "logits")
    logits = logits.contiguous()
    N, Cc, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    pred = torch.empty((N, H, W), dtype=torch.int32, device=logits.device)
    check(_lib.lib().dcfp_upsample_argmax_f32(_p(logits), N, Cc, h, w, H, W, int(bool(align_corners)),
                                              _p(pred), _stream()), "upsample_argmax")
    return pred


def confusion_matrix(pred, gt, num_classes, ignore_index=255, out=None):
    """conf[gt, pred] += 1 over pixels with gt != ignore (evaluate.py:229-247); int64 [C,C]."""
    if pred.dtype != torch.int32 or gt.dtype != torch.int64 or not pred.is_cuda or not gt.is_cuda:
        raise RuntimeError("confusion_matrix: pred int32 / gt int64 CUDA tensors expected")
    pred = pred.contiguous(); gt = gt.contiguous()
    if out is None:
        out = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=pred.device)
    check(_lib.lib().dcfp_confusion_matrix_i64(_p(pred), _p(gt), int(ignore_index), pred.numel(),
                                               int(num_classes), _p(out), _stream()), "confusion_matrix")
    return out
