"""ops.neck -- depthwise 3x3, max pools / up-sampling, BiFPN fusion nodes and gradient slots (reference: net/bifpn.py:156-233,
net/common.py:76-152)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib, policy
from .core import *        # noqa: F401,F403
from .backbone import *        # noqa: F401,F403


# --------------------------------------------------------------------------------------------------------------
# depthwise 3x3 (zero pad 1)
# --------------------------------------------------------------------------------------------------------------
def k_dwconv(x, wk, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = new_act(n, h, w, c, x.device)
    lib().call("hn_dwconv_fwd", ptr(x), ld(x), ptr(wk), ptr(out), ld(out), n, h, w, c)
    return out


def k_dwconv_wgrad(x, dz):
    n, h, w, c = x.shape
    chunks = lib().query("hn_dwconv_wgrad_blocks", n * h * ((w + 3) // 4), c)
    part = torch.empty((chunks, c * 9), device=x.device, dtype=F32)
    lib().call("hn_dwconv_wgrad", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), n, h, w, c)
    return k_rows_reduce(part, 1, chunks, c * 9).view(c, 1, 3, 3)


def k_dwconv_bwd(dz, x, wf, geom=None, want_dx=True, into=None, queue=None, weight=None):
    """depthwise 3x3 backward in one launch (+ the partial-row reduce): (dx | None, dweight [C,1,3,3]).  geom: level-packed tensors;
    into: an existing tensor dx is ADDED to (GradSlot accumulation); queue (GradQueue): the partial-row fold of the weight gradient is
    queued for `weight` instead of launched (returns None for it)."""
    c = x.shape[3]
    if geom is None:
        n, h, w, _ = x.shape
        H, W, nl, align = (ctypes.c_int * 1)(h), (ctypes.c_int * 1)(w), 1, 1
    else:
        nl, H, W, _, _ = _geom_arrays(geom)
        n, align = geom[0], LEVEL_ALIGN
    blocks = lib().query("hn_dwconv_bwd_blocks_levels", n, c, nl, ctypes.addressof(H), ctypes.addressof(W))
    part = torch.empty((blocks, c * 9), device=x.device, dtype=F32)
    dx = None
    if want_dx:
        dx = torch.empty_like(dz) if into is None else into
    lib().call("hn_dwconv_bwd_levels", ptr(dz), ld(dz), ptr(x), ld(x), ptr(wf), ptr(dx), ld(dx) if dx is not None else 0, ptr(part), n, c, nl,
               ctypes.addressof(H), ctypes.addressof(W), align, 0 if into is None else 1)
    if queue is not None:
        queue.add_rows(weight, part, blocks, c * 9, (c, 1, 3, 3))
        return dx, None
    return dx, k_rows_reduce(part, 1, blocks, c * 9).view(c, 1, 3, 3)


class DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        wk, wf = pack_dw_weight(weight)
        ctx.queue, ctx.wref = (cur_queue() if weight.requires_grad else None), weight
        ctx.save_for_backward(x, wf)
        return k_dwconv(x, wk)

    @staticmethod
    def backward(ctx, dout):
        x, wf = ctx.saved_tensors
        dout = dense(dout)
        return k_dwconv_bwd(dout, x, wf, want_dx=ctx.needs_input_grad[0], queue=ctx.queue, weight=ctx.wref)


# --------------------------------------------------------------------------------------------------------------
# max pools, up-sampling
# --------------------------------------------------------------------------------------------------------------
def k_maxpool(x, mode, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = new_act(n, h // 2, w // 2, c, x.device)
    lib().call("hn_maxpool_fwd", ptr(x), ld(x), ptr(out), ld(out), n, h, w, c, mode)
    return out


def k_maxpool_bwd(x, dout, mode, wscale=None, into=None, accumulate=False):
    """into: destination of x's shape (default: a new tensor); accumulate: add to what `into` already holds (GradSlot)"""
    n, h, w, c = x.shape
    dx = new_act(n, h, w, c, x.device) if into is None else into
    arg = torch.empty((n * (h // 2) * (w // 2) * c,), device=x.device, dtype=torch.uint8)
    lib().call("hn_maxpool_bwd2", ptr(x), ld(x), ptr(dout), ld(dout), ptr(dx), ld(dx), ptr(wscale), ptr(arg), n, h, w, c, mode,
               1 if accumulate else 0)
    return dx


class MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        ctx.save_for_backward(x)
        return k_maxpool(x, mode)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return k_maxpool_bwd(x, dense(dout), ctx.mode), None


# --------------------------------------------------------------------------------------------------------------
# BiFPN fusion node
# --------------------------------------------------------------------------------------------------------------
def _fuse_args(ins, modes):
    arr_p = (ctypes.c_void_p * 3)(*[ptr(t) if t is not None else None for t in ins])
    arr_l = (ctypes.c_int * 3)(*[ld(t) if t is not None else 0 for t in ins])
    arr_m = (ctypes.c_int * 3)(*modes)
    return arr_p, arr_l, arr_m


class GradSlot:
    """Where the consumers of one multi-consumer tensor meet in backward: the first consumer to run allocates `buf` and stores its
    contribution, the later ones add theirs in place inside their own kernels.  Share.backward hands `buf` to the producer."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class Share(torch.autograd.Function):
    """Identity with n aliases, one per consumer of x (a BiFPN map feeds 2-3 fusion nodes, net/bifpn.py:186-231).  Consumers that were
    given the slot accumulate their input gradient into slot.buf inside their own backward kernels and return None for this input;
    consumers that do not know about slots return a gradient as usual and it is added here with one elementwise launch.  Without this
    node the autograd engine sums the k gradients of a k-consumer tensor with k-1 separate ATen add kernels (54 per step)."""

    @staticmethod
    def forward(ctx, x, slot, n):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        buf, ctx.slot.buf = ctx.slot.buf, None
        gs = [dense(g) for g in grads if g is not None]
        if buf is None and gs:
            buf = gs.pop(0)
            if gs and buf.data_ptr() in [g.data_ptr() for g in gs]:     # (never: every consumer returns its own tensor)
                buf = buf.clone()
        while gs:                                                   # up to three addends per launch, fp32 sum, one rounding
            part, gs = gs[:3], gs[3:]
            part += [None] * (3 - len(part))
            lib().call("hn_add_n", ptr(buf), ld(buf), ptr(part[0]), ld(part[0]), ptr(part[1]), ld(part[1]) if part[1] is not None else 0,
                       ptr(part[2]), ld(part[2]) if part[2] is not None else 0, rows(buf), buf.shape[3])
        return buf, None, None


def share(x, n):
    """-> (aliases, slot) for a tensor with n consumers; n == 1: the tensor itself and no slot"""
    if n <= 1 or not x.requires_grad:
        return (x,) * max(n, 1), None
    slot = GradSlot()
    return Share.apply(x, slot, n), slot


# The 2x2 gradient sums of up-sampled fusion inputs inside hn_fuse_bwd instead of 12 hn_sum2x2 launches per step.  With the generic quad
# walk (four pixels in sequence per thread) this measured SLOWER (861 vs 863 img/s); with fuse_bwd_quads_kernel, which issues every load of
# a quad before the first use, 869.  HN_FUSE_SUM2X2=0: the separate launches.
FUSE_SUM2X2 = policy("HN_FUSE_SUM2X2", "1") != "0"
FUSE_ARG = policy("HN_FUSE_ARG", "1") != "0"      # the fusion backward kernel leaves the pooling arg-max bytes behind (0: hn_maxpool_bwd2's own pass)


class Fuse(torch.autograd.Function):
    """out = swish(w0*T0(a) + w1*T1(b) [+ w2*T2(c)]), w = relu(p)/(sum relu(p) + 1e-4) from the raw fusion parameter p (2 or 3 values).
    slots (optional): one GradSlot | None per input -- the gradient of a slotted input is accumulated into slot.buf (see Share)."""

    @staticmethod
    def forward(ctx, praw, m0, m1, m2, a, b, c, slots=None):
        ins = [a, b, c]
        modes = [m0, m1, m2]
        n, h, wd, ch = a.shape                        # input 0 is always at the output resolution (mode 1)
        assert m0 == 1
        dev = a.device
        w = torch.empty((3,), device=dev, dtype=F32)
        out = new_act(n, h, wd, ch, dev)
        ap, al, am = _fuse_args(ins, modes)
        lib().call("hn_fuse_fwd_raw", ctypes.addressof(ap), ctypes.addressof(al), ctypes.addressof(am), ptr(praw), praw.numel(), 1e-4, ptr(w),
                   ptr(out), ld(out), n, h, wd, ch)
        ctx.modes = modes
        ctx.slots = slots if slots is not None else (None, None, None)
        ctx.queue, ctx.pref = (cur_queue() if praw.requires_grad else None), praw
        ctx.save_for_backward(praw, w, *[t for t in ins if t is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        praw, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
        rest = list(ctx.saved_tensors[2:])
        ins = [rest.pop(0) if m else None for m in ctx.modes]
        dpraw, dins = fuse_backward(praw, w, ins, ctx.modes, ctx.slots, ctx.queue, ctx.pref, dense(dout))
        return dpraw, None, None, None, dins[0], dins[1], dins[2], None


def fuse_backward(praw, w, ins, modes, slots, queue, pref, dout):
    """backward of out = swish(sum_i w_i T_i(in_i)) given dout (dense NHWC bf16) -> (dpraw | None when queued, [d in_i | None] * 3):
    g = dout * swish'(pre) and the identity / up-sampled inputs' gradients out of ONE kernel, max-pooled inputs by the scatter pass
    behind it; slotted inputs are accumulated into their GradSlot and come back as None (see Share)."""
    n, h, wd, ch = dout.shape
    dev = dout.device
    g = new_act(n, h, wd, ch, dev)
    # destination of every input gradient: a fresh tensor, or the slot's buffer (first consumer: allocate + store, later: accumulate)
    dst, accum = [None] * 3, [0] * 3
    for i, m in enumerate(modes):
        if not m:
            continue
        sl = slots[i]
        if sl is not None and sl.buf is not None:
            dst[i], accum[i] = sl.buf, 1
        else:
            dst[i] = new_act(*ins[i].shape, dev)
            if sl is not None:
                sl.buf = dst[i]
    ap, al, am = _fuse_args(ins, modes)
    # mode 1 (same grid) and mode 2 (nearest x2 of a half-resolution map: the kernel sums its 2x2 quads itself) gradients come out of
    # the fusion kernel; mode 3 (max-pooled input) is routed by the max-pool backward below
    inside = (1, 2) if FUSE_SUM2X2 else (1,)
    dp_ = (ctypes.c_void_p * 3)(*[ptr(dst[i]) if modes[i] in inside else None for i in range(3)])
    dl = (ctypes.c_int * 3)(*[ld(dst[i]) if modes[i] in inside else 0 for i in range(3)])
    da = (ctypes.c_int * 3)(*accum)
    blocks = lib().query("hn_fuse_bwd_blocks", n, h, wd, ch)
    pw = torch.empty((blocks, 3), device=dev, dtype=F32)
    # the fusion kernel recomputes the pooling windows of a max-pooled input anyway: it leaves their arg-max bytes behind, and the
    # max-pool backward below is its scatter pass alone (hn_maxpool_bwd2's own arg pass was one more ~7 us launch per pooled input)
    args = [torch.empty((n * h * wd * ch,), device=dev, dtype=torch.uint8) if (m == 3 and FUSE_ARG) else None for m in modes]
    aa = (ctypes.c_void_p * 3)(*[ptr(t) for t in args])
    lib().call("hn_fuse_bwd_arg", ctypes.addressof(ap), ctypes.addressof(al), ctypes.addressof(am), ptr(w), ptr(dout), ld(dout), ptr(g),
               ld(g), ctypes.addressof(dp_), ctypes.addressof(dl), ctypes.addressof(da), ptr(pw), ctypes.addressof(aa), n, h, wd, ch)
    if queue is not None:
        dpraw = queue.add_fuse(pref, pw, blocks, 1e-4)
    else:
        dpraw = torch.empty_like(praw)
        lib().call("hn_fuse_dweights", ptr(pw), blocks, ptr(praw), praw.numel(), 1e-4, ptr(dpraw))
    for i, m in enumerate(modes):
        if m == 2 and not FUSE_SUM2X2:                         # nearest x2 of a half-res input: 2x2 sum of g
            lib().call("hn_sum2x2", ptr(g), ld(g), ptr(dst[i]), ld(dst[i]), ptr(w[i]), n, h // 2, wd // 2, ch, accum[i])
        elif m == 3 and not FUSE_ARG:
            k_maxpool_bwd(ins[i], g, 0, wscale=w[i], into=dst[i], accumulate=bool(accum[i]))
        elif m == 3:                                           # zero-pad-same max pool of a double-res input
            lib().call("hn_maxpool_bwd_from_arg", ptr(args[i]), ptr(g), ld(g), ptr(dst[i]), ld(dst[i]), ptr(w[i]), n, 2 * h, 2 * wd, ch, 0,
                       accum[i])
    return dpraw, [None if (slots[i] is not None or not modes[i]) else dst[i] for i in range(3)]


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
