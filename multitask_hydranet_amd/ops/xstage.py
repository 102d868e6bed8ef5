"""ops.xstage -- the stride-1 identity XBlocks of a backbone stage as ONE persistent launch (csrc/hn_xstage.hip; reference:
net/anynet.py:65-76,84-86).  Forward: `hn_xstage_fwd` writes every tensor XBlockFn.forward saves, for all blocks of the run; backward walks
the blocks in reverse through XBlockFn.backward (the 21-launch chain) on those tensors."""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import torch

from .._lib import lib, policy
from .core import *        # noqa: F401,F403
from . import backbone as _bb

XSTAGE = policy("HN_XSTAGE", "1") != "0"          # persistent stage kernel for the identity blocks (0: the launch chain)
XSTAGE_MODE = int(policy("HN_XSTAGE_MODE", "0"))  # 0: XCD-local counters, 1: agent-scope counters + fences (hydranet_hip.h)
PER_BLOCK = 19                                    # parameter tensors of one block, in XBlockFn.forward's argument order (after x)

_WS = {}          # device index -> (workspace uint8 tensor, status view)
_CHECKED = {}     # (device, shape signature) -> number of eager launches whose status word was read back
_DISABLED = [False]


def xstage_ws(dev):
    """the launches' sync workspace: one per device, zeroed once (hydranet_hip.h: counters and tags continue across launches)"""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _WS.get(key)
    if w is None:
        nbytes = lib().query("hn_xstage_ws_bytes")
        buf = torch.zeros((nbytes,), device=dev, dtype=torch.uint8)
        w = _WS[key] = (buf, buf[256:260].view(torch.int32))
    return w


def xstage_status(dev) -> int:
    """status word of the device's workspace (synchronises)"""
    return int(xstage_ws(dev)[1].item())


def xstage_assert_ok(dev=None):
    """raise if a persistent launch on `dev` (default: every device that has a workspace) ended with an expired wait (synchronises).  The
    launches themselves never hang: a wait that expires raises the status word and the workgroups retire, leaving invalid outputs -- callers
    that replay captured steps (where the per-launch check cannot run) call this where they synchronise anyway."""
    for key, (buf, status) in list(_WS.items()):
        if dev is not None and (dev.index if dev.index is not None else torch.cuda.current_device()) != key:
            continue
        st = int(status.item())
        if st != 0:
            _DISABLED[0] = True
            raise RuntimeError(f"persistent stage kernel: a bounded wait expired on device {key} (status 0x{st:x}): the results of the affected "
                               "steps are invalid; the launch chain (HN_XSTAGE=0 semantics) takes over for the rest of the process")


def xstage_ok(x, w1, cs) -> bool:
    """the persistent kernel covers this run of identity blocks (shape, batch, dense rows); else the caller keeps the launch chain"""
    if not (XSTAGE and not _DISABLED[0] and x.is_cuda and x.dim() == 4 and x.dtype == BF16 and x.is_contiguous()):
        return False
    n, h, w, c = x.shape
    return w1.shape[0] == c and w1.shape[1] == c and lib().query("hn_xstage_supported", n, h, w, c, cs) != 0


def xstage_forward_raw(x, params, eps, momentum, stamps=None, mode=None):
    """-> dict of the run's tensors.  params: nb * PER_BLOCK tensors (w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, sw1, sb1, sw2, sb2, w3, g3,
    b3, rm3, rv3 per block)."""
    nb = len(params) // PER_BLOCK
    n, h, w, c = x.shape
    dev = x.device
    cs = params[10].shape[0]
    tab = (ctypes.c_long * (19 * nb))()
    packs = []
    for b in range(nb):
        w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, sw1, sb1, sw2, sb2, w3, g3, b3, rm3, rv3 = params[b * PER_BLOCK:(b + 1) * PER_BLOCK]
        wp1, wt1 = pack_conv_weight(w1)
        wk2, wd2 = pack_gconv_diag(w2)
        wp3, wt3 = pack_conv_weight(w3)
        packs.append((wt1, wd2, wt3, None))
        tab[19 * b:19 * b + 19] = [t.data_ptr() for t in (wp1, wk2, wp3, sw1, sb1, sw2, sb2, g1, b1, rm1, rv1, g2, b2, rm2, rv2, g3, b3, rm3, rv3)]
    acts = {k: torch.empty((nb, n, h, w, c), device=dev, dtype=BF16) for k in ("z1", "a", "z2", "bg", "z3", "out")}
    coef = torch.empty((nb, 3, 4, c), device=dev, dtype=F32)
    pooled = torch.empty((nb, n, c), device=dev, dtype=F32)
    hid = torch.empty((nb, n, cs), device=dev, dtype=F32)
    gate = torch.empty((nb, n, c), device=dev, dtype=F32)
    ws, status = xstage_ws(dev)
    lib().call("hn_xstage_fwd", ctypes.addressof(tab), nb, ptr(x), ptr(acts["z1"]), ptr(acts["a"]), ptr(acts["z2"]), ptr(acts["bg"]),
               ptr(acts["z3"]), ptr(acts["out"]), ptr(coef), ptr(pooled), ptr(hid), ptr(gate), n, h, w, c, cs, float(eps), float(momentum),
               1.0 / (h * w), ptr(ws), ptr(stamps), XSTAGE_MODE if mode is None else mode)
    # the first eager launches of a shape are checked: a launch that could not become co-resident raises the status word instead of hanging
    key = (dev.index, n, h, w, c, nb)
    if _CHECKED.get(key, 0) < 2 and not torch.cuda.is_current_stream_capturing():
        _CHECKED[key] = _CHECKED.get(key, 0) + 1
        st = int(status.item())
        if st != 0:
            _DISABLED[0] = True
            raise RuntimeError(f"hn_xstage_fwd: a bounded wait expired (status 0x{st:x}): the persistent stage kernel could not become "
                               "co-resident on this device; it is disabled for this process (the launch chain runs instead)")
    return dict(acts, coef=coef, pooled=pooled, hid=hid, gate=gate, packs=packs)


XSTAGE_BWD = policy("HN_XSTAGE_BWD", "1") != "0"  # the run's backward as one persistent launch too (0: XBlockFn.backward per block)


def xstage_backward_raw(dout, r_or_saved, packs, sws, stamps=None, mode=None):
    """hn_xstage_bwd on the stacked forward tensors -> dict(dz1, dz2, dz3 [nb, n, h, w, c], dx, dgb [nb, 3, 2, c], dpre2, dpre1).
    r_or_saved: dict with z1, z2, z3, out, coef, hid, gate; packs: per block (wt1, wd2, wt3, ...); sws: per block (se.1.weight, se.3.weight)."""
    z1, z2, z3, out, coef, hid, gate = (r_or_saved[k] for k in ("z1", "z2", "z3", "out", "coef", "hid", "gate"))
    nb, n, h, w, c = z1.shape
    cs = hid.shape[2]
    dev = z1.device
    tab = (ctypes.c_long * (5 * nb))()
    for b in range(nb):
        tab[5 * b:5 * b + 5] = [packs[b][0].data_ptr(), packs[b][1].data_ptr(), packs[b][2].data_ptr(), sws[b][0].data_ptr(), sws[b][1].data_ptr()]
    dz = {k: torch.empty((nb, n, h, w, c), device=dev, dtype=BF16) for k in ("dz1", "dz2", "dz3")}
    dx = torch.empty((n, h, w, c), device=dev, dtype=BF16)
    dgb = torch.empty((nb, 3, 2, c), device=dev, dtype=F32)
    dpre2 = torch.empty((nb, n, c), device=dev, dtype=F32)
    dpre1 = torch.empty((nb, n, cs), device=dev, dtype=F32)
    ws, status = xstage_ws(dev)
    lib().call("hn_xstage_bwd", ctypes.addressof(tab), nb, ptr(dout), ptr(z1), ptr(z2), ptr(z3), ptr(out), ptr(coef), ptr(hid), ptr(gate),
               ptr(dz["dz1"]), ptr(dz["dz2"]), ptr(dz["dz3"]), ptr(dx), ptr(dgb), ptr(dpre2), ptr(dpre1), n, h, w, c, cs, ptr(ws), ptr(stamps),
               XSTAGE_MODE if mode is None else mode)
    key = ("bwd", dev.index, n, h, w, c, nb)
    if _CHECKED.get(key, 0) < 2 and not torch.cuda.is_current_stream_capturing():
        _CHECKED[key] = _CHECKED.get(key, 0) + 1
        st = int(status.item())
        if st != 0:
            _DISABLED[0] = True
            raise RuntimeError(f"hn_xstage_bwd: a bounded wait expired (status 0x{st:x}): the persistent stage kernel could not become "
                               "co-resident on this device; it is disabled for this process (the launch chain runs instead)")
    return dict(dz, dx=dx, dgb=dgb, dpre2=dpre2, dpre1=dpre1)


class XStageFn(torch.autograd.Function):
    """out = XBlock_{nb}(... XBlock_1(x)) for identity blocks (stride 1, no projection shortcut), training mode.  One forward launch; the
    backward is one launch too (hn_xstage_bwd) when the stage defers its weight gradients (group), else XBlockFn.backward per block."""

    @staticmethod
    def forward(ctx, x, group, eps, momentum, *params):
        nb = len(params) // PER_BLOCK
        r = xstage_forward_raw(x, params, eps, momentum)
        n, h, w, c = x.shape
        m = n * h * w
        ctx.persistent_bwd = XSTAGE_BWD and group is not None
        # hand-over of the BatchNorm-3 backward partials between consecutive identity blocks (XBlockFn.forward: BN3_PARTS_FROM_DGRAD): only
        # the per-block backward uses it
        hand_ok = (not ctx.persistent_bwd and _bb.BN3_PARTS_FROM_DGRAD and _bb.EPILOGUE_STATS and group is not None and c > 64
                   and lib().query("hn_nt_stat_rows", m, c) == (m + 63) // 64 <= MAX_PROLOGUE_ROWS)
        first = (None, None)
        if hand_ok:
            last = getattr(group, "bn3_last", None)
            if last is not None and last[0] == x.data_ptr() and last[1] == tuple(x.shape):
                first = (last[2], last[3])
        ctx.hand_ok = hand_ok
        ctx.has_first = first[0] is not None
        ctx.packs, ctx.group, ctx.nb = r["packs"], group, nb
        ctx.wrefs = [(p[0], p[14], None, p[5], p[10], p[11], p[12], p[13]) for p in (params[b * PER_BLOCK:(b + 1) * PER_BLOCK] for b in range(nb))]
        sws = [t for b in range(nb) for t in (params[b * PER_BLOCK + 10], params[b * PER_BLOCK + 12])]
        ctx.save_for_backward(x, r["z1"], r["a"], r["z2"], r["bg"], r["z3"], r["out"], r["coef"], r["pooled"], r["hid"], r["gate"],
                              *([first[0], first[1]] if ctx.has_first else []), *sws)
        out = r["out"][nb - 1]
        if group is not None:
            group.bn3_last = (out.data_ptr(), tuple(out.shape), r["z3"][nb - 1], r["coef"][nb - 1, 2]) if _bb.BN3_PARTS_FROM_DGRAD else None
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        x, z1, a, z2, bg, z3, out, coef, pooled, hid, gate = saved[:11]
        k = 11
        first = (None, None)
        if ctx.has_first:
            first = (saved[11], saved[12])
            k = 13
        sws = [(saved[k + 2 * b], saved[k + 2 * b + 1]) for b in range(ctx.nb)]
        nb, group = ctx.nb, ctx.group
        n, h, w, c = x.shape
        grads = [None] * (nb * PER_BLOCK)
        dout = dense(dout)
        if ctx.persistent_bwd and not _DISABLED[0]:
            if not dout.is_contiguous():
                dout = dout.contiguous()
            r = xstage_backward_raw(dout, dict(z1=z1, z2=z2, z3=z3, out=out, coef=coef, hid=hid, gate=gate), ctx.packs, sws)
            grid = (n, h, w)
            for b in reversed(range(nb)):           # the deferred parameter gradients, queued in the order XBlockFn.backward queues them
                w1_, w3_, _, w2_, sw1_, sb1_, sw2_, sb2_ = ctx.wrefs[b]
                group.add(w3_, bg[b], r["dz3"][b], 0, grid, c, c)
                group.add_outer(sw2_, sb2_, r["dpre2"][b], hid[b])
                group.add_outer(sw1_, sb1_, r["dpre1"][b], pooled[b])
                group.add_gconv(w2_, a[b], r["dz2"][b], grid, c)
                group.add(w1_, x if b == 0 else out[b - 1], r["dz1"][b], 0, grid, c, c)
                g = grads[b * PER_BLOCK:(b + 1) * PER_BLOCK]
                g[1], g[2], g[6], g[7], g[15], g[16] = (r["dgb"][b, 0, 0], r["dgb"][b, 0, 1], r["dgb"][b, 1, 0], r["dgb"][b, 1, 1],
                                                        r["dgb"][b, 2, 0], r["dgb"][b, 2, 1])
                grads[b * PER_BLOCK:(b + 1) * PER_BLOCK] = g
            return (r["dx"], None, None, None, *grads)
        for b in reversed(range(nb)):
            z3p = coef3p = None
            if ctx.hand_ok and b > 0:
                z3p, coef3p = z3[b - 1], coef[b - 1, 2]
            elif ctx.hand_ok:
                z3p, coef3p = first
            fake = SimpleNamespace(saved_tensors=(x if b == 0 else out[b - 1], z1[b], a[b], z2[b], z3[b], out[b], coef[b, 0], coef[b, 1], coef[b, 2],
                                                  pooled[b], hid[b], gate[b], sws[b][0], sws[b][1], bg[b], None, None, z3p, coef3p),
                                   training=True, stride=1, packs=ctx.packs[b], group=group, wrefs=ctx.wrefs[b], needs_input_grad=(True,) * 30)
            ret = _bb.XBlockFn.backward(fake, dout)
            dout = ret[0]
            grads[b * PER_BLOCK:(b + 1) * PER_BLOCK] = ret[1:1 + PER_BLOCK]
        return (dout, None, None, None, *grads)


def xstage_apply(x, group, eps, momentum, params):
    return XStageFn.apply(x, group, eps, momentum, *params)


__all__ = ["XSTAGE", "XSTAGE_BWD", "XStageFn", "xstage_apply", "xstage_ok", "xstage_forward_raw", "xstage_backward_raw", "xstage_ws",
           "xstage_status", "xstage_assert_ok", "PER_BLOCK"]
