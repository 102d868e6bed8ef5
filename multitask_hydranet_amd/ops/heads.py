"""ops.heads -- head output convs, lane-head input fusion and the level-packed detection towers (reference: head_detect/detection.py:11-83,
head_lane/lanedetect.py:66-96)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib, policy
from .core import *        # noqa: F401,F403
from .backbone import *        # noqa: F401,F403
from .neck import *        # noqa: F401,F403
from .seg import *        # noqa: F401,F403


# --------------------------------------------------------------------------------------------------------------
# head outputs: (optional depthwise 3x3) -> 1x1 conv + bias -> fp32, written straight into the caller's layout
# --------------------------------------------------------------------------------------------------------------
class HeadOut(torch.autograd.Function):
    """feats: list of NHWC maps (pyramid levels) sharing one (dw, pw, bias) -> fp32 [N, sum_l H_l*W_l*rep, k] (rep*k = Cout).
    Covers SeparableConvBlock(norm=False) headers of the det towers (head_detect/detection.py:24,36-44,61,73-83) and, with
    dw_weight=None, the final 1x1(+bias) of a lane branch (head_lane/lanedetect.py:49,56,63,86-92)."""

    @staticmethod
    def forward(ctx, dw_weight, pw_weight, bias, k, act, *feats):
        n = feats[0].shape[0]
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        wp, wt = pack_conv_weight(pw_weight)
        rows_total = sum(f.shape[1] * f.shape[2] for f in feats)
        rep = cout // k
        out = torch.empty((n, rows_total * rep, k), device=feats[0].device, dtype=F32)
        ldc, img_stride, ch_off = cout, rows_total * cout, 0
        mids = []
        off = 0
        for f in feats:
            _, h, w, _ = f.shape
            mid = k_dwconv(f, pack_dw_weight(dw_weight)[0]) if dw_weight is not None else f
            mids.append(mid if dw_weight is not None else None)
            dst = out.view(-1)[off * ldc + ch_off:]
            k_gemm_nt(mid, None, 0, (n, h, w), wp, cout, kp32(cin), 1, bias=bias, act=act, out=dst, out_f32=True, ldc=ldc,
                      rpi=h * w, img_stride=img_stride)
            off += h * w
        ctx.meta = (k, act, ch_off, ldc, img_stride, dw_weight is not None, len(feats))
        ctx.wt = wt
        ctx.save_for_backward(dw_weight, pw_weight, out if act == ACT_SIGMOID else None, *feats, *[t for t in mids if t is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        k, act, ch_off, ldc, img_stride, has_dw, nf = ctx.meta
        saved = ctx.saved_tensors
        dw_weight, pw_weight, yout = saved[0], saved[1], saved[2]
        feats = saved[3:3 + nf]
        mids = saved[3 + nf:] if has_dw else feats
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        dout = dout.contiguous()
        dev = dout.device
        ldz = pad8(cout)
        dpw = dbias = ddw = None                             # first level: the gradients themselves; further levels accumulate
        dfeats = []
        off = 0
        acc = lambda tot, part: part if tot is None else tot.add_(part)
        for f, mid in zip(feats, mids):
            n, h, w, _ = f.shape
            m = n * h * w
            dz = new_act(n, h, w, ldz, dev)
            base = off * ldc + ch_off
            lib().call("hn_head_grad", ptr(dout.view(-1)[base:]), ptr(yout.view(-1)[base:]) if yout is not None else None, h * w, img_stride,
                       ldc, cout, ptr(dz), ldz, m, 1 if act == ACT_SIGMOID else 0)
            dpw_l, db_l = k_gemm_tn(mid, None, 0, (n, h, w), dz, cout, kp32(cin), 1, cin, want_bias=True)
            dbias = acc(dbias, db_l)
            dpw = acc(dpw, dpw_l)
            dmid, _, _ = k_gemm_nt(dz, None, 0, (n, h, w), ctx.wt, cin, kp32(cout), 1, c0=ldz, c1=0)
            if has_dw:
                df, dwl = k_dwconv_bwd(dmid, f, pack_dw_weight(dw_weight)[1])
                ddw = acc(ddw, dwl)
                dfeats.append(df)
            else:
                dfeats.append(dmid)
            off += h * w
        return (ddw, dpw, dbias, None, None, *dfeats)


class HeadOutCat(torch.autograd.Function):
    """cat([1x1(ta; wa, ba), 1x1(tb; wb, bb)], -1) as fp32 [N, H*W, ca + cb]: both branch outputs are written side by side by their GEMM
    epilogues (channel offset + row stride), so neither the concat (head_lane/lanedetect.py:93: cat([down, up], 1)) nor the slicing of
    its gradient exist as separate kernels."""

    @staticmethod
    def forward(ctx, wa, ba, wb, bb, ta, tb):
        n, h, w, _ = ta.shape
        ca, cb = wa.shape[0], wb.shape[0]
        ldc = ca + cb
        out = torch.empty((n, h * w, ldc), device=ta.device, dtype=F32)
        wts = []
        for wgt, bias, t, off in ((wa, ba, ta, 0), (wb, bb, tb, ca)):
            wp, wt = pack_conv_weight(wgt)
            wts.append(wt)
            k_gemm_nt(t, None, 0, (n, h, w), wp, wgt.shape[0], kp32(wgt.shape[1]), 1, bias=bias, out=out.view(-1)[off:], out_f32=True,
                      ldc=ldc, rpi=h * w, img_stride=h * w * ldc)
        ctx.wts = wts
        ctx.save_for_backward(ta, tb, wa, wb)
        return out

    @staticmethod
    def backward(ctx, dout):
        ta, tb, wa, wb = ctx.saved_tensors
        n, h, w, _ = ta.shape
        ca, cb = wa.shape[0], wb.shape[0]
        ldc = ca + cb
        dout = dout.contiguous()
        dev = dout.device
        res = []
        for wgt, wt, t, off in ((wa, ctx.wts[0], ta, 0), (wb, ctx.wts[1], tb, ca)):
            cout, cin = wgt.shape[0], wgt.shape[1]
            ldz = pad8(cout)
            dz = new_act(n, h, w, ldz, dev)
            lib().call("hn_head_grad", ptr(dout.view(-1)[off:]), None, h * w, h * w * ldc, ldc, cout, ptr(dz), ldz, n * h * w, 0)
            dwgt, dbias = k_gemm_tn(t, None, 0, (n, h, w), dz, cout, kp32(cin), 1, cin, want_bias=True)
            dt, _, _ = k_gemm_nt(dz, None, 0, (n, h, w), wt, cin, kp32(cout), 1, c0=ldz, c1=0)
            res.append((dwgt, dbias, dt))
        return res[0][0], res[0][1], res[1][0], res[1][1], res[0][2], res[1][2]


# --------------------------------------------------------------------------------------------------------------
# lane-head input fusion: cat[mp(mp(P3)), mp(P4), P5, up2(P6)] with nn.MaxPool2d(3,2,1)   (head_lane/lanedetect.py:76-80)
# --------------------------------------------------------------------------------------------------------------
class LaneConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p3, p4, p5, p6):
        n, h, w, c = p5.shape
        out = new_act(n, h, w, 4 * c, p5.device)
        t3 = k_maxpool(p3, 1)
        k_maxpool(t3, 1, out=out[..., 0:c])
        k_maxpool(p4, 1, out=out[..., c:2 * c])
        k_eltwise(2, p5, alpha=1.0, out=out[..., 2 * c:3 * c])
        lib().call("hn_up2_fwd", ptr(p6), ld(p6), ptr(out[..., 3 * c:]), ld(out), n, h // 2, w // 2, c)
        ctx.save_for_backward(p3, p4, t3)
        return out

    @staticmethod
    def backward(ctx, dout):
        p3, p4, t3 = ctx.saved_tensors
        dout = dense(dout)
        n, h, w, c4 = dout.shape
        c = c4 // 4
        dt3 = k_maxpool_bwd(t3, dout[..., 0:c], 1)
        d3 = k_maxpool_bwd(p3, dt3, 1)
        d4 = k_maxpool_bwd(p4, dout[..., c:2 * c], 1)
        d5 = k_eltwise(2, dout[..., 2 * c:3 * c], alpha=1.0)
        d6 = new_act(n, h // 2, w // 2, c, dout.device)
        lib().call("hn_sum2x2", ptr(dout[..., 3 * c:]), ld(dout), ptr(d6), ld(d6), None, n, h // 2, w // 2, c, 0)
        return d3, d4, d5, d6



# --------------------------------------------------------------------------------------------------------------
# Level-packed det-head towers.  Regressor / Classifier apply the SAME SeparableConvBlock to the five pyramid levels and differ only
# in the per-level BatchNorm (head_detect/detection.py:20-35,57-72).  Launched level by level that is ~650 launches per step, most of
# them on 4x8 ... 16x32 maps where a launch is pure latency.  Here the levels live stacked in one [sum_l N*H_l*W_l, C] tensor: the
# depthwise conv, the pointwise GEMM (+ statistics), the BatchNorm passes and every backward kernel run ONCE for all levels, with the
# per-level BatchNorm parameters selected per row block inside the kernels.  Levels whose row count is not a multiple of 128 (640x640: P7 =
# 25 rows per image) are padded up to one ("ragged" packing): the depthwise kernel writes zeros to the alignment rows, so the pointwise conv
# output there is exactly bf16(bias) and is subtracted from the BatchNorm statistics; gradients at those rows are zero by construction.
# --------------------------------------------------------------------------------------------------------------
def packed_rows(geom):
    n, hs, ws = geom
    return sum(_pad_rows(n * h * w) for h, w in zip(hs, ws))


def has_pad_rows(geom):
    n, hs, ws = geom
    return any((n * h * w) % LEVEL_ALIGN for h, w in zip(hs, ws))


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def level_views(packed, geom):
    """NHWC views of the (real rows of the) levels of a packed [1, 1, rows, C] tensor; every level starts on an aligned row"""
    n, hs, ws = geom
    out, off = [], 0
    for h, w in zip(hs, ws):
        m = n * h * w
        out.append(packed[0, 0, off:off + m].view(n, h, w, packed.shape[3]))
        off += _pad_rows(m)
    return out


def k_dwconv_levels(x, wk, geom, into=None):
    """into: an existing tensor the result is ADDED to (GradSlot accumulation of a data gradient)"""
    nl, H, W, _, _ = _geom_arrays(geom)
    out = torch.empty_like(x) if into is None else into
    lib().call("hn_dwconv_fwd_levels", ptr(x), ld(x), ptr(wk), ptr(out), ld(out), geom[0], x.shape[3], nl, ctypes.addressof(H),
               ctypes.addressof(W), LEVEL_ALIGN, 0 if into is None else 1)
    return out


def k_dwconv_wgrad_levels(x, dz, geom):
    nl, H, W, _, _ = _geom_arrays(geom)
    c = x.shape[3]
    chunks = lib().query("hn_dwconv_wgrad_blocks", sum(geom[0] * hh * ((ww + 3) // 4) for hh, ww in zip(geom[1], geom[2])), c)
    part = torch.empty((chunks, c * 9), device=x.device, dtype=F32)
    lib().call("hn_dwconv_wgrad_levels", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), geom[0], c, nl, ctypes.addressof(H), ctypes.addressof(W),
               LEVEL_ALIGN)
    return k_rows_reduce(part, 1, chunks, c * 9).view(c, 1, 3, 3)


class PackLevels(torch.autograd.Function):
    """stack pyramid levels [N,H_l,W_l,C] into one [1, 1, sum rows, C] tensor (backward hands out views of the packed gradient)"""

    @staticmethod
    def forward(ctx, *feats):
        n, c = feats[0].shape[0], feats[0].shape[3]
        geom = (n, tuple(f.shape[1] for f in feats), tuple(f.shape[2] for f in feats))
        total = packed_rows(geom)
        out = torch.empty((1, 1, total, c), device=feats[0].device, dtype=BF16)
        if PACK_LEVELS_ONE and all(f.stride(3) == 1 and f.stride(1) == f.shape[2] * f.stride(2) and f.stride(0) == f.shape[1] * f.stride(1)
                                   for f in feats):
            nl, H, W, _, _ = _geom_arrays(geom)                   # one launch for all levels
            lds = (ctypes.c_int * nl)(*[ld(f) for f in feats])
            srcs = _ptr_array(feats)                               # (kept alive across the call)
            lib().call("hn_pack_levels", ctypes.addressof(srcs), ctypes.addressof(lds), ptr(out), ld(out), n, c, nl,
                       ctypes.addressof(H), ctypes.addressof(W), LEVEL_ALIGN)
        else:
            for v, f in zip(level_views(out, geom), feats):
                k_eltwise(2, f, alpha=1.0, out=v)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(level_views(dense(g), ctx.geom))


TOWER_BN_IN_GEMM = policy("HN_TOWER_BN_IN_GEMM", "1") != "0"
PACK_LEVELS_ONE = policy("HN_PACK_LEVELS_ONE", "1") != "0"       # PackLevels: one copy launch for all levels (0: one per level)
HEAD_OUT_LEVELS = policy("HN_HEAD_OUT_LEVELS", "1") != "0"       # the heads' output conv of all five levels in one launch (0: one per level)
HEAD_GRAD_LEVELS = policy("HN_HEAD_GRAD_LEVELS", "1") != "0"     # head-output gradient operand of all five levels in one launch (0: one per level)
_EVAL_COEF = {}             # id(gamma of level 0) -> (the 4 * nl BatchNorm tensors, their versions, eps, coef [nl, 4, cout])


def _eval_coef_levels(gam, bet, rms, rvs, eps, cout, coef):
    """eval-mode (running statistics) scale / shift rows of a tower layer's per-level BatchNorms: constants until a parameter or running
    statistic changes in place, so they are computed once (five ~5 us launches per layer and forward otherwise: 30 per deploy forward)"""
    tens = (*gam, *bet, *rms, *rvs)
    ver = (mutation_epoch(gam[0]), *[t._version for t in tens])   # raw-pointer updates (hn_adam_step, training-mode BatchNorm kernels) bump the owner's epoch only
    hit = _EVAL_COEF.get(id(gam[0]))
    if hit is not None and len(hit[0]) == len(tens) and all(a is b for a, b in zip(hit[0], tens)) and hit[1] == ver and hit[2] == eps:
        return hit[3]
    for l in range(len(gam)):
        lib().call("hn_bn_eval_coeff", ptr(gam[l]), ptr(bet[l]), ptr(rms[l]), ptr(rvs[l]), float(eps), cout, ptr(coef[l, 0]), ptr(coef[l, 1]))
    _EVAL_COEF[id(gam[0])] = (tens, ver, eps, coef)
    return coef


class TowerLayer(torch.autograd.Function):
    """out = act(BN_level(pointwise(depthwise(x)) + bias)) on level-packed rows; bn = nlev x (gamma, beta, running_mean, running_var)."""

    @staticmethod
    def forward(ctx, x, dw_w, pw_w, pw_b, geom, act, eps, momentum, training, slot, *bn):
        nl, H, W, R, CNT = _geom_arrays(geom)
        ctx.slot = slot                                     # GradSlot of x (the packed map feeds both towers), see Share
        total, c = x.shape[2], x.shape[3]
        cout = pw_w.shape[0]
        dev = x.device
        wk, wf = pack_dw_weight(dw_w)
        d = k_dwconv_levels(x, wk, geom)
        wp, wt = pack_conv_weight(pw_w)
        gam, bet = [bn[4 * l] for l in range(nl)], [bn[4 * l + 1] for l in range(nl)]
        rms, rvs = [bn[4 * l + 2] for l in range(nl)], [bn[4 * l + 3] for l in range(nl)]
        coef = torch.empty((nl, 4, cout), device=dev, dtype=F32)
        if not training and not torch.is_grad_enabled() and TOWER_BN_IN_GEMM:
            # inference: the per-level BatchNorm (running statistics) + activation ride in the pointwise conv's epilogue -- one launch
            coef = _eval_coef_levels(gam, bet, rms, rvs, eps, cout, coef)
            out = torch.empty((1, 1, total, cout), device=dev, dtype=BF16)
            lib().call("hn_conv_gemm_nt_lvl", ptr(d), ld(d), total, c, ptr(wp), cout, kp32(c), ptr(pw_b), act, ptr(out), ld(out), ptr(coef), nl,
                       ctypes.addressof(R))
            return out
        z, psum, psq = k_gemm_nt(d, None, 0, (1, 1, total), wp, cout, kp32(c), 1, bias=pw_b, stats=training)
        if training:
            div = total // psum.shape[0]
            ga, ba, rma, rva = _ptr_array(gam), _ptr_array(bet), _ptr_array(rms), _ptr_array(rvs)    # keep the host arrays alive
            lib().call("hn_bn_finalize_levels", ptr(psum), ptr(psq), div, cout, nl, ctypes.addressof(R), ctypes.addressof(CNT),
                       ctypes.addressof(ga), ctypes.addressof(ba), ctypes.addressof(rma), ctypes.addressof(rva), float(eps),
                       float(momentum), ptr(pw_b), ptr(coef))
        else:
            coef = _eval_coef_levels(gam, bet, rms, rvs, eps, cout, coef)
        out = torch.empty((1, 1, total, cout), device=dev, dtype=BF16)
        lib().call("hn_bn_act_levels", ptr(z), ld(z), ptr(coef), act, ptr(out), ld(out), cout, nl, ctypes.addressof(R))
        ctx.geom, ctx.act, ctx.training = geom, act, training
        ctx.has_bias = pw_b is not None
        ctx.packs = (wf, wt)
        ctx.queue, ctx.wrefs = (cur_queue() if (training and dw_w.requires_grad and pw_w.requires_grad) else None), (dw_w, pw_w)
        ctx.save_for_backward(x, d, z, coef, pw_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, d, z, coef, pw_w = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        geom, act = ctx.geom, ctx.act
        nl, H, W, R, CNT = _geom_arrays(geom)
        wf, wt = ctx.packs
        dout = dense(dout)
        total, c = x.shape[2], x.shape[3]
        cout = z.shape[3]
        dev = z.device
        r = lib().query("hn_colred_rows", total, 128)
        pr = total // r
        pg = torch.empty((pr, cout), device=dev, dtype=F32)
        pgx = torch.empty((pr, cout), device=dev, dtype=F32)
        lib().call("hn_bn_bwd_reduce_levels", ptr(dout), ld(dout), ptr(z), ld(z), None, 0, ptr(coef), act, cout, r, nl, ctypes.addressof(R),
                   ptr(pg), ptr(pgx))
        red = torch.empty((nl, 2, cout), device=dev, dtype=F32)
        # one owning tensor per parameter gradient: autograd's AccumulateGrad clones views before storing them in .grad
        dgam = [torch.empty((cout,), device=dev, dtype=F32) for _ in range(nl)]
        dbet = [torch.empty((cout,), device=dev, dtype=F32) for _ in range(nl)]
        dga, dba = _ptr_array(dgam), _ptr_array(dbet)
        dbias = torch.empty((cout,), device=dev, dtype=F32) if ctx.has_bias else None    # a bias feeding BatchNorm has zero gradient
        lib().call("hn_bn_bwd_finalize_levels", ptr(pg), ptr(pgx), r, cout, nl, ctypes.addressof(R), ctypes.addressof(CNT),
                   ctypes.addressof(dga), ctypes.addressof(dba), ptr(red), ptr(dbias))
        dz = torch.empty_like(z)
        lib().call("hn_bn_bwd_apply_levels", ptr(dout), ld(dout), ptr(z), ld(z), None, 0, ptr(coef), ptr(red), act, ptr(dz), ld(dz), cout, nl,
                   ctypes.addressof(R))
        dd, _, _ = k_gemm_nt(dz, None, 0, (1, 1, total), wt, c, kp32(cout), 1)
        q_, (dw_ref, pw_ref) = ctx.queue, ctx.wrefs
        if q_ is not None:
            dpw = q_.add_gemm(pw_ref, d, dz, 0, (1, 1, total), c, cout)
        else:
            dpw = k_gemm_tn(d, None, 0, (1, 1, total), dz, cout, kp32(c), 1, c)
        dx = None
        if ctx.needs_input_grad[0] and ctx.slot is not None:
            sl = ctx.slot
            if sl.buf is None:
                sl.buf, ddw = k_dwconv_bwd(dd, x, wf, geom, queue=q_, weight=dw_ref)
            else:
                _, ddw = k_dwconv_bwd(dd, x, wf, geom, into=sl.buf, queue=q_, weight=dw_ref)
        else:
            dx, ddw = k_dwconv_bwd(dd, x, wf, geom, want_dx=ctx.needs_input_grad[0], queue=q_, weight=dw_ref)
        bn_grads = []
        for l in range(nl):
            bn_grads += [dgam[l], dbet[l], None, None]
        return (dx, ddw, dpw, dbias, None, None, None, None, None, None, *bn_grads)


class HeadOutPacked(torch.autograd.Function):
    """HeadOut on a level-packed input: depthwise, pointwise GEMM (its epilogue maps the packed rows into the [N, sum_l H_l*W_l*rep, k]
    concat: hn_conv_gemm_nt_lvlout), gradient gather (hn_head_grad_levels), data gradient and all weight gradients run once for all levels."""

    @staticmethod
    def forward(ctx, dw_weight, pw_weight, bias, k, act, geom, x):
        n, hs, ws = geom
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        wp, wt = pack_conv_weight(pw_weight)
        wk, wf = pack_dw_weight(dw_weight)
        rows_total = sum(h * w for h, w in zip(hs, ws))
        rep = cout // k
        out = torch.empty((n, rows_total * rep, k), device=x.device, dtype=F32)
        ldc, img_stride = cout, rows_total * cout
        mid = k_dwconv_levels(x, wk, geom)
        off = 0
        if HEAD_OUT_LEVELS:                                   # all levels in one launch: the epilogue maps packed rows to the per-image layout
            nl, H, W, _, _ = _geom_arrays(geom)
            lib().call("hn_conv_gemm_nt_lvlout", ptr(mid), ld(mid), mid.shape[2], cin, ptr(wp), cout, kp32(cin), ptr(bias), act, ptr(out), ldc,
                       img_stride, n, nl, ctypes.addressof(H), ctypes.addressof(W), LEVEL_ALIGN)
        else:
            for v, h, w in zip(level_views(mid, geom), hs, ws):
                k_gemm_nt(v, None, 0, (n, h, w), wp, cout, kp32(cin), 1, bias=bias, act=act, out=out.view(-1)[off * ldc:], out_f32=True,
                          ldc=ldc, rpi=h * w, img_stride=img_stride)
                off += h * w
        ctx.meta = (k, act, ldc, img_stride, geom)
        ctx.packs = (wf, wt)
        ctx.queue, ctx.wref = (cur_queue() if dw_weight.requires_grad else None), dw_weight
        ctx.save_for_backward(pw_weight, out if act == ACT_SIGMOID else None, x, mid)
        return out

    @staticmethod
    def backward(ctx, dout):
        k, act, ldc, img_stride, geom = ctx.meta
        n, hs, ws = geom
        pw_weight, yout, x, mid = ctx.saved_tensors
        wf, wt = ctx.packs
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        dout = dout.contiguous()
        dev = dout.device
        ldz = pad8(cout)
        total = x.shape[2]
        # alignment rows of a ragged packing must read as zeros in the bias / weight gradient sums and in the data gradient
        dz = zeros((1, 1, total, ldz), dev, BF16) if has_pad_rows(geom) else torch.empty((1, 1, total, ldz), device=dev, dtype=BF16)
        if HEAD_GRAD_LEVELS:                                  # all levels in one launch
            nl, H, W, _, _ = _geom_arrays(geom)
            lib().call("hn_head_grad_levels", ptr(dout), ptr(yout), img_stride, ldc, cout, ptr(dz), ldz, n, nl, ctypes.addressof(H),
                       ctypes.addressof(W), LEVEL_ALIGN, 1 if act == ACT_SIGMOID else 0)
        else:
            off = 0
            for v, h, w in zip(level_views(dz, geom), hs, ws):
                base = off * ldc
                lib().call("hn_head_grad", ptr(dout.view(-1)[base:]), ptr(yout.view(-1)[base:]) if yout is not None else None, h * w,
                           img_stride, ldc, cout, ptr(v), ldz, n * h * w, 1 if act == ACT_SIGMOID else 0)
                off += h * w
        dpw, dbias = k_gemm_tn(mid, None, 0, (1, 1, total), dz, cout, kp32(cin), 1, cin, want_bias=True)
        dmid, _, _ = k_gemm_nt(dz, None, 0, (1, 1, total), wt, cin, kp32(cout), 1, c0=ldz, c1=0)
        dx, ddw = k_dwconv_bwd(dmid, x, wf, geom, queue=ctx.queue, weight=ctx.wref)
        return ddw, dpw, dbias, None, None, None, dx


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
