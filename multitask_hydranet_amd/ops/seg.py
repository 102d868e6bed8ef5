"""ops.seg -- segmentation decoder blocks: direct 3x3 conv nodes (full-resolution and phase form over up-sampled maps) and the phase-form
output conv (reference: head_seg/segmentation.py:16-105)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib, policy
from .core import *        # noqa: F401,F403
from .backbone import *        # noqa: F401,F403
from .neck import *        # noqa: F401,F403


# --------------------------------------------------------------------------------------------------------------
# segmentation decoder block: y = act(conv3x3(reflect_pad(cat[up2(x0)|x0, x1])) + bias)
# --------------------------------------------------------------------------------------------------------------
SEG_FOLD_DIRECT = policy("HN_SEG_FOLD_DIRECT", "1") != "0"


SEG_FOLD_MIN_ELEMS = 1 << 24   # measured (step trace, N = 16): the folding epilogue wins 130 / 45 / 19 / 11 us on the 134M / 67M / 33M / 17M-element
                               # gradients and loses 2...26 us on the <= 8M-element ones (a few hundred workgroups cannot hide the extra loads)


def dgrad_fold_ok(nout, h, w, n=None):
    """shapes hn_conv3x3_dgrad_fold covers (the staged bf16 epilogue of the 64 / 128-cout tiles); with n: and where it pays"""
    return SEG_FOLD_DIRECT and nout % 8 == 0 and nout > 32 and h >= 4 and w >= 4 and (n is None or n * h * w * nout >= SEG_FOLD_MIN_ELEMS)


def k_dgrad_fold(dz, wt, n, h, w, nout, kp, phase_k, clamp, yprev, s2d=0):
    """dx [N,h,w,nout] = folded data gradient (* ELU'(yprev)): conv with a folding epilogue + border fix-up, no padded-grid tensor.
    s2d = 1: the same values in space-to-depth order instead, [N,h/2,w/2,4*nout] (the operand form of the phase-form block that consumes
    them); s2d = 2: both -> (dx, dx_s2d)"""
    dev = dz.device
    dx = new_act(n, h, w, nout, dev) if s2d != 1 else None
    dxs = new_act(n, h // 2, w // 2, 4 * nout, dev) if s2d else None
    ring = torch.empty((n, lib().query("hn_fold_ring_rows", h, w), nout), device=dev, dtype=BF16)
    yargs = (ptr(yprev), ld(yprev) if yprev is not None else 0, ptr(ring))
    if s2d:
        lib().call("hn_conv3x3_dgrad_fold_s2d", ptr(dz), ld(dz), dz.shape[3], n, h, w, ptr(wt), nout, kp, phase_k, clamp, ptr(dx),
                   ld(dx) if dx is not None else 0, ptr(dxs), ld(dxs), *yargs)
    else:
        lib().call("hn_conv3x3_dgrad_fold", ptr(dz), ld(dz), dz.shape[3], n, h, w, ptr(wt), nout, kp, phase_k, clamp, ptr(dx), ld(dx), *yargs)
    return dxs if s2d == 1 else ((dx, dxs) if s2d == 2 else dx)


def seg_s2d_handover_ok(x0, skip, weight, n_out_classes=None):
    """May the block behind a SegConvUp hand the gradient over in space-to-depth order (SegOutUp(dx_s2d=True) -> SegConvUp(dy_is_s2d=True))?
    x0 / skip / weight: the SegConvUp's operands.  Its backward must need nothing but the space-to-depth form (no skip operand, phase-form
    data gradient) and the producer must take the folding epilogue at the up-sampled shape."""
    n, h, w, c0 = x0.shape
    k = weight.shape[0]
    if skip is not None or weight.shape[1] != c0 or SEG_DGRAD_PHASE is False:
        return False
    tiles = n * ((h + 2 + 15) // 16) * ((w + 2 + 15) // 16) * ((c0 + 127) // 128)
    phase = (tiles >= SEG_DGRAD_PHASE_MIN_TILES) if SEG_DGRAD_PHASE is None else SEG_DGRAD_PHASE
    return bool(SEG_S2D_HANDOVER and phase and dgrad_fold_ok(k, 2 * h, 2 * w, n))


class SegConv(torch.autograd.Function):
    """ConvBlock / Conv3x3 of the seg decoder.  Along the decoder chain every x0 is the ELU output of the previous block and has no other
    consumer, so ELU' of the previous block is applied where this block folds its data gradient (x0_is_elu: hn_seg_fold multiplies by
    ELU'(x0)) and the previous block is told that the gradient it receives is already its dz (dy_is_dz) -- one elementwise pass less per
    block."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, up, act, out_f32, x0_is_elu=False, dy_is_dz=False, dx_s2d_slot=None):
        """dx_s2d_slot (GradSlot shared with the SegConvUp that produced x0): where the backward leaves the gradient w.r.t. x0 a second time,
        in that block's space-to-depth operand order, when its folding epilogue runs (the block then skips its hn_space_to_depth_bf16 pass)"""
        ctx.dx_s2d_slot = dx_s2d_slot
        n, h0, w0, c0 = x0.shape
        h, w = (h0 * 2, w0 * 2) if up else (h0, w0)
        cout, cin = weight.shape[0], weight.shape[1]
        wp, wt = pack_conv_weight(weight)
        y, _, _ = k_gemm_nt(x0, x1, 2, (n, h, w), wp, cout, kp32(cin), 9, bias=bias, act=act, out_f32=out_f32, up=up)
        ctx.up, ctx.act, ctx.out_f32 = up, act, out_f32
        ctx.x0_is_elu, ctx.dy_is_dz = x0_is_elu, dy_is_dz
        ctx.wt = wt
        ctx.has_x1 = x1 is not None
        ctx.save_for_backward(x0, x1, weight, y if (act == ACT_ELU and not dy_is_dz) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight, y = ctx.saved_tensors
        up = ctx.up
        n, h0, w0, c0 = x0.shape
        h, w = (h0 * 2, w0 * 2) if up else (h0, w0)
        cout, cin = weight.shape[0], weight.shape[1]
        c1 = cin - c0
        dev = x0.device
        m = n * h * w
        if ctx.out_f32:                                              # logits gradient from the loss: fp32 [N,H,W,cout] -> padded bf16
            dy = dy.contiguous()
            dz = new_act(n, h, w, pad8(cout), dev)
            lib().call("hn_cast_f32_to_bf16_pad", ptr(dy), cout, ptr(dz), pad8(cout), m, cout)
        else:
            dy = dense(dy)
            dz = k_eltwise(1, dy, y, act=ACT_ELU) if (ctx.act == ACT_ELU and not ctx.dy_is_dz) else dy
        # bias gradient (per-channel sum of dz): out of the weight-gradient launches
        if wgrad_bias_ok(2, kp32(cin)):
            dw, dbias = k_gemm_tn(x0, x1, 2, (n, h, w), dz, cout, kp32(cin), 9, cin, up=up, kh=3, want_bias=True)
        else:
            ps, _, r = k_col_stats(dz)
            dbias = k_rows_reduce(ps, 1, ps.shape[0], dz.shape[3]).view(-1)
            if dbias.numel() != cout:
                dbias = dbias[:cout] + 0.0                           # owning copy by a kernel (a clone would be a memcpy node in the graph)
            dw = k_gemm_tn(x0, x1, 2, (n, h, w), dz, cout, kp32(cin), 9, cin, up=up, kh=3)
        # data gradient on the padded (H+2)x(W+2) grid, then fold the reflection / up-sampling / concat back
        dx0 = dx1 = None
        if not up and not ctx.has_x1 and dgrad_fold_ok(c0, h, w, n):
            if ctx.needs_input_grad[0]:
                both = ctx.dx_s2d_slot is not None and SEG_S2D_HANDOVER and ctx.x0_is_elu and not (h & 1) and not (w & 1)
                dx0 = k_dgrad_fold(dz, ctx.wt, n, h, w, c0, kp32(cout), 0, 0, x0 if ctx.x0_is_elu else None, s2d=2 if both else 0)
                if both:
                    dx0, ctx.dx_s2d_slot.buf = dx0
            return dx0, dx1, dw, dbias, None, None, None, None, None, None
        dvp, _, _ = k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), ctx.wt, cin, kp32(cout), 9, c0=dz.shape[3], c1=0)
        if ctx.needs_input_grad[0]:
            dx0 = new_act(n, h0, w0, c0, dev)
            yp = x0 if ctx.x0_is_elu else None
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c0, up)
        if ctx.has_x1 and ctx.needs_input_grad[1]:
            dx1 = new_act(n, h, w, c1, dev)
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), c0, ptr(dx1), ld(dx1), None, 0, n, h, w, c1, 0)
        return dx0, dx1, dw, dbias, None, None, None, None, None, None



# --------------------------------------------------------------------------------------------------------------
# Final seg conv in phase form.  Conv3x3(reflect-pad(nearest_up2(x))) (head_seg/segmentation.py:101-104) on the up-sampled grid reads
# every low-resolution pixel four times and, for the 5-class output layer, runs its data gradient on a 514x1026x64 padded grid.  On
# the LOW-resolution grid the same function is a 3x3 conv with replicate padding and 4*k outputs (one k-vector per output phase
# (py,px)): W_eff[(py,px,o)][c][dy][dx] = sum of the taps (ky,kx) whose up-sampled source row/col falls on low-res offset (dy,dx)
#   phase 0: ky=0 -> dy=-1, ky=1,2 -> dy=0;   phase 1: ky=0,1 -> dy=0, ky=2 -> dy=+1     (same for kx/dx)
# and reflection of the up-sampled index is exactly clamping of the low-res index.  4x fewer pixels forward, and the data gradient is
# produced directly at the producer's resolution (no full-resolution padded dgrad, fold, 2x2 sum).
# --------------------------------------------------------------------------------------------------------------
_PHASE_T = {}


def _phase_matrix(device):
    """T[(py,px,dy,dx), (ky,kx)] in {0,1}: W_eff.view(k*c, 36) = W.view(k*c, 9) @ T^T"""
    t = _PHASE_T.get(device)
    if t is None:
        a = torch.zeros(2, 3, 3)                       # a[p][d+1][k]
        a[0, 0, 0] = 1; a[0, 1, 1] = 1; a[0, 1, 2] = 1
        a[1, 1, 0] = 1; a[1, 1, 1] = 1; a[1, 2, 2] = 1
        t = torch.einsum("pdk,qel->pqdekl", a, a).reshape(36, 9).to(device)
        _PHASE_T[device] = t
    return t


class SegOutUp(torch.autograd.Function):
    """logits[N, 2h, 2w, k] (fp32) = Conv3x3(ReflectionPad2d(1)(nearest_up2(x))) + bias, x [N, h, w, c] bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, x_is_elu=False, slot=None, dx_s2d=False):
        """slot: GradSlot through which the loss may deliver the gradient already in this node's space-to-depth bf16 operand form.
        dx_s2d (only with seg_s2d_handover_ok for the producer of x): the gradient w.r.t. x is returned in space-to-depth memory order under
        x's shape; the producer is a SegConvUp(dy_is_s2d=True)"""
        ctx.dx_s2d = dx_s2d
        n, h, w, c = x.shape
        k = weight.shape[0]
        ctx.x_is_elu = x_is_elu
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        wp, wt, b_eff = pack_phase_weight(weight, c, bias)
        out = torch.empty((n, 2 * h, 2 * w, k), device=x.device, dtype=F32)
        # img_stride = -k: the conv epilogue scatters phase (py, px) of low-res pixel (y, x) to output pixel (2y+py, 2x+px) itself
        k_gemm_nt(x, None, 4, (n, h, w), wp, 4 * k, kp32(c), 9, bias=b_eff, out=out, out_f32=True, ldc=4 * k, img_stride=-k)
        ctx.wt = wt
        ctx.k = k
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        k = ctx.k
        dev = x.device
        ldz = pad8(4 * k)
        dz = None
        if ctx.slot is not None and ctx.slot.buf is not None:          # delivered by SegLoss.backward in operand form
            dz, ctx.slot.buf = ctx.slot.buf, None
        if dy is not None:
            d2 = new_act(n, h, w, ldz, dev)
            lib().call("hn_space_to_depth", ptr(dy.contiguous()), ptr(d2), ldz, n, h, w, k)
            dz = d2 if dz is None else k_eltwise(0, dz, d2)
        if dz is None:
            return None, None, None, None, None, None
        if wgrad_bias_ok(4, kp32(c)):
            dw_eff, db_eff = k_gemm_tn(x, None, 4, (n, h, w), dz, 4 * k, kp32(c), 9, c, kh=3, want_bias=True)       # [4k, c, 3, 3], [4k]
        else:
            ps, _, _ = k_col_stats(dz)
            db_eff = k_rows_reduce(ps, 1, ps.shape[0], ldz)
            dw_eff = k_gemm_tn(x, None, 4, (n, h, w), dz, 4 * k, kp32(c), 9, c, kh=3)
        dw = torch.empty((k, c, 3, 3), device=dev, dtype=F32)
        dbias = torch.empty((k,), device=dev, dtype=F32)
        lib().call("hn_phase_fold", ptr(dw_eff), None, ptr(db_eff), ptr(dw), ptr(dbias), k, c, 0)
        dx = None
        if ctx.needs_input_grad[0]:
            yp = x if ctx.x_is_elu else None
            if dgrad_fold_ok(c, h, w, n):
                # (dx_s2d: space-to-depth memory order under x's shape -- autograd checks the shape, the consuming SegConvUp knows the order)
                dx = k_dgrad_fold(dz, ctx.wt, n, h, w, c, kp32(4 * k), 0, 1, yp, s2d=1 if ctx.dx_s2d else 0)
                if ctx.dx_s2d:
                    dx = dx.view(n, h, w, c)
            else:
                assert not ctx.dx_s2d, "SegOutUp(dx_s2d=True) without seg_s2d_handover_ok"
                dvp, _, _ = k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), ctx.wt, c, kp32(4 * k), 9, c0=ldz, c1=0)
                dx = new_act(n, h, w, c, dev)
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx), ld(dx), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c, 2)
        return dx, dw, dbias, None, None, None


def seg_out_argmax(x, weight, bias):
    """deploy forward of the seg output layer fused with the arg-max over the classes (model/model.py:197): int64 mask [N, 2h, 2w]; the fp32
    logits are never written.  No gradient (inference only)."""
    n, h, w, c = x.shape
    k = weight.shape[0]
    wp, _, b_eff = pack_phase_weight(weight, c, bias)
    mask = torch.empty((n, 2 * h, 2 * w), device=x.device, dtype=torch.int64)
    lib().call("hn_conv3x3_out_argmax", ptr(x), n, h, w, c, ld(x), ptr(wp), k, kp32(c), ptr(b_eff), ptr(mask))
    return mask


def seg_out_argmax_ok(x, weight):
    k, c = weight.shape[0], x.shape[3]
    return x.is_cuda and not torch.is_grad_enabled() and 4 * k <= 32 and kp32(c) == 64 and (x.shape[2] * k) % 2 == 0


class SegConvUp(torch.autograd.Function):
    """y [N, 2h, 2w, k] = ELU(Conv3x3(ReflectionPad2d(1)(cat[nearest_up2(x0), x1])) + bias) in PHASE form (decoder blocks 1/3/5/7,
    head_seg/segmentation.py:92-100).  The up-sampled operand is convolved on its own low-resolution grid with the effective weights of
    SegOutUp (4 output phases, 2x2 non-zero taps each: 16 instead of 36 tap products per low-res pixel -- 2.25x fewer MACs in forward,
    data gradient and weight gradient); the skip operand x1 (already full resolution) goes through the ordinary direct 3x3 kernel and
    joins as a pre-activation addend in the phase conv's epilogue.  The data gradient w.r.t. x0 is produced directly at x0's resolution
    (no full-resolution padded grid, no 2x2 fold).  ELU' folding along the decoder chain as in SegConv (x0_is_elu / dy_is_dz)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, x0_is_elu=False, dy_is_dz=False, dy_is_s2d=False, dzs_slot=None):
        """dy_is_s2d (with dy_is_dz, only where seg_s2d_handover_ok): the incoming gradient is this block's dz already in space-to-depth memory
        order (written so by the consumer's folding data-gradient epilogue, SegOutUp(dx_s2d=True)).
        dzs_slot (with dy_is_dz): GradSlot in which the consumer (SegConv(dx_s2d_slot=...)) may leave dz a second time in that order"""
        ctx.dzs_slot = dzs_slot if dy_is_dz else None
        n, h, w, c0 = x0.shape
        k, cin = weight.shape[0], weight.shape[1]
        c1 = cin - c0
        dev = x0.device
        z1 = wt1 = wt_full = None
        ctx.dy_is_s2d = dy_is_s2d
        # Per-layer choice of form (measured, tools/bench_seg.py): the forward runs full-resolution when the skip operand is so narrow that its
        # own conv would be mostly K padding (decoder.5: 24 channels); the data gradient w.r.t. x0 runs full-resolution when the padded
        # low-resolution grid cannot fill the chip (decoder.1: 18x34 cells -> 384 workgroups); the weight gradient is always in phase form.
        fwd_phase = (c1 == 0 or c1 >= 32) if SEG_FWD_PHASE is None else SEG_FWD_PHASE
        tiles = n * ((h + 2 + 15) // 16) * ((w + 2 + 15) // 16) * ((c0 + 127) // 128)
        ctx.dgrad_phase = (tiles >= SEG_DGRAD_PHASE_MIN_TILES) if SEG_DGRAD_PHASE is None else SEG_DGRAD_PHASE
        wp_eff, wt_eff, b_eff = pack_phase_weight(weight, c0, bias, want_wt=ctx.dgrad_phase)
        if c1 and (fwd_phase or ctx.dgrad_phase):
            wp1, wt1 = pack_conv_weight_slice(weight, c0, c1)
        if not fwd_phase or not ctx.dgrad_phase:
            wp_full, wt_full = pack_conv_weight(weight)
        if fwd_phase:
            if c1:
                z1, _, _ = k_gemm_nt(x1, None, 2, (n, 2 * h, 2 * w), wp1, k, kp32(c1), 9)      # skip operand: plain reflect-pad 3x3, no bias / act
            y = new_act(n, 2 * h, 2 * w, k, dev)
            lib().call("hn_conv3x3_phase", ptr(x0), 4, n, h, w, c0, ld(x0), ptr(wp_eff), 4 * k, kp32(c0), ptr(b_eff), ACT_ELU, ptr(y), ld(y), k,
                       ptr(z1), ld(z1) if z1 is not None else 0)
        else:
            y, _, _ = k_gemm_nt(x0, x1, 2, (n, 2 * h, 2 * w), wp_full, k, kp32(cin), 9, bias=bias, act=ACT_ELU, up=1)
        ctx.x0_is_elu, ctx.dy_is_dz = x0_is_elu, dy_is_dz
        assert not dy_is_s2d or (dy_is_dz and c1 == 0 and ctx.dgrad_phase), "SegConvUp(dy_is_s2d=True) without seg_s2d_handover_ok"
        ctx.packs = (wt_eff, wt1, wt_full)
        ctx.save_for_backward(x0, x1, y if not dy_is_dz else None)
        ctx.dims = (k, c0, c1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, y = ctx.saved_tensors
        wt_eff, wt1, wt_full = ctx.packs
        k, c0, c1 = ctx.dims
        n, h, w, _ = x0.shape
        dev = x0.device
        dy = dense(dy)
        dz = dy if ctx.dy_is_dz else k_eltwise(1, dy, y, act=ACT_ELU)
        # space-to-depth gradient: the operand of both low-resolution contractions
        if ctx.dy_is_s2d:
            dzs, dz = dy.view(n, h, w, 4 * k), None                  # handed over in that order (nothing below reads the plain form)
        elif ctx.dzs_slot is not None and ctx.dzs_slot.buf is not None:
            dzs, ctx.dzs_slot.buf = ctx.dzs_slot.buf, None           # written beside the plain form by the consumer's folding epilogue
        else:
            dzs = new_act(n, h, w, 4 * k, dev)
            lib().call("hn_space_to_depth_bf16", ptr(dz), ld(dz), ptr(dzs), n, h, w, k, None)
        # effective-weight gradient (zeros at the taps a phase does not use), mapped back to the 3x3 weights by the phase matrix; the bias
        # gradient (channel sums of dz, per phase) comes out of the same launches
        splits, rps, wsb = ctypes.c_int(), ctypes.c_long(), ctypes.c_long()
        lib().query("hn_wgrad_plan_phase", n, h, w, 4 * k, kp32(c0), k, ctypes.addressof(splits), ctypes.addressof(rps), ctypes.addressof(wsb))
        ws = torch.empty((wsb.value // 4,), device=dev, dtype=F32)
        dw_eff = torch.empty((4 * k, c0, 3, 3), device=dev, dtype=F32)
        db_eff = torch.empty((4 * k,), device=dev, dtype=F32)
        lib().call("hn_conv_gemm_tn_phase", ptr(x0), n, h, w, c0, ld(x0), ptr(dzs), ld(dzs), 4 * k, kp32(c0), k, ptr(ws), ptr(dw_eff), ptr(db_eff))
        dx0 = dx1 = None
        if not ctx.dgrad_phase:
            # full-resolution data gradient for both operands at once (padded (2h+2) x (2w+2) grid), folded back per operand
            dvp, _, _ = k_gemm_nt(dz, None, 3, (n, 2 * h + 2, 2 * w + 2), wt_full, c0 + c1, kp32(k), 9, c0=k, c1=0)
            if ctx.needs_input_grad[0]:
                dx0 = new_act(n, h, w, c0, dev)
                yp = x0 if ctx.x0_is_elu else None
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, 2 * h, 2 * w, c0, 1)
            if c1 and ctx.needs_input_grad[1]:
                dx1 = new_act(n, 2 * h, 2 * w, c1, dev)
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), c0, ptr(dx1), ld(dx1), None, 0, n, 2 * h, 2 * w, c1, 0)
        elif ctx.needs_input_grad[0] and dgrad_fold_ok(c0, h, w, n):
            dx0 = k_dgrad_fold(dzs, wt_eff, n, h, w, c0, kp32(4 * k), k, 1, x0 if ctx.x0_is_elu else None)
        elif ctx.needs_input_grad[0]:
            dvp = new_act(n, h + 2, w + 2, c0, dev)
            lib().call("hn_conv3x3_phase", ptr(dzs), 3, n, h + 2, w + 2, 4 * k, ld(dzs), ptr(wt_eff), c0, kp32(4 * k), None, ACT_NONE, ptr(dvp),
                       ld(dvp), k, None, 0)
            dx0 = new_act(n, h, w, c0, dev)
            yp = x0 if ctx.x0_is_elu else None
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c0, 2)
        dw1 = None
        if c1:
            dw1 = k_gemm_tn(x1, None, 2, (n, 2 * h, 2 * w), dz, k, kp32(c1), 9, c1, kh=3)
            if ctx.dgrad_phase and ctx.needs_input_grad[1]:
                dvp1, _, _ = k_gemm_nt(dz, None, 3, (n, 2 * h + 2, 2 * w + 2), wt1, c1, kp32(k), 9, c0=k, c1=0)
                dx1 = new_act(n, 2 * h, 2 * w, c1, dev)
                lib().call("hn_seg_fold", ptr(dvp1), ld(dvp1), 0, ptr(dx1), ld(dx1), None, 0, n, 2 * h, 2 * w, c1, 0)
        # effective-weight gradient mapped back to the 3x3 taps (the transpose of the phase map), joined with the skip operand's part
        dw = torch.empty((k, c0 + c1, 3, 3), device=dev, dtype=F32)
        dbias = torch.empty((k,), device=dev, dtype=F32)
        lib().call("hn_phase_fold", ptr(dw_eff), ptr(dw1), ptr(db_eff), ptr(dw), ptr(dbias), k, c0, c1)
        return dx0, dx1, dw, dbias, None, None, None, None


SEG_PHASE_UP = policy("HN_SEG_PHASE_UP", "1") != "0"
SEG_DGRAD_PHASE_MIN_TILES = int(policy("HN_SEG_DGRAD_PHASE_MIN_TILES", "448"))
SEG_S2D_HANDOVER = policy("HN_SEG_S2D_HANDOVER", "1") != "0"   # the output conv's data gradient written in the last block's operand order
SEG_FWD_PHASE = None        # None: per-layer heuristic; True / False force the forward form (tests)
SEG_DGRAD_PHASE = None      # the same for the data gradient w.r.t. the up-sampled operand


def seg_up_phase_ok(x0, x1, weight):
    """phase form needs 64-aligned output channels (a cout tile / K chunk must lie inside one phase)"""
    return SEG_PHASE_UP and x0.is_cuda and weight.shape[0] % 64 == 0 and x0.shape[3] % 8 == 0


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
