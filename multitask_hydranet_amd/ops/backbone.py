"""ops.backbone -- conv + BatchNorm + activation nodes, the one-node XBlock (stride 1 and stride 2), the folded-BatchNorm inference
operators and Squeeze-and-Excitation (reference: net/anynet.py:8-76)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib, policy
from .core import *        # noqa: F401,F403


# --------------------------------------------------------------------------------------------------------------
# conv (1x1 | 1x1 stride 2 | grouped 3x3 | stem) + BatchNorm + activation (+ residual)
# --------------------------------------------------------------------------------------------------------------
class ConvBnAct(torch.autograd.Function):
    """out = act(BN(conv(x) [+ conv_bias]) [+ res]).  kind: "1x1", "g3x3", "stem"."""

    @staticmethod
    def forward(ctx, x, weight, conv_bias, gamma, beta, rm, rv, nbt, res, kind, stride, act, eps, momentum, training, slot=None):
        dev = x.device
        cout = weight.shape[0]
        ctx.slot = slot if (kind == "1x1" and stride == 1) else None     # GradSlot of x (see Share): dgrad accumulates in its GEMM epilogue
        if kind == "stem":
            n, _, hi, wi = x.shape
            ho, wo = hi // 2, wi // 2
            z = new_act(n, ho, wo, 32, dev)
            packs = new_act(n, ho, wo, 32, dev) if training else None       # bf16 im2col rows for the MFMA weight gradient
            lib().call("hn_stem_fwd", ptr(x), ptr(weight), ptr(z), ptr(packs), n, hi, wi)
            psum = psq = None
        elif kind == "g3x3":
            n, hi, wi, c = x.shape
            ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
            if stride == 1 and GCONV_MFMA:          # block-diagonal 64-channel tiles on MFMA (hn_conv_gemm_nt mode 5)
                packs = pack_gconv_diag(weight)
                z, _, _ = k_gemm_nt(x, None, 5, (n, ho, wo), packs[0], c, 64, 9)
            else:
                packs = pack_gconv_weight(weight, 1 if stride == 1 else 0)
                z = new_act(n, ho, wo, c, dev)
                # (stride 2 contracts with packed bf16 dots over the input channels: the pack with i contiguous = packs[1], hydranet_hip.h)
                lib().call("hn_gconv_fwd", ptr(x), ld(x), ptr(packs[0] if stride == 1 else packs[1]), ptr(z), ld(z), n, hi, wi, c, stride)
            psum = psq = None
        else:
            n, hi, wi, cin = x.shape
            ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
            packs = pack_conv_weight(weight)
            z, psum, psq = k_gemm_nt(x, None, 0 if stride == 1 else 1, (n, ho, wo), packs[0], cout, kp32(cin), 1, bias=conv_bias,
                                     stats=training)
        count = n * ho * wo
        if FUSED_BN:
            if training and psum is None:
                psum, psq = k_col_stats_fused(z)
            out, coef, _, _ = k_bn_apply_fused(z, psum, psq, count, gamma, beta, eps, momentum, rm, rv, act, res=res, training=training)
            if training and nbt is not None:
                nbt.add_(1)
        else:
            if training:
                if psum is None:
                    psum, psq, _ = k_col_stats(z)
                coef = k_bn_finalize(psum, psq, count, gamma, beta, eps, momentum, rm, rv)
                if nbt is not None:
                    nbt.add_(1)
            else:
                coef = k_bn_eval_coeff(gamma, beta, rm, rv, eps)
            out = k_bn_act(z, coef, act, res=res)
        ctx.kind, ctx.stride, ctx.act, ctx.count = kind, stride, act, count
        ctx.queue = cur_queue() if (kind == "1x1" and training and weight.requires_grad) else None
        ctx.wref = weight
        ctx.has_bias = conv_bias is not None
        ctx.has_res = res is not None
        ctx.training = training
        ctx.packs = packs
        # ReLU mask: recomputed from z in backward (scale*z+shift > 0); the saved output is only needed when a residual went into the ReLU
        y_save = out if (act == ACT_RELU and (res is not None or not FUSED_BN)) else None
        ctx.save_for_backward(x, weight, z, coef, y_save, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, z, coef, y, gamma = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        dout = dense(dout)
        kind, stride = ctx.kind, ctx.stride
        # a conv bias that feeds BatchNorm has zero gradient: the zeros come out of the BN backward launch (no fill kernel)
        dbias = torch.empty((z.shape[3],), device=z.device, dtype=F32) if ctx.has_bias else None
        if FUSED_BN:
            dz, dgamma, dbeta, g = bn_backward_fused(dout, z, y, coef, ctx.act, ctx.count, want_g=ctx.has_res and ctx.act != ACT_NONE,
                                                     zero_c=dbias)
        else:
            dz, dgamma, dbeta, g = bn_backward(dout, z, y, coef, ctx.act, ctx.count, want_g=ctx.has_res and ctx.act != ACT_NONE)
            if dbias is not None:
                dbias = zeros((z.shape[3],), z.device)
        dres = None
        if ctx.has_res:
            dres = g if g is not None else dout
        dev = z.device
        n, ho, wo, cout = z.shape
        dx = None
        if kind == "stem":
            dw = k_gemm_tn(ctx.packs, None, 0, (n, ho, wo), dz, 32, 32, 1, 32)[:, :27].reshape(32, 3, 3, 3)
        elif kind == "g3x3":
            _, hi, wi, c = x.shape
            wk, wd = ctx.packs
            mfma = stride == 1 and GCONV_MFMA
            if ctx.needs_input_grad[0]:
                if mfma:
                    dx, _, _ = k_gemm_nt(dz, None, 5, (n, hi, wi), wd, c, 64, 9)
                else:
                    dx = new_act(n, hi, wi, c, dev)
                    if stride == 1:
                        lib().call("hn_gconv_fwd", ptr(dz), ld(dz), ptr(wd), ptr(dx), ld(dx), n, hi, wi, c, 1)
                    else:
                        lib().call("hn_gconv_dgrad_s2", ptr(dz), ld(dz), ptr(wk), ptr(dx), ld(dx), n, hi, wi, c)
            if mfma:
                dw = k_gemm_tn(x, None, 5, (n, ho, wo), dz, c, 64, 9, 8, kh=3)
            else:
                chunks = lib().query("hn_wgrad_chunks", n * ho * wo, (c // 8) * 9)
                part = torch.empty((chunks, c * 72), device=dev, dtype=F32)
                lib().call("hn_gconv_wgrad", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), n, hi, wi, c, stride)
                dw = k_rows_reduce(part, 1, chunks, c * 72).view(c, 8, 3, 3)
        else:
            _, hi, wi, cin = x.shape
            wp, wt = ctx.packs
            if ctx.needs_input_grad[0] and ctx.slot is not None:
                # x has other consumers: the first one to run stores its data gradient, the others add theirs in the GEMM epilogue (in place)
                sl = ctx.slot
                if sl.buf is None:
                    sl.buf, _, _ = k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1)
                else:
                    k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1, addend=sl.buf, out=sl.buf)
            elif ctx.needs_input_grad[0]:
                dxs, _, _ = k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1)
                if stride == 1:
                    dx = dxs
                else:
                    dx = zeros((n, hi, wi, cin), dev, BF16)
                    lib().call("hn_add_strided2", ptr(dx), ld(dx), ptr(dxs), ld(dxs), n, ho, wo, cin)
            if ctx.queue is not None:
                dw = ctx.queue.add_gemm(ctx.wref, x, dz, 0 if stride == 1 else 1, (n, ho, wo), cin, cout)
            else:
                dw = k_gemm_tn(x, None, 0 if stride == 1 else 1, (n, ho, wo), dz, cout, kp32(cin), 1, cin)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, dres, None, None, None, None, None, None, None


def conv_bn_act(x, weight, conv_bias, bn, res=None, kind="1x1", stride=1, act=ACT_NONE, eps=1e-5, momentum=0.1, training=True, slot=None):
    gamma, beta, rm, rv, nbt = bn
    return ConvBnAct.apply(x, weight, conv_bias, gamma, beta, rm, rv, nbt, res, kind, stride, act, eps, momentum, training, slot)


# --------------------------------------------------------------------------------------------------------------
# Stride-1 identity XBlock as ONE autograd node (net/anynet.py:65-76; 25 of the 30 blocks of the big backbone):
#   z1 = conv1x1(x); a = relu(bn1(z1)); z2 = gconv3x3(a); b = relu(bn2(z2)); gate = SE(avgpool(b)); z3 = conv1x1(b * gate);
#   out = relu(bn3(z3) + x)
# Forward is 8 launches: the two 1x1 GEMMs and the grouped conv emit their BatchNorm partial statistics; BN1 apply and the final
# BN3 + residual + ReLU are materialising passes; BN2's output is never stored on its own -- one pass over z2 finalizes its statistics
# and produces the SE squeeze, the first excitation layer is a launch of its own, the second one is the prologue of the pass that writes
# relu(bn2(z2)) * gate, the operand of conv_block_3 (hn_se_gate_apply).  Backward is 21 launches
# (BN backward = reduce + apply with the finalize in the prologue; the SE gate gradient and the gated wgrad operand come out of one pass
# over (dbg, z2); the SE data-path backward is folded into the BN2 reduce/apply pair; the residual gradient is added in conv_block_1's
# dgrad epilogue).  The unfused composition of ConvBnAct / SEGate nodes is ~16 + ~27 launches per block.
# --------------------------------------------------------------------------------------------------------------
FUSED_XBLOCK = policy("HN_FUSED_XBLOCK", "1") != "0"
EPILOGUE_STATS = policy("HN_EPILOGUE_STATS", "1") != "0"   # backward reduce passes folded into their producers' epilogues
XBLOCK_XF_GEMM = policy("HN_XBLOCK_XF", "0") == "1"
SE_GATE_IN_APPLY = policy("HN_SE_GATE_APPLY", "1") != "0"    # second SE layer in the prologue of the gated apply pass (hn_se_gate_apply)
SE_GATE_APPLY_MAX = 1 << 24   # ... up to this many activation elements: a bandwidth-bound pass (the 32 x 1152 x 1920 inference maps of stages
                              # 2 / 3: 84 MB each way) runs faster on hn_bn_apply_fused's wider row pieces (tools/bench_gate_apply.py: 35.7 vs 39.2 us)


def se_gate_in_apply(m, c):
    return SE_GATE_IN_APPLY and m * c <= SE_GATE_APPLY_MAX


BN3_PARTS_FROM_DGRAD = policy("HN_BN3_PARTS_FROM_DGRAD", "1") != "0"   # the next block's last backward GEMM makes the BatchNorm-3 backward's partial sums


class XBlockFn(torch.autograd.Function):
    """stride 1 without shortcut: the identity blocks; stride 2 (or a channel change) with the projection shortcut conv + BN
    (ws, gs, bs, rms, rvs): the first block of every stage.  There the grouped conv runs on the stride-2 stencil kernels, and the data
    gradient of the shortcut (a stride-2 row gather) joins conv_block_1's data gradient in that GEMM's epilogue (add_s2): x has ONE
    consumer node, no zero-filled full-resolution tensor, no separate additions."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, sw1, sb1, sw2, sb2, w3, g3, b3, rm3, rv3, eps, momentum, training,
                stride=1, ws=None, gs=None, bs=None, rms=None, rvs=None, group=None):
        """group (WgradGroup of the stage, or None): the 1x1 weight gradients are queued there instead of being launched here"""
        n, h, w, cin = x.shape
        c = w1.shape[0]
        ho, wo = h // stride, w // stride
        m_in, m, hw = n * h * w, n * ho * wo, ho * wo
        cs = sw1.shape[0]
        dev = x.device
        grid = (n, ho, wo)
        wp1, wt1 = pack_conv_weight(w1)
        z1, ps, pq = k_gemm_nt(x, None, 0, (n, h, w), wp1, c, kp32(cin), 1, stats=training)
        a, coef1, _, _ = k_bn_apply_fused(z1, ps, pq, m_in, g1, b1, eps, momentum, rm1, rv1, ACT_RELU, training=training)
        if stride == 1:
            wk2, wd2 = pack_gconv_diag(w2)
            z2, ps, pq = k_gemm_nt(a, None, 5, grid, wk2, c, 64, 9, stats=training)
        else:
            # stride 2: packed-bf16-dot kernels, contraction index contiguous (hydranet_hip.h): the forward takes the (o, i)-swapped pack,
            # the data gradient the plain one
            wd2, wk2 = pack_gconv_weight(w2, 0)
            z2 = new_act(n, ho, wo, c, dev)
            lib().call("hn_gconv_fwd", ptr(a), ld(a), ptr(wk2), ptr(z2), ld(z2), n, h, w, c, stride)
            ps, pq = k_col_stats_fused(z2) if training else (None, None)
        _, coef2, pool, rb = k_bn_apply_fused(z2, ps, pq, m, g2, b2, eps, momentum, rm2, rv2, ACT_RELU, want_out=False, pool_align=hw,
                                              training=training)
        pooled = torch.empty((n, c), device=dev, dtype=F32)
        hid = torch.empty((n, cs), device=dev, dtype=F32)
        gate = torch.empty((n, c), device=dev, dtype=F32)
        fc2_in_apply = se_gate_in_apply(m, c) and not XBLOCK_XF_GEMM
        lib().call("hn_se_mlp_fwd_parts", ptr(pool), hw // rb, 1.0 / hw, ptr(sw1), ptr(sb1), ptr(sw2), ptr(sb2), ptr(pooled), ptr(hid),
                   None if fc2_in_apply else ptr(gate), n, c, cs)
        wp3, wt3 = pack_conv_weight(w3)
        if XBLOCK_XF_GEMM:      # BN2 + ReLU + gate in conv_block_3's operand loader (register-staged: measured 7-10 us slower per launch
            bg = None           # than the LDS-DMA loader, more than the extra pass below costs)
            z3, ps, pq = k_gemm_nt(z2, None, 0, grid, wp3, c, kp32(c), 1, stats=training, xform=(coef2[0], coef2[1], gate, hw, ACT_RELU))
        else:                   # second pass over z2: bg = relu(bn2(z2)) * gate, kept for conv_block_3's weight gradient
            if fc2_in_apply:    # ... with the second excitation layer in its prologue (one launch less per block)
                bg = new_act(n, ho, wo, c, dev)
                lib().call("hn_se_gate_apply", ptr(z2), ld(z2), ptr(coef2), ACT_RELU, ptr(hid), ptr(sw2), ptr(sb2), ptr(gate), ptr(bg), ld(bg),
                           n, hw, c, cs)
            else:
                bg, _, _, _ = k_bn_apply_fused(z2, None, None, m, g2, b2, eps, momentum, None, None, ACT_RELU, coef=coef2, gate=gate, hw=hw)
            z3, ps, pq = k_gemm_nt(bg, None, 0, grid, wp3, c, kp32(c), 1, stats=training)
        zs = coefs = wts = None
        res = x
        if ws is not None:      # projection shortcut: 1x1 conv (stride-2 row gather) + BatchNorm, no activation
            wps, wts = pack_conv_weight(ws)
            zs, pss, pqs = k_gemm_nt(x, None, 0 if stride == 1 else 1, grid, wps, c, kp32(cin), 1, stats=training)
            res, coefs, _, _ = k_bn_apply_fused(zs, pss, pqs, m, gs, bs, eps, momentum, rms, rvs, ACT_NONE, training=training)
        out, coef3, _, _ = k_bn_apply_fused(z3, ps, pq, m, g3, b3, eps, momentum, rm3, rv3, ACT_RELU, res=res, training=training)
        ctx.training, ctx.stride = training, stride
        ctx.packs = (wt1, wd2, wt3, wts)
        ctx.group = group
        # BN3_PARTS_FROM_DGRAD: an identity block whose input IS the previous block's output lets its last backward GEMM (dx = dz1 W1 + g)
        # make the reduce pass of that block's BatchNorm-3 backward (k_gemm_nt estat 3); the stage's group carries the hand-over
        z3p = coef3p = None
        if group is not None and training:
            last = getattr(group, "bn3_last", None)
            if (BN3_PARTS_FROM_DGRAD and EPILOGUE_STATS and last is not None and ws is None and stride == 1 and last[0] == x.data_ptr()
                    and last[1] == tuple(x.shape) and x.is_contiguous() and cin > 64
                    and lib().query("hn_nt_stat_rows", m_in, cin) == (m_in + 63) // 64 <= MAX_PROLOGUE_ROWS):
                _, _, z3p, coef3p = last
            # (the output's address, not the output: the group is reachable from the output's grad_fn, and a reference cycle through it would
            # keep the whole stage's activations alive until a garbage collection -- inside a graph capture, past the capture)
            group.bn3_last = (out.data_ptr(), tuple(out.shape), z3, coef3) if BN3_PARTS_FROM_DGRAD else None
        ctx.wrefs = (w1, w3, ws, w2, sw1, sb1, sw2, sb2)   # identities under which the stage's DeferredGrads node returns the gradients
        ctx.save_for_backward(x, z1, a, z2, z3, out, coef1, coef2, coef3, pooled, hid, gate, sw1, sw2, bg, zs, coefs, z3p, coef3p)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z1, a, z2, z3, out, coef1, coef2, coef3, pooled, hid, gate, sw1, sw2, bg, zs, coefs, z3p, coef3p = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        wt1, wd2, wt3, wts = ctx.packs
        stride = ctx.stride
        dout = dense(dout)
        n, h, w, cin = x.shape
        _, ho, wo, c = z2.shape
        m_in, m, hw = n * h * w, n * ho * wo, ho * wo
        cs = sw1.shape[0]
        dev = x.device
        grid = (n, ho, wo)
        # out = relu(bn3(z3) + shortcut): g = dout * [out > 0] is also the gradient of the shortcut branch
        parts3 = None
        handed = getattr(ctx.group, "bn3_parts", None) if ctx.group is not None else None
        if handed is not None:                                # (dx, pg, pgx) of the next block's last GEMM: valid iff that dx IS this dout
            ctx.group.bn3_parts = None
            if handed[0].data_ptr() == dout.data_ptr() and handed[0].shape == dout.shape:
                parts3 = handed[1:]
        dz3, dg3, db3, g = bn_backward_fused(dout, z3, out, coef3, ACT_RELU, m, want_g=True, parts=parts3)
        make_bg = bg is None
        # SE gate-gradient partials sum_rows dbg * relu(bn2(z2)) from the GEMM's own epilogue (one partial row per pixel tile) where a
        # tile lies inside one image and an image has few tiles (the deep stages); otherwise by a pass over (dbg, z2) below
        bp = lib().query("hn_nt_stat_tile", m, c)       # (not m // rows: a ragged last tile made that 60 for 8 x 60 rows in 64-row tiles)
        ep_dot = EPILOGUE_STATS and not make_bg and hw % bp == 0 and hw // bp <= 16
        dbg, pdot, _ = k_gemm_nt(dz3, None, 0, grid, wt3, c, kp32(c), 1, estat=(1, z2, coef2) if ep_dot else None)
        group = ctx.group                                     # WgradGroup: the 1x1 weight gradients wait for the stage boundary
        w1_, w3_, ws_, w2_, sw1_, sb1_, sw2_, sb2_ = ctx.wrefs
        batch = WgradBatch()                                  # the slab reduces of dw3 / dw2 / dw1 / dws: one launch at the end
        if not make_bg:                                       # dz3's second reader right behind the first: still in the XCDs' L2s
            dw3 = group.add(w3_, bg, dz3, 0, grid, c, c) if group is not None else k_gemm_tn(bg, None, 0, grid, dz3, c, kp32(c), 1, c, defer=batch)
        # one pass over (dbg, z2): gate-gradient partials and the gated operand bg = relu(bn2(z2)) * gate of conv_block_3's wgrad
        rb = bp
        if not ep_dot:
            rb = lib().query("hn_fused_row_block", m, c, hw, 0, 1)
            if make_bg:
                bg = new_act(n, ho, wo, c, dev)
            pdot = torch.empty(((m + rb - 1) // rb, c), device=dev, dtype=F32)
            lib().call("hn_se_bwd_reduce_fused", ptr(dbg), ld(dbg), ptr(z2), ld(z2), ptr(coef2), ptr(gate), hw, ptr(bg) if make_bg else None,
                       ld(bg), ptr(pdot), m, c, rb)
        if make_bg:
            dw3 = group.add(w3_, bg, dz3, 0, grid, c, c) if group is not None else k_gemm_tn(bg, None, 0, grid, dz3, c, kp32(c), 1, c, defer=batch)
        dpre2 = torch.empty((n, c), device=dev, dtype=F32)
        dpool = torch.empty((n, c), device=dev, dtype=F32)
        dpre1 = torch.empty((n, cs), device=dev, dtype=F32)
        if group is not None:                                  # the two outer products wait for the stage boundary (hn_grad_tail)
            dsw1 = dsb1 = dsw2 = dsb2 = None
            group.add_outer(sw2_, sb2_, dpre2, hid)
            group.add_outer(sw1_, sb1_, dpre1, pooled)
        else:
            dsw1, dsb1 = torch.empty_like(sw1), torch.empty((cs,), device=dev, dtype=F32)
            dsw2, dsb2 = torch.empty_like(sw2), torch.empty((c,), device=dev, dtype=F32)
        lib().call("hn_se_mlp_bwd_parts", ptr(pdot), hw // rb, ptr(gate), ptr(hid), ptr(pooled), ptr(sw1), ptr(sw2), ptr(dpre2), ptr(dpre1),
                   ptr(dpool), ptr(dsw1), ptr(dsb1), ptr(dsw2), ptr(dsb2), n, c, cs)
        # BN2 backward with the SE data path folded in: g2 = (dbg * gate + dpool / HW) * [bn2(z2) > 0]
        dz2, dg2, db2, _ = bn_backward_fused(dbg, z2, None, coef2, ACT_RELU, m, gate=gate, dpool=dpool, hw=hw)
        parts1 = None
        if stride == 1:
            # BatchNorm-1 backward partial sums (sum g, sum g * xhat over (da, z1)) from the data-gradient conv's epilogue: one row per
            # 16 x 16 patch, folded by the apply pass's prologue
            ep_bn = EPILOGUE_STATS and lib().query("hn_direct_stat_rows", n, ho, wo) <= MAX_PROLOGUE_ROWS
            da, pg1, pgx1 = k_gemm_nt(dz2, None, 5, grid, wd2, c, 64, 9, estat=(2, z1, coef1) if ep_bn else None)
            if ep_bn:
                parts1 = (pg1, pgx1)
            if group is not None:
                dw2 = group.add_gconv(w2_, a, dz2, grid, c)
            else:
                dw2 = k_gemm_tn(a, None, 5, grid, dz2, c, 64, 9, 8, kh=3, defer=batch)
        else:
            da = new_act(n, h, w, c, dev)
            lib().call("hn_gconv_dgrad_s2", ptr(dz2), ld(dz2), ptr(wd2), ptr(da), ld(da), n, h, w, c)
            chunks = lib().query("hn_wgrad_chunks", m, (c // 8) * 9)
            part = torch.empty((chunks, c * 72), device=dev, dtype=F32)
            lib().call("hn_gconv_wgrad", ptr(a), ld(a), ptr(dz2), ld(dz2), ptr(part), n, h, w, c, stride)
            if group is not None:
                dw2 = group.add_rows(w2_, part, chunks, c * 72, (c, 8, 3, 3))
            else:
                dw2 = k_rows_reduce(part, 1, chunks, c * 72).view(c, 8, 3, 3)
        dz1, dg1, db1, _ = bn_backward_fused(da, z1, None, coef1, ACT_RELU, m_in, parts=parts1)
        dws = dgs = dbs = None
        addend, add_s2 = g, False                               # identity block: + gradient of the identity branch
        if zs is not None:
            dzs, dgs, dbs, _ = bn_backward_fused(g, zs, None, coefs, ACT_NONE, m)
            addend, _, _ = k_gemm_nt(dzs, None, 0, grid, wts, cin, kp32(c), 1)           # shortcut data gradient on the output grid
            add_s2 = stride == 2
            if group is not None:
                group.add(ws_, x, dzs, 0 if stride == 1 else 1, grid, cin, c)
            else:
                dws = k_gemm_tn(x, None, 0 if stride == 1 else 1, grid, dzs, c, kp32(cin), 1, cin, defer=batch)
        dx = None
        if ctx.needs_input_grad[0] and z3p is not None:
            dx, pg3, pgx3 = k_gemm_nt(dz1, None, 0, (n, h, w), wt1, cin, kp32(c), 1, addend=addend, estat=(3, z3p, coef3p, x))
            group.bn3_parts = (dx, pg3, pgx3)
        elif ctx.needs_input_grad[0]:
            dx, _, _ = k_gemm_nt(dz1, None, 0, (n, h, w), wt1, cin, kp32(c), 1, addend=addend, add_s2=add_s2)
        if group is not None:
            dw1 = group.add(w1_, x, dz1, 0, (n, h, w), cin, c)
        else:
            dw1 = k_gemm_tn(x, None, 0, (n, h, w), dz1, c, kp32(cin), 1, cin, defer=batch)
        batch.flush()
        return (dx, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, dsw1, dsb1, dsw2, dsb2, dw3, dg3, db3, None, None,
                None, None, None, None, dws, dgs, dbs, None, None, None)


# --------------------------------------------------------------------------------------------------------------
# Inference with folded BatchNorm (BASELINE config 5; reference: demo.py:191-202 runs the eval-mode module).  In eval mode
# BN(conv(x)) = conv(x, W * scale) + shift with scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (+ conv bias
# * scale): HydraNet.prepare_inference() folds scale into the packed bf16 weights once and keeps shift as an fp32 bias, so conv + BN +
# activation (+ the XBlock's identity branch) is ONE GEMM launch with a bias / addend / activation epilogue.
# --------------------------------------------------------------------------------------------------------------
def fold_conv_bn(w, conv_bias, gamma, beta, rm, rv, eps, kind):
    """-> (packed bf16 operand, fp32 bias) for the inference path; kind "1x1" or "g3x3" """
    with torch.no_grad():
        scale = gamma.float() / torch.sqrt(rv.float() + eps)
        shift = beta.float() - rm.float() * scale
        if conv_bias is not None:
            shift = shift + conv_bias.float() * scale
        wf = (w.float() * scale.view(-1, 1, 1, 1)).contiguous()
        if kind == "g3x3":
            c = wf.shape[0]
            wk = torch.empty((c, 9 * 64), device=wf.device, dtype=BF16)
            wd = torch.empty((c, 9 * 64), device=wf.device, dtype=BF16)
            lib().call("hn_gconv_pack_diag", ptr(wf), ptr(wk), ptr(wd), c)
            return wk, shift.contiguous()
        cout, cin = wf.shape[0], wf.shape[1]
        wp = torch.empty((cout, kp32(cin)), device=wf.device, dtype=BF16)
        lib().call("hn_pack_weight", ptr(wf), ptr(wp), None, cout, cin, 1)
        return wp, shift.contiguous()


INFER_GATE_IN_WEIGHTS = policy("HN_INFER_GATE_IN_WEIGHTS", "1") != "0"


def conv_infer(x, packed, bias, cout, kind, stride, act, res=None, gate=None):
    """act(conv(x [* gate]) + bias [+ res]) with folded-BatchNorm operands: one launch.  gate [N, Cin] fp32 (the SE excitation of an XBlock,
    x = the un-gated activation): folded into the 1x1 weights per image (W_n = W diag(gate_n): one small launch over N x Cout x Cin weights
    instead of a read + write pass over the activation) where a pixel tile lies inside one image; else applied to x first."""
    n, hi, wi, cin = x.shape
    ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
    if gate is not None:
        hw = hi * wi
        if INFER_GATE_IN_WEIGHTS and kind == "1x1" and stride == 1 and hw % 128 == 0:
            kp = kp32(cin)
            wn = torch.empty((n, cout, kp), device=x.device, dtype=BF16)
            lib().call("hn_scale_weight_gate", ptr(packed), ptr(gate), ptr(wn), n, cout, cin, kp)
            out = new_act(n, ho, wo, cout, x.device)
            lib().call("hn_conv_gemm_nt_imgw", ptr(x), ld(x), n * hw, cin, ptr(wn), cout * kp, hw, cout, kp, ptr(bias), act, ptr(out), ld(out),
                       ptr(res), ld(res) if res is not None else 0)
            return out
        x = apply_gate_rows(x, gate)
    if kind == "g3x3":
        assert stride == 1 and res is None
        out, _, _ = k_gemm_nt(x, None, 5, (n, ho, wo), packed, cout, 64, 9, bias=bias, act=act)
    else:
        out, _, _ = k_gemm_nt(x, None, 0 if stride == 1 else 1, (n, ho, wo), packed, cout, kp32(cin), 1, bias=bias, act=act, addend=res,
                              add_pre=res is not None)
    return out


def gate_folds_into_weights(b):
    """conv_infer(gate=...) scales the 1x1 weights per image instead of the activation (a pixel tile must lie inside one image)"""
    return INFER_GATE_IN_WEIGHTS and (b.shape[1] * b.shape[2]) % 128 == 0


def se_gate_infer(b, w1, b1, w2, b2, apply=True):
    """SE squeeze / excite for the inference path: per-image channel sums (one pass), the MLP fed by the partial rows, then b * gate
    (apply = False: the gate [N, C] itself, for conv_infer(gate=...))"""
    n, h, w, c = b.shape
    hw, m = h * w, n * h * w
    cs = w1.shape[0]
    dev = b.device
    rb = lib().query("hn_fused_row_block", m, c, hw, 0, 1)
    if hw % rb:
        rb = hw
    pr = m // rb
    ps = torch.empty((pr, c), device=dev, dtype=F32)
    pq = torch.empty((pr, c), device=dev, dtype=F32)
    lib().call("hn_col_stats_fused", ptr(b), ld(b), m, c, rb, ptr(ps), ptr(pq))
    pooled = torch.empty((n, c), device=dev, dtype=F32)
    hid = torch.empty((n, cs), device=dev, dtype=F32)
    gate = torch.empty((n, c), device=dev, dtype=F32)
    if apply and se_gate_in_apply(m, c):      # second layer + product in one launch
        lib().call("hn_se_mlp_fwd_parts", ptr(ps), hw // rb, 1.0 / hw, ptr(w1), ptr(b1), None, None, ptr(pooled), ptr(hid), None, n, c, cs)
        out = new_act(n, h, w, c, dev)
        lib().call("hn_se_gate_apply", ptr(b), ld(b), None, ACT_NONE, ptr(hid), ptr(w2), ptr(b2), None, ptr(out), ld(out), n, hw, c, cs)
        return out
    lib().call("hn_se_mlp_fwd_parts", ptr(ps), hw // rb, 1.0 / hw, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(pooled), ptr(hid), ptr(gate), n, c, cs)
    return gate if not apply else apply_gate_rows(b, gate)


def apply_gate_rows(b, gate):
    """b * gate[n][c] (one pass over the activation)"""
    n, h, w, c = b.shape
    hw, m = h * w, n * h * w
    out = new_act(n, h, w, c, b.device)
    rb2 = lib().query("hn_fused_row_block", m, c, hw, 0, 0)
    lib().call("hn_bn_apply_fused", ptr(b), ld(b), m, c, None, None, 0, m, None, None, 0.0, 0.0, None, None, None, None, 0, ACT_NONE, ptr(out),
               ld(out), None, ptr(gate), hw, rb2)
    return out


def xblock_fusable(x, w1, stride, has_se, has_shortcut):
    """the fused node covers XBlocks with SE whose channel counts are multiples of 8: the stride-1 identity blocks and the stride-2 first
    block of a stage (projection shortcut), at ANY output grid (round 5: the 128-pixel multiple was a requirement of the register-staged
    operand-transform loader only -- with it the reference's default 640 x 640 input, whose stages 2-4 are 40 x 40, 20 x 20 and 10 x 10
    maps, ran 28 of its 30 blocks as the unfused 16 + 27 launch composition and was SLOWER than 512 x 1024).  Every per-image quantity
    (SE squeeze partial rows, gate rows) uses row blocks that divide the image's pixel count (hn_fused_row_block); the GEMM and conv
    statistics epilogues mask their tail rows."""
    cout, cin = w1.shape[0], w1.shape[1]
    ho, wo = x.shape[1] // stride, x.shape[2] // stride
    shape_ok = (stride == 1 and not has_shortcut and cout == cin) or (stride == 2 and has_shortcut and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0)
    xf_ok = not XBLOCK_XF_GEMM or ((ho * wo) % 128 == 0 and 3 * kp32(cout) * 4 <= 32768)
    return (FUSED_XBLOCK and FUSED_BN and GCONV_MFMA and x.is_cuda and has_se and shape_ok and cout % 8 == 0 and cin % 8 == 0
            and xf_ok and x.shape[0] * x.shape[1] * x.shape[2] < (1 << 31))


# --------------------------------------------------------------------------------------------------------------
# Squeeze-and-Excitation: out = b * sigmoid(W2 relu(W1 avgpool(b) + b1) + b2)        (net/anynet.py:40-48,68-69)
# pooling, the [N, C] x [C, C/4] excitation MLP (hn_se_mlp_fwd / hn_se_mlp_bwd), gating and their backward are all HIP kernels.
# --------------------------------------------------------------------------------------------------------------
class SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, w1, b1, w2, b2):
        n, h, w, c = b.shape
        hw = h * w
        cs = w1.shape[0]
        dev = b.device
        ps, _, r = k_col_stats(b, align=hw)
        pooled = k_rows_reduce(ps, n, hw // r, c, 1.0 / hw)                      # [N, C]
        hid = torch.empty((n, cs), device=dev, dtype=F32)
        gate = torch.empty((n, c), device=dev, dtype=F32)
        lib().call("hn_se_mlp_fwd", ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hid), ptr(gate), n, c, cs)
        out = new_act(n, h, w, c, dev)
        lib().call("hn_scale_rows", ptr(b), ld(b), ptr(gate), hw, ptr(out), ld(out), n * hw, c)
        ctx.save_for_backward(b, pooled, hid, gate, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, pooled, hid, gate, w1, w2 = ctx.saved_tensors
        dout = dense(dout)
        n, h, w, c = b.shape
        hw = h * w
        m = n * hw
        cs = w1.shape[0]
        dev = b.device
        r = lib().query("hn_colred_rows", m, hw)
        pr = (m + r - 1) // r
        pd = torch.empty((pr, c), device=dev, dtype=F32)
        pz = torch.empty((pr, c), device=dev, dtype=F32)
        lib().call("hn_col_dot", ptr(dout), ld(dout), ptr(b), ld(b), m, c, r, ptr(pd), ptr(pz))
        dgate = k_rows_reduce(pd, n, hw // r, c, 1.0)                            # sum_hw dout * b
        dpre2 = torch.empty((n, c), device=dev, dtype=F32)
        dpool = torch.empty((n, c), device=dev, dtype=F32)
        dpre1 = torch.empty((n, cs), device=dev, dtype=F32)
        dw1, db1 = torch.empty_like(w1), torch.empty((cs,), device=dev, dtype=F32)
        dw2, db2 = torch.empty_like(w2), torch.empty((c,), device=dev, dtype=F32)
        lib().call("hn_se_mlp_bwd", ptr(dgate), ptr(gate), ptr(hid), ptr(pooled), ptr(w1), ptr(w2), ptr(dpre2), ptr(dpre1), ptr(dpool),
                   ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), n, c, cs)
        db = new_act(n, h, w, c, dev)
        lib().call("hn_se_bwd_apply", ptr(dout), ld(dout), ptr(gate), ptr(dpool), hw, ptr(db), ld(db), m, c)
        return db, dw1, db1, dw2, db2


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
