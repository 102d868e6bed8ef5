"""ops.losses -- segmentation / detection / lane losses, deploy arg-max and the weighted loss sum on HIP kernels (reference:
head_seg/segmentation_loss.py:27-65, head_detect/detection_loss.py:132-267, head_lane/lanedetect_loss.py:18-78, train.py:192-203)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib
from .core import *        # noqa: F401,F403
from .backbone import *        # noqa: F401,F403
from .neck import *        # noqa: F401,F403
from .seg import *        # noqa: F401,F403
from .heads import *        # noqa: F401,F403


# --------------------------------------------------------------------------------------------------------------
# segmentation loss (weighted CE + ignore_index + top-k hardest pixels) and deploy-mode argmax
# --------------------------------------------------------------------------------------------------------------
class SegLoss(torch.autograd.Function):
    """logits: fp32 NHWC [N, H, W, C] (dense rows); target: [N, H, W] int64 or float32 class ids."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, use_top_k, top_k_ratio, ignore_index, slot=None):
        """slot (GradSlot of the producing SegOutUp node): the gradient is handed over as that node's space-to-depth bf16 operand
        (hn_seg_loss_bwd_s2d) instead of an fp32 dlogits tensor"""
        n, h, w, c = logits.shape
        hw = h * w
        ctx.slot = slot if (slot is not None and h % 2 == 0 and w % 2 == 0) else None
        ctx.hw_dims = (h, w)
        k = int(top_k_ratio * hw) if use_top_k else hw
        dev = logits.device
        ws = torch.empty((lib().query("hn_seg_loss_ws_bytes", n, hw),), device=dev, dtype=torch.uint8)
        out = torch.empty((1,), device=dev, dtype=F32)
        tf = 1 if target.dtype == torch.float32 else 0
        assert target.dtype in (torch.float32, torch.int64) and target.is_contiguous()
        lib().call("hn_seg_loss_fwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(class_weights), ignore_index, n, hw,
                   1 if use_top_k else 0, k, ptr(ws), ptr(out))
        ctx.meta = (n, hw, c, tf, 1 if use_top_k else 0, k, ignore_index)
        ctx.save_for_backward(logits, target, class_weights, ws)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, cw, ws = ctx.saved_tensors
        n, hw, c, tf, topk, k, ign = ctx.meta
        g = gout.contiguous().to(F32).view(1)
        if ctx.slot is not None and ctx.slot.buf is None:
            h, w = ctx.hw_dims
            dz = new_act(n, h // 2, w // 2, pad8(4 * c), logits.device)
            lib().call("hn_seg_loss_bwd_s2d", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), ign, n, h, w, topk, k, ptr(ws), ptr(g),
                       ptr(dz), ld(dz))
            ctx.slot.buf = dz
            return None, None, None, None, None, None, None
        dl = torch.empty_like(logits)
        lib().call("hn_seg_loss_bwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), ign, n, hw, topk, k, ptr(ws), ptr(g),
                   ptr(dl), dl.stride(2))
        return dl, None, None, None, None, None, None


def seg_loss_hip(seg_nchw, target, class_weights, use_top_k, top_k_ratio, ignore_index=255, slot=None):
    """seg_nchw: the module's "seg" output (fp32, NCHW-shaped view of NHWC memory).  slot: GradSlot of the SegOutUp node that produced
    exactly this tensor (HydraNet passes it when its own cal_loss consumes its own "seg" output)."""
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits, slot = logits.contiguous(), None
    return SegLoss.apply(logits, target.contiguous(), class_weights, use_top_k, top_k_ratio, ignore_index, slot)


class SegFocalLoss(torch.autograd.Function):
    """focal variant of the seg loss (head_seg/segmentation_loss.py:31-46): logits fp32 NHWC [N, H, W, C] (dense rows), target [N, H, W]
    int64 or float32 class ids; mean over all pixels"""

    @staticmethod
    def forward(ctx, logits, target, class_weights, gamma, alpha):
        n, h, w, c = logits.shape
        hw = h * w
        dev = logits.device
        tf = 1 if target.dtype == torch.float32 else 0
        assert target.dtype in (torch.float32, torch.int64) and target.is_contiguous()
        ws = torch.empty((lib().query("hn_seg_loss_blocks", n, hw),), device=dev, dtype=F32)
        out = torch.empty((1,), device=dev, dtype=F32)
        lib().call("hn_seg_focal_fwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(class_weights), float(gamma), float(alpha), n, hw,
                   ptr(ws), ptr(out))
        ctx.meta = (n, hw, c, tf, float(gamma), float(alpha))
        ctx.save_for_backward(logits, target, class_weights)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, cw = ctx.saved_tensors
        n, hw, c, tf, gamma, alpha = ctx.meta
        g = gout.contiguous().to(F32).view(1)
        dl = torch.empty_like(logits)
        lib().call("hn_seg_focal_bwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), gamma, alpha, n, hw, ptr(g), ptr(dl),
                   dl.stride(2))
        return dl, None, None, None, None


def seg_focal_loss_hip(seg_nchw, target, class_weights, gamma=2.0, alpha=1.0):
    """CrossEntropyLoss.forward with use_focal=True (gamma 2, alpha 1: the defaults model.py:119-124 leaves untouched)"""
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits = logits.contiguous()
    return SegFocalLoss.apply(logits, target.contiguous(), class_weights, gamma, alpha)


def argmax_channels(seg_nchw):
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits = logits.contiguous()
    n, h, w, c = logits.shape
    out = torch.empty((n, h, w), device=logits.device, dtype=torch.int64)
    lib().call("hn_argmax_channels", ptr(logits), logits.stride(2), c, n * h * w, ptr(out))
    return out


# --------------------------------------------------------------------------------------------------------------
# detection loss (focal BCE + smooth-L1 with IoU anchor assignment)
# --------------------------------------------------------------------------------------------------------------
class DetLoss(torch.autograd.Function):
    """returns a 2-vector (classification loss, regression loss), both batch means like FocalLoss.forward."""

    @staticmethod
    def forward(ctx, cls, reg, anchors, ann):
        n, a, k = cls.shape
        mx = ann.shape[1]
        dev = cls.device
        cls, reg, ann = cls.contiguous(), reg.contiguous(), ann.contiguous().float()
        anc = anchors.reshape(-1, 4).contiguous()
        blocks = lib().query("hn_det_loss_blocks", a)
        assign = torch.empty((n, a), device=dev, dtype=torch.int16)
        part = torch.empty((n, blocks, 3), device=dev, dtype=F32)
        npos = torch.empty((n,), device=dev, dtype=F32)
        out = torch.empty((2,), device=dev, dtype=F32)
        lib().call("hn_det_loss_fwd", ptr(cls), ptr(reg), ptr(anc), ptr(ann), n, a, k, mx, ptr(assign), ptr(part), ptr(npos), ptr(out))
        ctx.save_for_backward(cls, reg, anc, ann, assign, npos)
        # the two losses leave as two outputs (slicing a 2-vector OUTSIDE the node cost two SliceBackward nodes -- fill + copy each -- and
        # an add per step: seven 5 us launches)
        return out[0:1], out[1:2]

    @staticmethod
    def backward(ctx, gcls, greg):
        cls, reg, anc, ann, assign, npos = ctx.saved_tensors
        n, a, k = cls.shape
        dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)
        if (gcls is not None and greg is not None and gcls.dtype == F32 and greg.dtype == F32
                and greg.data_ptr() == gcls.data_ptr() + 4):       # neighbours in WeightedLossSum.backward's gradient vector: used in place
            g = gcls
        else:
            z = zeros((1,), cls.device)
            g = torch.cat([(gcls if gcls is not None else z).reshape(1).to(F32), (greg if greg is not None else z).reshape(1).to(F32)])
        lib().call("hn_det_loss_bwd", ptr(cls), ptr(reg), ptr(anc), ptr(ann), n, a, k, ann.shape[1], ptr(assign), ptr(npos), ptr(g), ptr(dcls),
                   ptr(dreg))
        return dcls, dreg, None, None


def det_loss_hip(classification, regression, anchors, annotations):
    return DetLoss.apply(classification, regression, anchors, annotations)



# --------------------------------------------------------------------------------------------------------------
# Lane losses on the device (head_lane/lanedetect_loss.py:18-78): one workgroup does the OHEM classification loss (log-softmax, counts,
# radix select of the k-th smallest background log-prob instead of torch.sort/topk, both sums); the location loss is a row kernel + a
# one-block finalize.  ~5 launches instead of ~60 tiny torch ops, and no memcpy nodes in the captured step.
# --------------------------------------------------------------------------------------------------------------
class LaneClsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_targets, cls_preds, negative_ratio, alpha):
        tgt = cls_targets.reshape(-1, 2).float().contiguous()
        z = cls_preds.reshape(-1, 2).float().contiguous()
        m = z.shape[0]
        dev = z.device
        lsm = torch.empty((m, 2), device=dev, dtype=F32)
        pmask = torch.empty((m,), device=dev, dtype=torch.uint8)
        out = torch.empty((2,), device=dev, dtype=F32)
        aux = torch.empty((4,), device=dev, dtype=F32)
        lib().call("hn_lane_cls_loss_fwd", ptr(z), ptr(tgt), m, float(negative_ratio), float(alpha), ptr(lsm), ptr(pmask), ptr(out), ptr(aux))
        ctx.alpha, ctx.shape = float(alpha), cls_preds.shape
        ctx.save_for_backward(lsm, pmask, aux)
        ctx.mark_non_differentiable(pmask, aux)
        ctx.set_materialize_grads(False)        # (no zero-filled "gradients" of the byte mask and the aux vector: two fill launches per step)
        return out[0], out[1], pmask, aux

    @staticmethod
    def backward(ctx, gpos, gneg, _gm, _ga):
        lsm, pmask, aux = ctx.saved_tensors
        m = lsm.shape[0]
        dz = torch.empty((m, 2), device=lsm.device, dtype=F32)
        gp = gpos.reshape(1).to(F32) if gpos is not None else zeros((1,), lsm.device)
        gn = gneg.reshape(1).to(F32) if gneg is not None else zeros((1,), lsm.device)
        lib().call("hn_lane_cls_loss_bwd", ptr(lsm), ptr(pmask), ptr(aux), ptr(gp), ptr(gn), ctx.alpha, m, ptr(dz))
        return None, dz.view(ctx.shape), None, None


class LaneLocLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pmask, aux, loc_targets, loc_preds, wcol, alpha):
        L = loc_preds.shape[-1]
        if wcol + 1 >= L:
            raise IndexError(f"index {wcol + 1} is out of bounds for dimension 1 with size {L}")   # the reference's own failure mode
        p = loc_preds.reshape(-1, L).float().contiguous()
        t = loc_targets.reshape(-1, L).float().contiguous()
        m = p.shape[0]
        dev = p.device
        rowloss = torch.empty((m,), device=dev, dtype=F32)
        rownorm = torch.empty((m,), device=dev, dtype=F32)
        out = torch.empty((1,), device=dev, dtype=F32)
        lib().call("hn_lane_loc_loss_fwd", ptr(p), ptr(t), ptr(pmask), ptr(aux), m, L, int(wcol), float(alpha), ptr(rowloss), ptr(rownorm),
                   ptr(out))
        ctx.meta = (int(wcol), float(alpha), loc_preds.shape)
        ctx.save_for_backward(p, t, pmask, rownorm, aux)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        p, t, pmask, rownorm, aux = ctx.saved_tensors
        wcol, alpha, shape = ctx.meta
        m, L = p.shape
        dp = torch.empty((m, L), device=p.device, dtype=F32)
        g = gout.reshape(1).to(F32)
        lib().call("hn_lane_loc_loss_bwd", ptr(p), ptr(t), ptr(pmask), ptr(rownorm), ptr(aux), ptr(g), m, L, wcol, alpha, ptr(dp))
        return None, None, None, dp.view(shape), None, None


def lane_cls_loss_hip(cls_targets, cls_preds, negative_ratio=15, alpha=10.0):
    """cal_loss_cls (lanedetect_loss.py:18-54): returns (pos, neg, pmask, positive_num) like the reference; pmask / positive_num are
    device-side handles (byte mask, aux vector) consumed by lane_loc_loss_hip."""
    pos, neg, pmask, aux = LaneClsLoss.apply(cls_targets, cls_preds, negative_ratio, alpha)
    return pos, neg, pmask, aux


def lane_loc_loss_hip(pmask, positive_num, loc_targets, loc_preds, alpha=10.0, points_per_line=160):
    """cal_loss_regress (lanedetect_loss.py:57-78) incl. its hard-coded points_per_line = 160 default (x10 weights on columns 160/161)."""
    return LaneLocLoss.apply(pmask, positive_num, loc_targets, loc_preds, points_per_line, alpha)


# --------------------------------------------------------------------------------------------------------------
# HydraTrainer.cal_total_loss (model/train.py:192-203) as one launch forward and one backward instead of ~22 scalar torch kernels.
# --------------------------------------------------------------------------------------------------------------
class WeightedLossSum(torch.autograd.Function):
    """total = sum_g (sum_{i in g} x_i * w_i) * gw_g; meta = (w tuple, gw tuple, group-id tuple), xs = fp32 scalar device tensors"""

    @staticmethod
    def forward(ctx, meta, *xs):
        w, gw, grp = meta
        n = len(xs)
        ctx.shapes = [x.shape for x in xs]
        xs = [x.reshape(1) for x in xs]
        assert all(x.dtype == F32 and x.is_cuda for x in xs)
        out = torch.empty((1,), device=xs[0].device, dtype=F32)
        ctx.host = (_ptr_array(xs), (ctypes.c_float * n)(*w), (ctypes.c_float * len(gw))(*gw), (ctypes.c_int * n)(*grp), n)
        pa, wa, ga, ia, _ = ctx.host
        lib().call("hn_weighted_sum", ctypes.addressof(pa), ctypes.addressof(wa), ctypes.addressof(ga), ctypes.addressof(ia), n, None, ptr(out), None)
        ctx.keep = xs                                     # the pointer table refers to these
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        pa, wa, ga, ia, n = ctx.host
        grads = torch.empty((n,), device=gout.device, dtype=F32)
        g = gout.reshape(1)
        if g.dtype != F32:
            g = g.float()
        lib().call("hn_weighted_sum", ctypes.addressof(pa), ctypes.addressof(wa), ctypes.addressof(ga), ctypes.addressof(ia), n, ptr(g), None, ptr(grads))
        return (None, *[grads[i:i + 1].view(shape) for i, shape in enumerate(ctx.shapes)])


def weighted_loss_sum(groups):
    """groups = [(group weight, [(loss tensor, weight), ...]), ...] -> the reference's total loss, same association order"""
    xs, w, gw, grp = [], [], [], []
    for gi, (gweight, terms) in enumerate(groups):
        gw.append(float(gweight))
        for x, wi in terms:
            xs.append(x)
            w.append(float(wi))
            grp.append(gi)
    return WeightedLossSum.apply((tuple(w), tuple(gw), tuple(grp)), *xs)


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
