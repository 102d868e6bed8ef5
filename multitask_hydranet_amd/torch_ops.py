"""Dispatcher-visible registration of the HIP entry points (`torch.ops.hydranet_hip.*`), SURVEY.md section 8(b): the reference plugs its one
hand-written op in through torch.autograd.Function (net/common.py:11-22); a drop-in under PyTorch-ROCm additionally registers its native
ops with the dispatcher so that they carry schemas, work under torch.no_grad / inference_mode bookkeeping, can be traced / exported and
checked with torch.library.opcheck.  Every op here is a thin shim over the same C-ABI call the nn.Module path makes (no second
implementation): the forward / backward pairs are tied together with register_autograd, fake (meta) kernels give the output shapes.

    import multitask_hydranet_amd.torch_ops          # registers the namespace
    y = torch.ops.hydranet_hip.conv1x1_bn_act(x, w, gamma, beta, rm, rv, 1, 1e-5, 0.1, True)
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from . import ops as K
from ._lib import lib

NS = "hydranet_hip"


# ---- conv 1x1 + training/eval BatchNorm + activation (net/anynet.py:29-33) -------------------------------------------------------------
@torch.library.custom_op(f"{NS}::conv1x1_bn_act_fwd", mutates_args=(), device_types="cuda")
def conv1x1_bn_act_fwd(x: Tensor, weight: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, act: int, eps: float,
                       momentum: float, training: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """x NHWC bf16 [N,H,W,Cin], weight fp32 [Cout,Cin,1,1] -> (out NHWC bf16, z = conv output before BN, coef [4,Cout] = scale, shift,
    batch mean, rstd).  FUNCTIONAL (an operator with an autograd formula may not mutate its inputs): the running statistics are read in
    eval mode and left untouched in training mode -- conv1x1_bn_act() below applies the momentum update from `coef`."""
    n, h, w, cin = x.shape
    cout = weight.shape[0]
    wp, _ = K.pack_conv_weight(weight)
    z, ps, pq = K.k_gemm_nt(x, None, 0, (n, h, w), wp, cout, K.kp32(cin), 1, stats=training)
    out, coef, _, _ = K.k_bn_apply_fused(z, ps, pq, n * h * w, gamma, beta, eps, momentum, None if training else running_mean,
                                         None if training else running_var, act, training=training)
    return out, z, coef


@conv1x1_bn_act_fwd.register_fake
def _(x, weight, gamma, beta, running_mean, running_var, act, eps, momentum, training):
    n, h, w, _ = x.shape
    cout = weight.shape[0]
    return x.new_empty((n, h, w, cout)), x.new_empty((n, h, w, cout)), gamma.new_empty((4, cout))


@torch.library.custom_op(f"{NS}::conv1x1_bn_act_bwd", mutates_args=(), device_types="cuda")
def conv1x1_bn_act_bwd(dout: Tensor, x: Tensor, weight: Tensor, z: Tensor, coef: Tensor, act: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (dx, dweight, dgamma, dbeta)"""
    n, h, w, cin = x.shape
    cout = weight.shape[0]
    dz, dgamma, dbeta, _ = K.bn_backward_fused(K.dense(dout), z, None, coef, act, n * h * w)
    _, wt = K.pack_conv_weight(weight)
    dx, _, _ = K.k_gemm_nt(dz, None, 0, (n, h, w), wt, cin, K.kp32(cout), 1)
    dw = K.k_gemm_tn(x, None, 0, (n, h, w), dz, cout, K.kp32(cin), 1, cin)
    return dx, dw, dgamma, dbeta


@conv1x1_bn_act_bwd.register_fake
def _(dout, x, weight, z, coef, act):
    return torch.empty_like(x), torch.empty_like(weight), coef.new_empty((weight.shape[0],)), coef.new_empty((weight.shape[0],))


def _cba_setup(ctx, inputs, output):
    x, weight, *_rest = inputs
    ctx.act = inputs[6]
    ctx.save_for_backward(x, weight, output[1], output[2])


def _cba_backward(ctx, dout, dz_unused, dcoef_unused):
    x, weight, z, coef = ctx.saved_tensors
    dx, dw, dg, db = torch.ops.hydranet_hip.conv1x1_bn_act_bwd(dout, x, weight, z, coef, ctx.act)
    return dx, dw, dg, db, None, None, None, None, None, None


conv1x1_bn_act_fwd.register_autograd(_cba_backward, setup_context=_cba_setup)


def conv1x1_bn_act(x, weight, gamma, beta, running_mean, running_var, act=1, eps=1e-5, momentum=0.1, training=True):
    out, _, coef = torch.ops.hydranet_hip.conv1x1_bn_act_fwd(x, weight, gamma, beta, running_mean, running_var, act, eps, momentum, training)
    if training:                                            # F.batch_norm's running-statistics update, from the batch mean / rstd
        with torch.no_grad():
            cnt = x.shape[0] * x.shape[1] * x.shape[2]
            var = (1.0 / (coef[3] * coef[3]) - eps).clamp_min(0.0)
            running_mean.mul_(1.0 - momentum).add_(coef[2], alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(var * (cnt / max(cnt - 1, 1)), alpha=momentum)
    return out


# ---- weighted top-k cross entropy (head_seg/segmentation_loss.py:48-65) ------------------------------------------------------------------
@torch.library.custom_op(f"{NS}::seg_topk_ce_fwd", mutates_args=(), device_types="cuda")
def seg_topk_ce_fwd(logits: Tensor, target: Tensor, class_weights: Tensor, use_top_k: bool, top_k_ratio: float, ignore_index: int) -> Tuple[Tensor, Tensor]:
    """logits fp32 NHWC [N,H,W,C] dense, target int64|float32 [N,H,W] -> (mean loss [], workspace kept for backward)"""
    n, h, w, c = logits.shape
    hw = h * w
    k = int(top_k_ratio * hw) if use_top_k else hw
    ws = torch.empty((lib().query("hn_seg_loss_ws_bytes", n, hw),), device=logits.device, dtype=torch.uint8)
    out = torch.empty((1,), device=logits.device, dtype=torch.float32)
    lib().call("hn_seg_loss_fwd", logits.data_ptr(), logits.stride(2), c, target.data_ptr(), 1 if target.dtype == torch.float32 else 0,
               class_weights.data_ptr(), ignore_index, n, hw, 1 if use_top_k else 0, k, ws.data_ptr(), out.data_ptr())
    return out.view(()), ws


@seg_topk_ce_fwd.register_fake
def _(logits, target, class_weights, use_top_k, top_k_ratio, ignore_index):
    return logits.new_empty(()), logits.new_empty((1,), dtype=torch.uint8)


@torch.library.custom_op(f"{NS}::seg_topk_ce_bwd", mutates_args=(), device_types="cuda")
def seg_topk_ce_bwd(gout: Tensor, logits: Tensor, target: Tensor, class_weights: Tensor, ws: Tensor, use_top_k: bool, top_k_ratio: float,
                    ignore_index: int) -> Tensor:
    n, h, w, c = logits.shape
    hw = h * w
    k = int(top_k_ratio * hw) if use_top_k else hw
    dl = torch.empty_like(logits)
    g = gout.contiguous().to(torch.float32).view(1)
    lib().call("hn_seg_loss_bwd", logits.data_ptr(), logits.stride(2), c, target.data_ptr(), 1 if target.dtype == torch.float32 else 0,
               class_weights.data_ptr(), ignore_index, n, hw, 1 if use_top_k else 0, k, ws.data_ptr(), g.data_ptr(), dl.data_ptr(), dl.stride(2))
    return dl


@seg_topk_ce_bwd.register_fake
def _(gout, logits, target, class_weights, ws, use_top_k, top_k_ratio, ignore_index):
    return torch.empty_like(logits)


def _ce_setup(ctx, inputs, output):
    logits, target, cw, ctx.use_top_k, ctx.ratio, ctx.ignore = inputs
    ctx.save_for_backward(logits, target, cw, output[1])


def _ce_backward(ctx, gloss, gws_unused):
    logits, target, cw, ws = ctx.saved_tensors
    return torch.ops.hydranet_hip.seg_topk_ce_bwd(gloss, logits, target, cw, ws, ctx.use_top_k, ctx.ratio, ctx.ignore), None, None, None, None, None


seg_topk_ce_fwd.register_autograd(_ce_backward, setup_context=_ce_setup)


# ---- inference-side ops (no gradient) ------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{NS}::argmax_channels", mutates_args=(), device_types="cuda")
def argmax_channels(logits: Tensor) -> Tensor:
    """fp32 NHWC [N,H,W,C] dense -> int64 [N,H,W], first maximum wins (model/model.py:197)"""
    n, h, w, c = logits.shape
    out = torch.empty((n, h, w), device=logits.device, dtype=torch.int64)
    lib().call("hn_argmax_channels", logits.data_ptr(), logits.stride(2), c, n * h * w, out.data_ptr())
    return out


@argmax_channels.register_fake
def _(logits):
    return logits.new_empty(logits.shape[:3], dtype=torch.int64)


@torch.library.custom_op(f"{NS}::det_postprocess", mutates_args=(), device_types="cuda")
def det_postprocess(anchors: Tensor, regression: Tensor, classification: Tensor, img_h: int, img_w: int, threshold: float, iou_threshold: float,
                    cap: int) -> List[Tensor]:
    """-> [rois [N,cap,4], class_ids int64 [N,cap], scores [N,cap], kept int32 [N], total int32 [N]] (head_detect/detection_loss.py:70-108)"""
    from .postprocess import postprocess_device
    r = postprocess_device((img_h, img_w), anchors, regression, classification, threshold, iou_threshold, cap)
    return [r["rois"], r["class_ids"], r["scores"], r["kept"], r["total"]]


@det_postprocess.register_fake
def _(anchors, regression, classification, img_h, img_w, threshold, iou_threshold, cap):
    n = regression.shape[0]
    c = min(cap, regression.shape[1])
    f = regression
    return [f.new_empty((n, c, 4)), f.new_empty((n, c), dtype=torch.int64), f.new_empty((n, c)), f.new_empty((n,), dtype=torch.int32),
            f.new_empty((n,), dtype=torch.int32)]


@torch.library.custom_op(f"{NS}::preprocess_bgr", mutates_args=(), device_types="cuda")
def preprocess_bgr(frames: Tensor, out_h: int, out_w: int) -> Tensor:
    """uint8 [N,H,W,3] BGR -> fp32 [N,3,out_h,out_w] RGB, ImageNet-normalised (demo.py:186-196)"""
    from .preprocess import preprocess_bgr as _pp
    return _pp(frames, (out_h, out_w))


@preprocess_bgr.register_fake
def _(frames, out_h, out_w):
    return frames.new_empty((frames.shape[0], 3, out_h, out_w), dtype=torch.float32)
