"""TEST REFERENCE (not part of the product package): the multitask loss of HydraNet.cal_loss (model/model.py:201-264) as static-shape
torch ops that run on any device -- the fp32 checker the GPU tests hold the HIP loss kernels against on the device.  Values are identical to the reference's per-image Python loops
(head_detect/detection_loss.py:132-267, head_lane/lanedetect_loss.py:18-78, head_seg/segmentation_loss.py:27-65); only the control
flow differs (masks instead of boolean compaction, a sort instead of a data-dependent top-k).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def seg_loss(logits: torch.Tensor, target: torch.Tensor, class_weights: torch.Tensor, use_top_k: bool, top_k_ratio: float,
             use_focal: bool, ignore_index: int = 255, gamma: float = 2.0, alpha: float = 1.0) -> torch.Tensor:
    """logits [N, C, H, W] (any memory format), target int64 [N, H, W]."""
    b = logits.shape[0]
    cw = class_weights.to(device=logits.device, dtype=logits.dtype)
    if use_focal:
        soft = F.softmax(logits, dim=1) + 1e-8
        one_hot = torch.zeros_like(logits, dtype=target.dtype).scatter_(1, target.unsqueeze(1), 1.0) + 1e-8
        focal = -alpha * torch.pow(1.0 - soft, gamma) * torch.log(soft) * cw.view(1, -1, 1, 1)
        loss = torch.sum(one_hot * focal, dim=1).reshape(b, -1)
    else:
        loss = F.cross_entropy(logits, target, weight=cw, ignore_index=ignore_index, reduction="none").reshape(b, -1)
        if use_top_k:
            k = int(top_k_ratio * loss.shape[1])
            loss, _ = torch.sort(loss, dim=1, descending=True)
            loss = loss[:, :k]
    return torch.mean(loss)


def det_loss(classification: torch.Tensor, regression: torch.Tensor, anchors: torch.Tensor, annotations: torch.Tensor):
    """classification [N, A, K] (post-sigmoid), regression [N, A, 4], anchors [1, A, 4] (y1,x1,y2,x2), annotations [N, M, 5]
    (x1,y1,x2,y2,cls; rows of -1 are padding).  Returns (cls_loss[1], reg_loss[1]) like FocalLoss.forward."""
    alpha, gamma = 0.25, 2.0
    a = anchors[0]
    aw, ah = a[:, 3] - a[:, 1], a[:, 2] - a[:, 0]
    acx, acy = a[:, 1] + 0.5 * aw, a[:, 0] + 0.5 * ah
    dtype = anchors.dtype
    c = classification.clamp(1e-4, 1.0 - 1e-4)                                     # [N, A, K]
    valid = annotations[:, :, 4] != -1                                             # [N, M]
    bx = annotations[:, :, :4]
    area = (bx[:, :, 2] - bx[:, :, 0]) * (bx[:, :, 3] - bx[:, :, 1])               # [N, M]
    iw = torch.min(a[None, :, None, 3], bx[:, None, :, 2]) - torch.max(a[None, :, None, 1], bx[:, None, :, 0])
    ih = torch.min(a[None, :, None, 2], bx[:, None, :, 3]) - torch.max(a[None, :, None, 0], bx[:, None, :, 1])
    iw, ih = iw.clamp(min=0), ih.clamp(min=0)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[None, :, None] + area[:, None, :] - iw * ih
    iou = iw * ih / ua.clamp(min=1e-8)                                             # [N, A, M]
    iou = torch.where(valid[:, None, :], iou, torch.full_like(iou, -1.0))          # padded boxes never match
    iou_max, iou_arg = iou.max(dim=2)                                              # first maximum, as torch.max on the compacted rows
    has_ann = valid.any(dim=1)                                                     # [N]
    pos = (iou_max >= 0.5) & has_ann[:, None]
    neg = (iou_max < 0.4) | ~has_ann[:, None]                                      # an image without boxes: every anchor is background
    npos = pos.sum(dim=1).to(dtype)
    assigned = torch.gather(annotations, 1, iou_arg[:, :, None].expand(-1, -1, 5))  # [N, A, 5]
    cls_id = assigned[:, :, 4].long().clamp(min=0)
    one_hot = (cls_id[:, :, None] == torch.arange(c.shape[2], device=c.device)) & pos[:, :, None]
    # targets: 1 at (positive, class), 0 at other positive/negative entries, -1 (ignored) elsewhere
    counted = pos[:, :, None] | neg[:, :, None]
    af = torch.where(one_hot, alpha, 1.0 - alpha)
    fw = af * torch.where(one_hot, 1.0 - c, c).pow(gamma)
    bce = -torch.where(one_hot, torch.log(c), torch.log(1.0 - c))
    cl = torch.where(counted, fw * bce, torch.zeros_like(c))
    cls_loss = cl.sum(dim=(1, 2)) / npos.clamp(min=1.0)
    gw, gh = assigned[:, :, 2] - assigned[:, :, 0], assigned[:, :, 3] - assigned[:, :, 1]
    gcx, gcy = assigned[:, :, 0] + 0.5 * gw, assigned[:, :, 1] + 0.5 * gh
    gw, gh = gw.clamp(min=1), gh.clamp(min=1)
    t = torch.stack(((gcy - acy) / ah, (gcx - acx) / aw, torch.log(gh / ah), torch.log(gw / aw)), dim=2)   # [N, A, 4]
    diff = (t - regression).abs()
    rl = torch.where(diff <= 1.0 / 9.0, 0.5 * 9.0 * diff.pow(2), diff - 0.5 / 9.0)
    rl = torch.where(pos[:, :, None], rl, torch.zeros_like(rl))
    reg_loss = rl.sum(dim=(1, 2)) / (4.0 * npos).clamp(min=1.0)                    # mean over (positives x 4); 0 when none
    return cls_loss.mean(dim=0, keepdim=True), reg_loss.mean(dim=0, keepdim=True)


def lane_cls_loss(cls_targets: torch.Tensor, cls_preds: torch.Tensor, negative_ratio: int = 15, alpha: float = 10.0):
    t = cls_targets[..., 1].reshape(-1)
    pmask = t > 0
    nmask = ~pmask
    fp, fn = pmask.float(), nmask.float()
    preds = cls_preds.reshape(-1, cls_preds.shape[-1])
    npos_f, nneg_f = fp.sum(), fn.sum()
    neg_num = torch.maximum(torch.minimum(npos_f * negative_ratio, nneg_f), torch.ones_like(npos_f)).to(torch.int64)
    pos_num = torch.maximum(npos_f, torch.ones_like(npos_f)).int()
    lsm = F.log_softmax(preds, dim=-1)
    fg, bg = lsm[..., 1], lsm[..., 0]
    # k-th smallest background log-prob among the negatives (find_k_th_small_in_a_tensor, lanedetect_loss.py:5-8):
    # sort with the positives pushed to +inf instead of compacting (no data-dependent shape)
    ordered, _ = torch.sort(torch.where(nmask, bg.detach(), torch.full_like(bg, float("inf"))))
    kth = ordered.gather(0, (neg_num - 1).reshape(1))[0]
    ohem = (bg <= kth).float() * fn
    pos = -torch.sum(alpha * fg * fp) / pos_num
    neg = -torch.sum(alpha * bg * ohem) / pos_num
    return pos, neg, pmask, pos_num


def lane_loc_loss(pmask, positive_num, loc_targets, loc_preds, alpha: float = 10.0, points_per_line: int = 160):
    """cal_loss_regress (lanedetect_loss.py:57-78), including its hard-coded points_per_line=160 default: the x10 weights land on
    columns 160/161 whatever the real layout is, and a tensor narrower than 162 columns raises IndexError exactly as the reference."""
    lp = loc_preds.reshape(-1, loc_preds.shape[-1])
    lt = loc_targets.reshape(-1, loc_targets.shape[-1])
    lw = torch.ones_like(lt)
    lw[..., points_per_line + 1] = alpha
    lw[..., points_per_line] = alpha
    valid_pts = lt != 0
    mask = lw * pmask.unsqueeze(-1).expand_as(lt).float() * valid_pts.float()
    d = lp - lt
    ad = d.abs()
    hub = torch.where(ad < 1, d * d / 2, ad - 0.5) * mask
    per = hub.sum(-1) / valid_pts.float().sum(-1).clamp(min=1)
    return per.sum() / positive_num
