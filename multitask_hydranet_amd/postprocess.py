"""Detection post-processing (head_detect/detection_loss.py:7-108 of the reference): box decode, clip, score threshold, per-class
greedy NMS.  Integer/index bookkeeping is bit-exact with the reference given identical fp inputs.  torchvision's batched_nms
(unpinned third party in the reference) is restated from its published semantics: descending stable score order, suppress when
IoU > threshold, IoU = inter / (a + b - inter), classes separated by a coordinate offset of class_id * (max_coord + 1).
Decode / clip / threshold / sort are device tensor ops; the O(K^2) suppression runs in HIP kernels for CUDA inputs (nms_device, SURVEY.md
section 8(f) row 1) and in numpy for CPU inputs (the form the CPU parity tests pin against the oracle)."""
from __future__ import annotations

import numpy as np
import torch


def decode_boxes(anchors: torch.Tensor, regression: torch.Tensor) -> torch.Tensor:
    yca = (anchors[..., 0] + anchors[..., 2]) / 2
    xca = (anchors[..., 1] + anchors[..., 3]) / 2
    ha = anchors[..., 2] - anchors[..., 0]
    wa = anchors[..., 3] - anchors[..., 1]
    w = regression[..., 3].exp() * wa
    h = regression[..., 2].exp() * ha
    yc = regression[..., 0] * ha + yca
    xc = regression[..., 1] * wa + xca
    return torch.stack([xc - w / 2.0, yc - h / 2.0, xc + w / 2.0, yc + h / 2.0], dim=2)


def nms_device(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """same contract as nms() with the O(K^2) part on the GPU (hn_nms_sorted: IoU bit-mask + one-wave scan); no host fallback"""
    from ._lib import lib
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].float().contiguous()
    k = b.shape[0]
    mask = torch.empty((lib().query("hn_nms_mask_words", k),), device=b.device, dtype=torch.int64)
    keep = torch.empty((k,), device=b.device, dtype=torch.uint8)
    lib().call("hn_nms_sorted", b.data_ptr(), k, float(thr), mask.data_ptr(), keep.data_ptr())
    return order[keep.bool()]


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.is_cuda and boxes.shape[0] <= 32768:
        return nms_device(boxes, scores, thr)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].float().cpu().numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(b.shape[0], dtype=bool)
    keep = []
    t = np.float32(thr)
    for i in range(b.shape[0]):
        if not alive[i]:
            continue
        keep.append(i)
        rest = slice(i + 1, None)
        iw = np.clip(np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]), 0, None).astype(np.float32)
        ih = np.clip(np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]), 0, None).astype(np.float32)
        inter = iw * ih
        alive[rest] &= ~(inter / (area[i] + area[rest] - inter) > t)
    return order[torch.as_tensor(keep, dtype=torch.int64, device=order.device)]


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    off = idxs.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes))
    return nms(boxes + off[:, None], scores, thr)


def postprocess(img_hw, anchors, regression, classification, threshold, iou_threshold):
    h, w = img_hw
    boxes = decode_boxes(anchors, regression)
    boxes[:, :, 0] = boxes[:, :, 0].clamp(min=0)
    boxes[:, :, 1] = boxes[:, :, 1].clamp(min=0)
    boxes[:, :, 2] = boxes[:, :, 2].clamp(max=w - 1)
    boxes[:, :, 3] = boxes[:, :, 3].clamp(max=h - 1)
    scores = torch.max(classification, dim=2, keepdim=True)[0]
    over = (scores > threshold)[:, :, 0]
    out = []
    empty = lambda: dict(rois=np.array(()), class_ids=np.array(()), scores=np.array(()))
    for i in range(regression.shape[0]):
        if over[i].sum() == 0:
            out.append(empty())
            continue
        cper = classification[i, over[i], :].permute(1, 0)
        bper = boxes[i, over[i], :]
        sper = scores[i, over[i], 0]
        sc, cl = cper.max(dim=0)
        keep = batched_nms(bper, sper, cl, iou_threshold)
        if keep.shape[0] == 0:
            out.append(empty())
        else:
            out.append(dict(rois=bper[keep, :].cpu().numpy(), class_ids=cl[keep].cpu().numpy(), scores=sc[keep].cpu().numpy()))
    return out
