"""Detection post-processing on the device (head_detect/detection_loss.py:7-108 of the reference: BBoxTransform, ClipBoxes, postprocess).

One HIP pipeline for the whole batch (hn_det_postprocess, csrc/hn_post.hip): box decode + clip + per-anchor max / arg-max class + score
threshold compaction + stable descending-score order + class-offset greedy NMS + gather.  No per-image host loop, no host NMS: the only
device-to-host traffic is the final result (kept boxes, classes, scores and two counters per image), which the reference's contract returns
as numpy arrays.  Index bookkeeping is exact; torchvision.ops.batched_nms (unpinned third party in the reference, absent here) is restated
from its published semantics -- stable descending score order, suppress when IoU > threshold, IoU = inter / (a + b - inter), classes
separated by an offset of class_id * (max kept coordinate + 1) -- see DESIGN.md ("parity unpinned for NMS").
There is no CPU path: CPU tensors are moved to the current HIP device first; without the HIP library this module raises.
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import lib

MAX_CAP = 32768


def decode_boxes(anchors: torch.Tensor, regression: torch.Tensor) -> torch.Tensor:
    """BBoxTransform.forward (detection_loss.py:7-33) as tensor ops, kept for callers that want the raw decoded boxes"""
    yca = (anchors[..., 0] + anchors[..., 2]) / 2
    xca = (anchors[..., 1] + anchors[..., 3]) / 2
    ha = anchors[..., 2] - anchors[..., 0]
    wa = anchors[..., 3] - anchors[..., 1]
    w = regression[..., 3].exp() * wa
    h = regression[..., 2].exp() * ha
    yc = regression[..., 0] * ha + yca
    xc = regression[..., 1] * wa + xca
    return torch.stack([xc - w / 2.0, yc - h / 2.0, xc + w / 2.0, yc + h / 2.0], dim=2)


def postprocess_device(img_hw, anchors, regression, classification, threshold, iou_threshold, cap: int = 4096):
    """device tensors in, device tensors out: dict(rois [N,cap,4], class_ids [N,cap] int64, scores [N,cap], kept [N] int32, total [N] int32).
    Row i < kept[n] of image n is its i-th detection in descending score order.  total[n] > cap means image n overflowed the capacity."""
    h, w = img_hw
    dev = regression.device if regression.is_cuda else torch.device("cuda", torch.cuda.current_device())
    reg = regression.detach().to(dev, torch.float32).contiguous()
    cls = classification.detach().to(dev, torch.float32).contiguous()
    anc = anchors.detach().to(dev, torch.float32)
    anc = (anc[0] if anc.dim() == 3 else anc).contiguous()
    n, a, k = cls.shape
    assert reg.shape == (n, a, 4) and anc.shape == (a, 4), (reg.shape, anc.shape)
    cap = int(min(max(cap, 1), MAX_CAP, a))
    ws = torch.empty((lib().query("hn_det_post_ws_bytes", n, cap),), device=dev, dtype=torch.uint8)
    out = dict(rois=torch.empty((n, cap, 4), device=dev, dtype=torch.float32), class_ids=torch.empty((n, cap), device=dev, dtype=torch.int64),
               scores=torch.empty((n, cap), device=dev, dtype=torch.float32), kept=torch.empty((n,), device=dev, dtype=torch.int32),
               total=torch.empty((n,), device=dev, dtype=torch.int32))
    lib().call("hn_det_postprocess", anc.data_ptr(), reg.data_ptr(), cls.data_ptr(), n, a, k, int(h), int(w), float(threshold),
               float(iou_threshold), cap, ws.data_ptr(), out["rois"].data_ptr(), out["class_ids"].data_ptr(), out["scores"].data_ptr(),
               out["kept"].data_ptr(), out["total"].data_ptr())
    out["cap"] = cap
    return out


def postprocess(img_hw, anchors, regression, classification, threshold, iou_threshold):
    """the reference's return contract: one dict(rois, class_ids, scores) of numpy arrays per image (empty arrays when nothing is kept)"""
    cap = 4096
    while True:
        res = postprocess_device(img_hw, anchors, regression, classification, threshold, iou_threshold, cap)
        total = res["total"].cpu().numpy()
        if int(total.max(initial=0)) <= res["cap"]:
            break
        if res["cap"] >= min(MAX_CAP, regression.shape[1]):
            raise RuntimeError(f"{int(total.max())} anchors over the score threshold in one image: the device NMS holds at most {MAX_CAP}")
        cap = min(MAX_CAP, max(2 * cap, int(total.max())))
    kept = res["kept"].cpu().numpy()
    rois, cids, scores = res["rois"].cpu().numpy(), res["class_ids"].cpu().numpy(), res["scores"].cpu().numpy()
    out = []
    for i in range(len(kept)):
        k = int(kept[i])
        if k == 0:
            out.append(dict(rois=np.array(()), class_ids=np.array(()), scores=np.array(())))
        else:
            out.append(dict(rois=rois[i, :k].copy(), class_ids=cids[i, :k].copy(), scores=scores[i, :k].copy()))
    return out
