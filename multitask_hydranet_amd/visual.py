"""Segmentation overlay on the device: SegmentHeader.decode (reference head_seg/segmentation.py:107-125; one of demo.py's three decodes,
demo.py:232-235).  arg-max (hn_argmax / already fused into the deploy forward) -> colour LUT -> 8-bit bilinear resize to the source frame
-> saturating blend 0.8 * frame + 0.5 * colours, one launch (hn_seg_overlay), one D2H of the blended frames.

cv2 is absent from this image, so the two cv2 calls are restated from OpenCV's published arithmetic (parity UNPINNED, checked against the
oracle's restatement): `cv2.resize(vis_seg, org_size, cv2.INTER_NEAREST)` passes the flag as the `dst` argument -- the effective
interpolation is the default INTER_LINEAR in its 11-bit fixed-point form; `cv2.addWeighted` on uint8 = float32 arithmetic, round half to
even, saturate.  LaneHeader.visual / DetectionHeader.display (cv2 line / box / text drawing) stay outside the scope."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from ._lib import lib


def colour_lut(vis_color_id: Dict[int, Sequence[int]], device) -> torch.Tensor:
    """{class id: (c0, c1, c2)} -> uint8 [max id + 1, 3]; ids without an entry stay black (the reference paints onto zeros)"""
    n = max(int(k) for k in vis_color_id) + 1
    lut = np.zeros((n, 3), np.uint8)
    for k, c in vis_color_id.items():
        lut[int(k)] = np.asarray(c, dtype=np.int64).astype(np.uint8)[:3]
    return torch.from_numpy(lut).to(device)


def seg_overlay(frames: torch.Tensor, mask: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
    """frames uint8 [N, Ho, Wo, 3] (device), mask int64 [N, H, W], lut uint8 [ncls, 3] -> blended uint8 [N, Ho, Wo, 3] on the device"""
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    assert mask.dtype == torch.int64 and mask.dim() == 3 and mask.shape[0] == frames.shape[0]
    frames, mask, lut = frames.contiguous(), mask.contiguous(), lut.contiguous()
    out = torch.empty_like(frames)
    n, h, w = mask.shape
    lib().call("hn_seg_overlay", mask.data_ptr(), n, h, w, lut.data_ptr(), lut.shape[0], frames.data_ptr(), out.data_ptr(), frames.shape[1],
               frames.shape[2])
    return out


def seg_decode(imgs, masks, org_size, vis_color_id) -> List[np.ndarray]:
    """SegmentHeader.decode(imgs, masks, org_size, vis_color_id), same arguments and return type as the reference:
    imgs: list of uint8 HWC frames (all of org_size = (width, height)); masks: seg logits [N, C, H, W] (any float dtype / memory format)
    or an int64 class-id mask [N, H, W] (the deploy forward's first output); returns the list of blended frames (numpy, uint8)."""
    from . import ops as K
    if not torch.is_tensor(masks):
        masks = torch.as_tensor(masks)
    dev = masks.device if masks.is_cuda else torch.device("cuda", torch.cuda.current_device())
    masks = masks.to(dev)
    if masks.dim() == 4:
        masks = K.argmax_channels(masks.detach().float())
    ow, oh = int(org_size[0]), int(org_size[1])
    frames = np.stack([np.ascontiguousarray(im) for im in imgs], 0)
    assert frames.dtype == np.uint8 and frames.shape[1:] == (oh, ow, 3), (frames.dtype, frames.shape, org_size)
    out = seg_overlay(torch.from_numpy(frames).to(dev), masks, colour_lut(vis_color_id, dev)).cpu().numpy()
    return [out[i] for i in range(out.shape[0])]
