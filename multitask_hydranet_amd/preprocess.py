"""Input pre-processing on the device (reference: demo.py:26-50,186-196 and dataset/utility.py:213-227): BGR uint8 frame(s) -> RGB ->
bilinear resize to the network input -> /255 -> ImageNet mean / std -> fp32 NCHW, one kernel launch (hn_preprocess_bgr).
cv2 is a third-party dependency of the reference that is absent here: its 8-bit INTER_LINEAR resize is restated from OpenCV's published
fixed-point algorithm (11-bit coefficients); frames that already have the network size take no resize and are bit-exact against the
reference's numpy arithmetic (evaluated in float64, then cast to float32, exactly as demo.py does)."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import lib


def preprocess_bgr(frames, out_hw, device=None) -> torch.Tensor:
    """frames: uint8 [H, W, 3] or [N, H, W, 3] (numpy array or torch tensor, BGR as cv2.imread delivers) -> fp32 [N, 3, out_h, out_w]"""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    if frames.dim() == 3:
        frames = frames[None]
    assert frames.dtype == torch.uint8 and frames.shape[-1] == 3, (frames.dtype, frames.shape)
    dev = torch.device(device) if device is not None else (frames.device if frames.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    src = frames.to(dev).contiguous()
    n, hs, ws, _ = src.shape
    hd, wd = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((n, 3, hd, wd), device=dev, dtype=torch.float32)
    lib().call("hn_preprocess_bgr", src.data_ptr(), n, hs, ws, out.data_ptr(), hd, wd)
    return out
