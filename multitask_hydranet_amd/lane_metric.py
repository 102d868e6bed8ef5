"""Lane F1 for validation (reference: head_lane/lane_metric.py:166-440, driven by train.py:188,380-397,433).

`LaneMetric(method="f1_measure", iou_thresh=0.5, lane_width=30, thresh_list=[0.5])` keeps the reference's interface: call it with
`output=[dict(pr_result={"Lines": [...], "Shape": {...}}, gt_result={"Lines": [...], "Labels": [...], "Shape": {...}}), ...]`, then
`summary()`.  Per image: every lane is a natural cubic spline through its points, sampled at unit arc steps (spline_interp, restated with
the reference's arithmetic incl. its 1e-8 guards), drawn with width `lane_width` into a mask of the source-image size; IoU matrix of ground
truth x prediction masks; Hungarian assignment (scipy, as the reference); a matched pair with IoU > iou_thresh is a hit; precision / recall /
F1 over the epoch.  The rasterisation and the pixel counts run on the device (hn_lane_raster, hn_lane_iou: one launch each per image, exact
integer counts); the 1080 x 1920 masks never leave HBM.  cv2.line is restated (pixels within lane_width / 2 of the segment): parity with
OpenCV's thick-line fill at boundary pixels is unpinned (cv2 is absent), everything else is pinned by tests/golden/lane_metric.json."""
from __future__ import annotations

import sys
from typing import Dict, List, Sequence

import numpy as np
import torch

from ._lib import lib


def calc_params(lane: Sequence[dict]) -> List[dict]:
    """natural cubic spline over the chord-length parameter, one segment record per point pair (lane_metric.py:71-146)"""
    n = len(lane)
    if n < 2:
        return []
    xs = [p["x"] for p in lane]
    ys = [p["y"] for p in lane]
    if n == 2:
        h0 = np.sqrt((xs[0] - xs[1]) * (xs[0] - xs[1]) + (ys[0] - ys[1]) * (ys[0] - ys[1]))
        return [dict(a_x=xs[0], b_x=(xs[1] - xs[0]) / (h0 + 1e-8), c_x=0, d_x=0, a_y=ys[0], b_y=(ys[1] - ys[0]) / (h0 + 1e-8), c_y=0, d_y=0, h=h0)]
    h = [np.sqrt((xs[i] - xs[i + 1]) * (xs[i] - xs[i + 1]) + (ys[i] - ys[i + 1]) * (ys[i] - ys[i + 1])) for i in range(n - 1)]
    # Thomas sweep of the tridiagonal system for the second derivatives M_1 .. M_{n-2} (M_0 = M_{n-1} = 0)
    cs, dxs, dys = [], [], []
    for i in range(n - 2):
        a, b, c = h[i], 2 * (h[i] + h[i + 1]), h[i + 1]
        tx = 6 * ((xs[i + 2] - xs[i + 1]) / (h[i + 1] + 1e-8) - (xs[i + 1] - xs[i]) / (h[i] + 1e-8))
        ty = 6 * ((ys[i + 2] - ys[i + 1]) / (h[i + 1] + 1e-8) - (ys[i + 1] - ys[i]) / (h[i] + 1e-8))
        if i == 0:
            cs.append(c / (b + 1e-8))
            dxs.append(tx / (b + 1e-8))
            dys.append(ty / (b + 1e-8))
        else:
            base = b - a * cs[i - 1]
            cs.append(c / (base + 1e-8))
            dxs.append((tx - a * dxs[i - 1]) / (base + 1e-8))
            dys.append((ty - a * dys[i - 1]) / (base + 1e-8))
    mx, my = np.zeros(n), np.zeros(n)
    mx[n - 2], my[n - 2] = dxs[n - 3], dys[n - 3]
    for i in range(n - 4, -1, -1):
        mx[i + 1] = dxs[i] - cs[i] * mx[i + 2]
        my[i + 1] = dys[i] - cs[i] * my[i + 2]
    mx[0] = mx[-1] = my[0] = my[-1] = 0
    out = []
    for i in range(n - 1):
        out.append(dict(a_x=xs[i], b_x=(xs[i + 1] - xs[i]) / (h[i] + 1e-8) - (2 * h[i] * mx[i] + h[i] * mx[i + 1]) / 6, c_x=mx[i] / 2,
                        d_x=(mx[i + 1] - mx[i]) / (6 * (h[i] + 1e-8)),
                        a_y=ys[i], b_y=(ys[i + 1] - ys[i]) / (h[i] + 1e-8) - (2 * h[i] * my[i] + h[i] * my[i + 1]) / 6, c_y=my[i] / 2,
                        d_y=(my[i + 1] - my[i]) / (6 * (h[i] + 1e-8)), h=h[i]))
    return out


def spline_interp(*, lane: Sequence[dict], step_t=1) -> List[dict]:
    """lane_metric.py:45-68: samples of the spline at t = 0, step_t, ... < h per segment, then the last point"""
    if len(lane) < 2:
        return list(lane)
    pts = []
    for f in calc_params(lane):
        t = 0
        while t < f["h"]:
            pts.append({"x": f["a_x"] + f["b_x"] * t + f["c_x"] * t * t + f["d_x"] * t * t * t,
                        "y": f["a_y"] + f["b_y"] * t + f["c_y"] * t * t + f["d_y"] * t * t * t})
            t += step_t
    pts.append(lane[-1])
    return pts


def iou_matrix(gt_lanes: Sequence[Sequence[dict]], pr_lanes: Sequence[Sequence[dict]], height: int, width: int, lane_width: int,
               device=None) -> np.ndarray:
    """calc_iou (lane_metric.py:166-209) for every (ground truth, prediction) pair of one image, on the device.  Any number of lanes (the
    reference has no limit; a noisy early-epoch decode can keep more than the pair kernel's 32 x 32 block: it then runs per block)"""
    g, p = len(gt_lanes), len(pr_lanes)
    assert g > 0 and p > 0, (g, p)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    pts, seg_lane, seg_first = [], [], []
    for li, lane in enumerate(list(gt_lanes) + list(pr_lanes)):
        ip = spline_interp(lane=lane, step_t=1)
        base = len(pts)
        pts += [(int(q["x"]), int(q["y"])) for q in ip]                     # int(): truncation toward zero, as the reference's cv2.line arguments
        for i in range(len(ip) - 1):
            seg_lane.append(li)
            seg_first.append(base + i)
    masks = torch.zeros((g + p, height, width), dtype=torch.uint8, device=dev)
    if seg_lane:
        tp = torch.tensor(pts, dtype=torch.int32).to(dev)
        tl, tf = torch.tensor(seg_lane, dtype=torch.int32).to(dev), torch.tensor(seg_first, dtype=torch.int32).to(dev)
        lib().call("hn_lane_raster", tp.data_ptr(), tl.data_ptr(), tf.data_ptr(), len(seg_lane), int(lane_width), height, width, masks.data_ptr())
    B = 32                                                                  # hn_lane_iou's pair block (bit masks per pixel)
    if g <= B and p <= B:
        inter = torch.zeros((g, p), dtype=torch.int64, device=dev)
        area = torch.zeros((g + p,), dtype=torch.int64, device=dev)
        lib().call("hn_lane_iou", masks.data_ptr(), g, p, height * width, inter.data_ptr(), area.data_ptr())
    else:
        inter = torch.zeros((g, p), dtype=torch.int64, device=dev)
        area = torch.zeros((g + p,), dtype=torch.int64, device=dev)
        for g0 in range(0, g, B):
            for p0 in range(0, p, B):
                gc, pc = min(B, g - g0), min(B, p - p0)
                sub = torch.cat([masks[g0:g0 + gc], masks[g + p0:g + p0 + pc]]).contiguous()
                bi = torch.zeros((gc, pc), dtype=torch.int64, device=dev)
                ba = torch.zeros((gc + pc,), dtype=torch.int64, device=dev)
                lib().call("hn_lane_iou", sub.data_ptr(), gc, pc, height * width, bi.data_ptr(), ba.data_ptr())
                inter[g0:g0 + gc, p0:p0 + pc] = bi
                area[g0:g0 + gc] = ba[:gc]
                area[g + p0:g + p0 + pc] = ba[gc:]
    inter, area = inter.cpu().numpy().astype(np.float64), area.cpu().numpy().astype(np.float64)
    union = area[:g, None] + area[None, g:] - inter
    # (the reference sums uint8 masks of value 255: the factor cancels in the ratio; an empty union scores 0)
    return np.where(union > 0, inter / np.maximum(union, 1.0), 0.0)


def evaluate_core(*, gt_lanes, pr_lanes, gt_wh, pr_wh, hyperp) -> Dict[str, int]:
    """lane_metric.py:215-272: Hungarian assignment on 1 - IoU, hits = matched pairs with IoU > iou_thresh"""
    from scipy.optimize import linear_sum_assignment
    gt_num, pr_num, hit = len(gt_lanes), len(pr_lanes), 0
    if gt_num > 0 and pr_num > 0:
        iou = iou_matrix(gt_lanes, pr_lanes, hyperp["eval_height"], hyperp["eval_width"], hyperp["lane_width"])
        for gi, pi in zip(*linear_sum_assignment(1 - iou)):
            if iou[gi][pi] > hyperp["iou_thresh"]:
                hit += 1
    return dict(gt_num=gt_num, pr_num=pr_num, hit_num=hit)


class LaneMetricCore:
    """lane_metric.py:311-389"""

    def __init__(self, *, iou_thresh, lane_width, prob_thresh=None):
        self.eval_params = dict(iou_thresh=iou_thresh, lane_width=lane_width)
        self.prob_thresh = prob_thresh
        self.result_record: List[dict] = []

    def __call__(self, gt_result, pr_result, *args, **kwargs):
        gt_wh, pr_wh = gt_result["Shape"], pr_result["Shape"]
        gt_lanes = [line for line, _ in zip(gt_result["Lines"], gt_result["Labels"]) if len(line) > 0]
        pr_lanes = []
        for line in pr_result["Lines"]:
            if "score" in line:
                line = line["points"] if line["score"] > self.prob_thresh else []
            if len(line) > 0:
                pr_lanes.append(line)
        self.eval_params["eval_width"], self.eval_params["eval_height"] = gt_wh["width"], gt_wh["height"]
        self.result_record.append(evaluate_core(gt_lanes=gt_lanes, pr_lanes=pr_lanes, gt_wh=gt_wh, pr_wh=pr_wh, hyperp=self.eval_params))

    def reset(self):
        self.result_record = []

    def summary(self):
        hit = sum(r["hit_num"] for r in self.result_record)
        pr = sum(r["pr_num"] for r in self.result_record)
        gt = sum(r["gt_num"] for r in self.result_record)
        precision = hit / (pr + sys.float_info.epsilon)
        recall = hit / (gt + sys.float_info.epsilon)
        return dict(f1_measure=2 * precision * recall / (precision + recall + sys.float_info.epsilon), precision=precision, recall=recall)


class LaneMetric:
    """lane_metric.py:392-440"""

    def __init__(self, *, method, iou_thresh, lane_width, thresh_list=None):
        if method not in ("f1_measure", "precision", "recall"):
            raise NotImplementedError("method should be one of ['f1_measure', 'precision', 'recall']")
        self.method = method
        self.eval_params = dict(iou_thresh=iou_thresh, lane_width=lane_width)
        self.metric_handlers = [LaneMetricCore(**self.eval_params, prob_thresh=t) for t in thresh_list] if thresh_list is not None \
            else [LaneMetricCore(**self.eval_params, prob_thresh=None)]

    def __call__(self, output, *args, **kwargs):
        for handler in self.metric_handlers:
            for pair in output:
                handler(**pair)

    def reset(self):
        for handler in self.metric_handlers:
            handler.reset()

    def summary(self):
        return max(h.summary()[self.method] for h in self.metric_handlers)
