"""Lane decode + lane NMS on the device (reference: head_lane/lanedetect.py:103-125 LaneHeader.decode / scale_to_org,
head_lane/lane_codec.py:25-51,116-219 LaneCodec.decode_lane, head_lane/lane_codec_utils.py:6-64,185-282,487-543).

`decode(predict_cls, predict_loc, pointlane, conf_thres, nms_line_thres, use_mean)` keeps the reference's signature and return type (a list
of Lane objects in descending-probability order); the per-anchor point walk, the stable probability sort and the greedy distance NMS run
in ONE kernel launch for the whole batch (hn_lane_decode_nms, one workgroup per image).  `pointlane` may be the reference's own LaneCodec
object or this module's LaneCodec (only the geometry fields are read).  scale_to_org / order_lane_x_axis / convert_lane_to_dict are the
host-side bookkeeping on the handful of surviving lanes, restated.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from ._lib import lib


class Point:
    def __init__(self, x=0, y=0):
        self.x, self.y = x, y

    def __repr__(self):
        return "{}, {}".format(self.x, self.y)


class Lane:
    def __init__(self, prob=0, start_pos=0, end_pos=0, anchor_x=0, anchor_y=0, type=0, lane=None):
        self.prob, self.start_pos, self.end_pos = prob, start_pos, end_pos
        self.lane = lane if lane is not None else np.array([])
        self.idx, self.ax, self.ay, self.type = 0, anchor_x, anchor_y, type

    def __lt__(self, other):
        return self.prob > other.prob


class LaneCodec:
    """geometry of the reference's LaneCodec (lane_codec.py:25-51); encode_lane (ground-truth generation, scipy splines) is data-pipeline
    work outside the hot path"""

    def __init__(self, input_width, input_height, anchor_stride, points_per_line, do_interpolate=False, anchor_lane_num=1,
                 scale_invariance=True):
        self.input_width, self.input_height, self.stride = input_width, input_height, anchor_stride
        self.feature_width, self.feature_height = int(input_width / anchor_stride), int(input_height / anchor_stride)
        self.points_per_line = points_per_line
        self.pt_nums_single_lane = 2 * points_per_line + 2
        self.points_per_anchor = points_per_line / self.feature_height
        self.interval = float(input_height) / points_per_line
        self.feature_size = self.feature_width * self.feature_height
        self.step_w = self.step_h = anchor_stride
        self.anchor_lane_num, self.interpolation, self.scale_invariance = anchor_lane_num, do_interpolate, scale_invariance

    def decode_lane(self, predict_type, predict_loc, exist_threshold=0.5, margin_width=100.0):
        """candidates before NMS (LaneCodec.decode_lane takes POST-softmax probabilities): runs the device kernel with the NMS disabled"""
        logits = torch.log(predict_type.clamp_min(1e-38))
        return _decode_batch(logits[None], predict_loc[None], self, exist_threshold, -1.0, False, margin_width, keep_all=True)[0]


def _decode_batch(cls, loc, codec, conf_thres, nms_thres, use_mean, margin=100.0, keep_all=False) -> List[List[Lane]]:
    assert getattr(codec, "scale_invariance", True), "only the scale-invariant location encoding of the shipped cfgs is on the device path"
    dev = cls.device if cls.is_cuda else torch.device("cuda", torch.cuda.current_device())
    cls = cls.detach().to(dev, torch.float32).contiguous()
    loc = loc.detach().to(dev, torch.float32).contiguous()
    n, hw, _ = cls.shape
    W, H, stride, ppl = int(codec.input_width), int(codec.input_height), int(codec.step_w), int(codec.points_per_line)
    assert hw == (W // stride) * (H // stride) and loc.shape == (n, hw, 2 * ppl + 2), (cls.shape, loc.shape)
    X = torch.empty((n, hw, ppl), device=dev, dtype=torch.float32)
    prob = torch.empty((n, hw), device=dev, dtype=torch.float32)
    ints = torch.empty((4, n, hw), device=dev, dtype=torch.int32)
    counts = torch.empty((n,), device=dev, dtype=torch.int32)
    lib().call("hn_lane_decode_nms", cls.data_ptr(), loc.data_ptr(), n, W, H, stride, ppl, float(conf_thres), float(nms_thres), 1 if use_mean else 0,
               float(margin), X.data_ptr(), prob.data_ptr(), ints[0].data_ptr(), ints[1].data_ptr(), ints[2].data_ptr(), ints[3].data_ptr(),
               counts.data_ptr())
    counts, prob, ints, X = counts.cpu().numpy(), prob.cpu().numpy(), ints.cpu().numpy(), X.cpu().numpy()
    start, end, order, keep = ints
    fw = W // stride
    out = []
    for i in range(n):
        lanes = []
        for j in range(int(counts[i])):
            if not (keep_all or keep[i, j]):
                continue
            a = int(order[i, j])
            s, e = int(start[i, a]), int(end[i, a])
            pts = np.array([Point(X[i, a, p], H - 1 - p * codec.interval) for p in range(s, e)])
            ah, aw = divmod(a, fw)
            lanes.append(Lane(prob[i, a], s, e, (1.0 * aw + 0.5) * stride, (1.0 * ah + 0.5) * stride, 1, pts))
        if keep_all:                       # decode_lane returns raster order
            lanes.sort(key=lambda l: (l.ay, l.ax))
        out.append(lanes)
    return out


def decode(predict_cls, predict_loc, pointlane, conf_thres=0.5, nms_line_thres=100, use_mean=False):
    """LaneHeader.decode (lanedetect.py:103-116) for ONE image: predict_cls [hw, 2] logits, predict_loc [hw, 2*ppl+2]"""
    return _decode_batch(predict_cls[None], predict_loc[None], pointlane, conf_thres, nms_line_thres, use_mean)[0]


def decode_batch(predict_cls, predict_loc, pointlane, conf_thres=0.5, nms_line_thres=100, use_mean=False):
    """the same for a whole batch [N, hw, 2] / [N, hw, L] in one launch"""
    return _decode_batch(predict_cls, predict_loc, pointlane, conf_thres, nms_line_thres, use_mean)


# ---- host-side bookkeeping on the surviving lanes (lane_codec_utils.py:66-124,185-282) ----------------------------------------------
def _calc_y_cross(p1, p2, y):
    if abs(p1.y - p2.y) < 1e-6:
        return -1
    k = (p1.x - p2.x) / (p1.y - p2.y)
    return k * y + (p1.x - k * p1.y)


class _LaneWithCrossK:
    def __init__(self, lane_, idx_in, y_in):
        self.lane, self.idx, self.y = lane_, idx_in, y_in
        pts = lane_.lane
        if pts[1].y < pts[0].y:
            self.k = (pts[1].x - pts[0].x) / (pts[1].y - pts[0].y)
            self.cross_x = _calc_y_cross(pts[0], pts[1], y_in)
        elif pts[1].y > pts[0].y:
            self.k = (pts[-1].x - pts[-2].x) / (pts[-1].y - pts[-2].y)
            self.cross_x = _calc_y_cross(pts[-2], pts[-1], y_in)
        else:
            self.k = 1000
            self.cross_x = _calc_y_cross(pts[-2], pts[-1], y_in)

    def __lt__(self, other):
        if abs(self.cross_x - other.cross_x) > 2.0:
            return self.cross_x < other.cross_x
        if self.lane.lane[1].y < self.lane.lane[0].y:
            return self.lane.lane[-1].x < other.lane.lane[-1].x
        return self.lane.lane[0].x < other.lane.lane[0].x


def order_lane_x_axis(lane_set, h):
    if len(lane_set) == 0:
        return list()
    srt = sorted(_LaneWithCrossK(l, i, h - 1.0) for i, l in enumerate(lane_set))
    right = len(srt)
    for i, l in enumerate(srt):
        if l.k > 0:
            right = i
            break
    idx = [None] * len(srt)
    for j, i in enumerate(range(right - 1, -1, -1)):
        idx[i] = -1 - j
    for j, i in enumerate(range(right, len(srt))):
        idx[i] = 1 + j
    out = []
    for i, l in enumerate(srt):
        l.lane.idx = idx[i]
        out.append(l.lane)
    return out


def convert_lane_to_dict(lane_set, sx, sy):
    lines = []
    for l in lane_set:
        if l.prob < 0.01:
            continue
        lines.append({"score": l.prob, "points": [{"x": p.x * sx, "y": p.y * sy} for p in l.lane]})
    return {"Lines": lines}


def scale_to_org(lane_nms_set, net_input_width, net_input_height, org_width, org_height):
    """LaneHeader.scale_to_org, lanedetect.py:118-124"""
    ordered = order_lane_x_axis(list(lane_nms_set), net_input_height)
    return convert_lane_to_dict(ordered, org_width / net_input_width, org_height / net_input_height)
