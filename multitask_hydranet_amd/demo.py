"""demo.py of the reference (model/demo.py:52-261) on the HIP path: frame -> pre-processing -> HydraNet forward -> the three decodes.

    python -m multitask_hydranet_amd.demo [--cfg cfgs/hydranet_big.yml] [--weights ckpt.pth] [--frames frames.npy] [--out demo_out]

What is kept: the configuration handling (network input size, which heads run, lane codec geometry, colour table), `module.`-prefixed
checkpoints (deparallel_model, demo.py:33-50), the per-frame sequence BGR -> RGB -> resize -> imagenet_normalize -> forward ->
laneheader.decode + scale_to_org, segheader.decode, detectheader.decode, and the thresholds demo.py hard-codes (lane 0.90 / 80, detection
0.4 / 0.3).  Every stage runs on the device (hn_preprocess_bgr, the folded-BatchNorm inference forward, hn_lane_decode_nms,
hn_seg_overlay, hn_det_postprocess).
What is not: cv2 (absent from this image).  Frames come from a .npy array [T, H, W, 3] uint8 BGR or are synthesised; instead of a window
/ video writer the blended frames are returned (and written as .npy by the command line); the line / box / text drawing of
laneheader.visual and detectheader.display is visualisation outside SURVEY 8 -- the decoded lanes and boxes are returned as data."""
from __future__ import annotations

import argparse
import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch
import yaml

SEG_CLASS_COLOR_ID = {0: (0, 0, 0), 1: (128, 0, 128), 2: (255, 255, 255), 3: (0, 255, 255), 4: (0, 255, 0)}    # demo.py:91-96


class Demo:
    """the state demo.py sets up before its frame loop (demo.py:68-134)"""

    def __init__(self, cfgs: dict, weights: Optional[str] = None, device="cuda:0", fold_batchnorm: bool = True):
        from . import HydraNet
        from .lane_codec import LaneCodec
        self.cfgs = cfgs
        dl = cfgs["dataloader"]
        self.net_w, self.net_h = dl["network_input_width"], dl["network_input_height"]
        tr = cfgs["train"]
        self.train_detect, self.train_seg, self.train_lane = tr["train_detect"], tr["train_seg"], tr["train_lane"]
        self.obj_list = cfgs["detection"]["class_list"][1:]
        lane = cfgs["lane"]
        self.lane_coder = LaneCodec(input_width=self.net_w, input_height=self.net_h, anchor_stride=lane["anchor_stride"],
                                    points_per_line=int(self.net_h / lane["interval"]), do_interpolate=lane["interpolate"],
                                    anchor_lane_num=lane["anchor_lane_num"], scale_invariance=lane["scale_invariance"])
        self.colors = dict(SEG_CLASS_COLOR_ID)
        self.lane_conf, self.lane_nms = 0.90, 80          # demo.py:212-213 (hard-coded there, not the cfg's values)
        self.det_conf, self.det_iou = 0.4, 0.3            # demo.py:241
        self.device = torch.device(device)
        self.net = HydraNet(cfgs=cfgs, onnx_export=False).to(self.device)
        if weights:
            self.net.load_state_dict(torch.load(weights, map_location="cpu"))    # (`module.` prefixes are stripped by load_state_dict)
        self.net.eval()
        if fold_batchnorm:
            self.net.prepare_inference()

    @torch.no_grad()
    def process(self, input_img: np.ndarray) -> Dict[str, object]:
        """one iteration of demo.py's loop (demo.py:176-246) for one BGR frame [H, W, 3] uint8"""
        from .preprocess import preprocess_bgr
        net = self.net
        org_h, org_w = input_img.shape[:2]
        org_size = (org_w, org_h)
        tic = time.time()
        img = preprocess_bgr(input_img, (self.net_h, self.net_w), device=self.device)       # demo.py:186-196
        outputs = net(img)                                                                    # demo.py:201
        res: Dict[str, object] = {"org_size": org_size}
        imgs: List[np.ndarray] = [input_img]
        if self.train_lane:                                                                   # demo.py:209-228
            cls_preds, loc_preds = outputs["lane"]["predict_cls"], outputs["lane"]["predict_loc"]
            lanes = []
            for b in range(len(imgs)):
                nms_set = net.laneheader.decode(cls_preds[b], loc_preds[b], self.lane_coder, self.lane_conf, self.lane_nms, False)
                lanes.append(net.laneheader.scale_to_org(nms_set, self.net_w, self.net_h, org_size[0], org_size[1])["Lines"])
            res["lanes"] = lanes
        if self.train_seg:                                                                    # demo.py:230-233
            imgs = net.segheader.decode(imgs, outputs["seg"], org_size, self.colors)
        if self.train_detect:                                                                 # demo.py:236-242
            det = outputs["detection"]
            res["detections"] = net.detectheader.decode(img, det["regression"], det["classification"], det["anchors"], conf_thres=self.det_conf,
                                                        iou_thres=self.det_iou)
        torch.cuda.synchronize(self.device)
        res["visual"] = imgs[0]
        res["ms"] = 1000.0 * (time.time() - tic)
        return res


def synthetic_frames(n: int, h: int = 1080, w: int = 1920, seed: int = 0) -> np.ndarray:
    """road-like BGR frames: sky / ground gradient, two lane-ish bright bands, a few boxes, noise"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        base = np.where(yy < h * 0.45, 150 + 60 * yy / h, 70 + 40 * yy / h)
        img = np.stack([base * 1.05, base, base * 0.9], -1)
        for off in (-0.18 + 0.02 * i, 0.2 - 0.01 * i):
            cx = w * (0.5 + off * (yy - h * 0.45) / (h * 0.55))
            img[(np.abs(xx - cx) < 6 + 10 * yy / h) & (yy > h * 0.45)] = 235
        for _ in range(4):
            x0, y0 = rs.randint(0, w - 200), rs.randint(int(h * 0.4), h - 150)
            img[y0:y0 + rs.randint(40, 140), x0:x0 + rs.randint(60, 200)] = rs.randint(20, 230, size=3)
        out[i] = np.clip(img + rs.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    return out


def main(argv=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--cfg", default=os.path.join(root, "cfgs", "hydranet_big.yml"))
    ap.add_argument("--weights", default=None, help="checkpoint written by train.py (module.-prefixed keys are accepted); random init without")
    ap.add_argument("--frames", default=None, help=".npy uint8 [T, H, W, 3] BGR frames; synthetic 1080p frames without")
    ap.add_argument("--count", type=int, default=4)
    ap.add_argument("--out", default=None, help="directory for frame_%%04d.npy (blended frames) and results.json")
    args = ap.parse_args(argv)
    cfgs = yaml.safe_load(open(args.cfg))
    torch.manual_seed(0)
    demo = Demo(cfgs, args.weights)
    if not args.weights:
        # random initialisation: every anchor scores ~0.5, far more candidates than any real frame has (the device NMS holds 32 768)
        print("no --weights: random initialisation, detection threshold raised to 0.95 for this run")
        demo.det_conf = 0.95
    frames = np.load(args.frames) if args.frames else synthetic_frames(args.count)
    if args.out:
        os.makedirs(args.out, exist_ok=True)
    summary = []
    for t, frame in enumerate(frames):
        r = demo.process(frame)
        nd = sum(len(d["rois"]) for d in r.get("detections", []) or [])
        nl = sum(len(l) for l in r.get("lanes", []))
        print("frame %d: total process time is %i ms, %d lanes, %d boxes" % (t, r["ms"], nl, nd))
        summary.append({"frame": t, "ms": r["ms"], "lanes": nl, "boxes": nd})
        if args.out:
            np.save(os.path.join(args.out, "frame_%04d.npy" % t), r["visual"])
    if args.out:
        import json
        json.dump(summary, open(os.path.join(args.out, "results.json"), "w"), indent=1)
    return summary


if __name__ == "__main__":
    main()
