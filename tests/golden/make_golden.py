#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (/root/reference, read-only) is imported with three throw-away shims that live in a
temp dir created here (never committed, never shipped):
  * stub modules for absent third-party imports (torchvision.ops.boxes, cv2, webcolors, imgaug ...),
  * Tensor.cuda / Module.cuda patched to identity (the reference hard-codes .cuda()),
  * sys.dont_write_bytecode (reference tree is read-only).
Only *data* is written: inputs, the reference's state_dict, and the reference's outputs / losses /
gradients.  Run:  python tests/golden/make_golden.py
"""
import functools
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/model"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import hydranet_oracle as O  # noqa: E402  (used for the NMS stub + synthetic batch only)


def _install_shims():
    d = tempfile.mkdtemp(prefix="refstubs_")
    os.makedirs(os.path.join(d, "torchvision", "ops"))
    open(os.path.join(d, "torchvision", "__init__.py"), "w").write("")
    open(os.path.join(d, "torchvision", "ops", "__init__.py"), "w").write("")
    open(os.path.join(d, "torchvision", "ops", "boxes.py"), "w").write(
        "import sys\nsys.path.insert(0, %r)\nfrom oracle.hydranet_oracle import nms_greedy as nms, batched_nms\n" % ROOT)
    # cv2 stand-in: the constant model.py reads, and -- for head_lane/lane_metric.py -- line / bitwise_or (cv2.line = the oracle's restated
    # thick-line rasteriser: parity with OpenCV itself stays unpinned, everything around it is the reference's own code)
    open(os.path.join(d, "cv2.py"), "w").write(
        "INTER_NEAREST = 0\nimport sys\nsys.path.insert(0, %r)\nfrom oracle.hydranet_oracle import cv2_line as line\n"
        "def bitwise_or(a, b):\n    return a | b\n" % ROOT)
    open(os.path.join(d, "webcolors.py"), "w").write(
        "class _C:\n    red = green = blue = 0\n\ndef name_to_rgb(name):\n    return _C()\n")
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _np(t):
    return t.detach().cpu().numpy()


def tiny_fixture(cfg_name="hydranet_tiny.yml", out_name="tiny_hydranet.npz"):
    """tiny 5-stage cfg (default) or the tiny 4-stage / focal cfg (hydranet_tiny4.yml: the family of the reference's
    hydranet_joint_small_backbone.yml -- net/bifpn.py:158-160 p5_to_p6 path, segmentation_loss.py:31-46 focal loss)"""
    from model import HydraNet  # the reference
    from head_lane.lanedetect_loss import cal_loss_regress

    cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", cfg_name)))
    h = w = 128
    n = 2
    torch.manual_seed(0)
    net = HydraNet(cfgs)
    # perturb BN affine / biases / fusion weights away from their trivial init so parity is meaningful
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            if k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("running_mean") or k.endswith(".bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1 and "_w" in k.split(".")[-1]:           # BiFPN fusion weights (one negative -> relu)
                v.copy_(torch.rand(v.shape, generator=g) + 0.1)
            elif v.dim() == 1:                                       # BN gamma
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
        net.state_dict()["neck.bifpn.1.p4_w2"][1] = -0.3
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    ppl = h // cfgs["lane"]["interval"]
    net.loss_reg = functools.partial(cal_loss_regress, points_per_line=ppl)   # SURVEY section 0 #3 deviation

    batch = O.synthetic_batch(cfgs, n, h, w, seed=1)
    net.train()
    x = batch["image"]
    feats = net.backbone(x)
    fused = net.neck(feats)
    torch.manual_seed(0)
    # second, clean forward for outputs (BN running stats get updated twice; record after this pass)
    net.load_state_dict(sd0)
    out = net(x)
    ld = net.cal_loss(out, batch)
    s, d, l = cfgs["segment"], cfgs["detection"], cfgs["lane"]
    total = ld["loss_seg"] * s["segment_weight"] \
        + (ld["loss_det_cls"] * d["loss_cls_weight"] + ld["loss_det_reg"] * d["loss_reg_weight"]) * d["detection_weight"] \
        + (ld["loss_lane_cls_pos"] + ld["loss_lane_cls_neg"] + ld["loss_lane_loc"]) * l["lane_weight"]
    total.backward()
    res = {}
    for k, v in sd0.items():
        res["sd/" + k] = _np(v)
    for k, v in net.state_dict().items():
        if "running_" in k or "num_batches" in k:
            res["sd_after/" + k] = _np(v)
    for k, v in batch.items():
        res["in/" + k] = _np(v)
    for i, f in enumerate(feats):
        res[f"feat/{i}"] = _np(f)
    for i, f in enumerate(fused):
        res[f"fused/{i}"] = _np(f)
    res["out/seg"] = _np(out["seg"])
    res["out/anchors"] = _np(out["detection"]["anchors"])
    res["out/regression"] = _np(out["detection"]["regression"])
    res["out/classification"] = _np(out["detection"]["classification"])
    res["out/lane_cls"] = _np(out["lane"]["predict_cls"])
    res["out/lane_loc"] = _np(out["lane"]["predict_loc"])
    for k, v in ld.items():
        res["loss/" + k] = _np(v)
    res["loss/total"] = _np(total)
    res["meta/lane_points_per_line"] = np.array(ppl)
    nograd = []
    for k, p in net.named_parameters():
        if p.grad is None:
            nograd.append(k)
        else:
            res["grad/" + k] = _np(p.grad)
    res["meta/nograd"] = np.array(nograd)
    # deploy tuple + eval-mode forward
    net.eval()
    with torch.no_grad():
        dep = net(x, "deploy")
    res["deploy/seg_argmax"] = _np(dep[0]).astype(np.int64)
    res["deploy/regression"] = _np(dep[2])
    res["deploy/classification"] = _np(dep[3])
    res["deploy/lane_cls"] = _np(dep[4])
    # postprocess through the reference's own code (NMS = restated stub; parity unpinned there)
    from head_detect.detection import DetectionHeader
    pp = DetectionHeader.decode(x, dep[2], dep[3], dep[1], conf_thres=float(dep[3].max()) * 0.9, iou_thres=0.3)
    res["deploy/pp_thresh"] = np.array(float(dep[3].max()) * 0.9, dtype=np.float64)
    for i, o in enumerate(pp):
        res[f"deploy/pp{i}/rois"] = np.asarray(o["rois"], dtype=np.float32)
        res[f"deploy/pp{i}/class_ids"] = np.asarray(o["class_ids"], dtype=np.int64)
        res[f"deploy/pp{i}/scores"] = np.asarray(o["scores"], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, out_name), **res)
    print("%s: %d arrays, losses:" % (out_name, len(res)), {k: float(v) for k, v in ld.items()}, "nograd", nograd)


def loss_kats():
    """Known-answer tests built from the reference's own smoke inputs."""
    from head_detect.detection_loss import FocalLoss, calc_iou
    from head_lane.lanedetect_loss import cal_loss_cls, cal_loss_regress
    from head_seg.segmentation_loss import CrossEntropyLoss
    from head_detect.detection import Anchors

    res = {}
    g = torch.Generator().manual_seed(7)
    # --- seg: gt = ones (segmentation.py:226) and random gt with ignore pixels, top-k and focal variants
    logits = torch.randn(2, 5, 32, 64, generator=g)
    gt1 = torch.ones(2, 32, 64).long()
    gt2 = torch.randint(0, 5, (2, 32, 64), generator=g)
    gt2[0, :3] = 255
    cw = [0.1, 0.5, 1.0, 5.0, 5.0]
    res["seg/logits"], res["seg/gt_ones"], res["seg/gt_rand"] = _np(logits), _np(gt1), _np(gt2)
    for name, kw in (("topk", dict(use_top_k=True, top_k_ratio=0.3, use_focal=False)),
                     ("plain", dict(use_top_k=False, top_k_ratio=1.0, use_focal=False)),
                     ("focal", dict(use_top_k=False, top_k_ratio=0.3, use_focal=True))):
        m = CrossEntropyLoss(class_weights=torch.tensor(cw), **kw)
        res[f"seg/{name}/ones"] = _np(m(logits, gt1))
        if name != "focal":                                          # focal path cannot take ignore=255 (scatter)
            res[f"seg/{name}/rand"] = _np(m(logits, gt2))
    # --- det: annotations = ones([B,16,5]) (detection.py:351), random boxes, and an empty image
    cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
    x = torch.zeros(3, 3, 128, 256)
    anc = Anchors(anchor_scale=2.0, pyramid_levels=[3, 4, 5, 6, 7], scales=[2 ** 0.0, 2 ** 0.333, 2 ** 0.667],
                  ratio=[(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)])(x, torch.float32)
    a = anc.shape[1]
    cls = torch.rand(3, a, 9, generator=g) * 0.2
    reg = torch.randn(3, a, 4, generator=g) * 0.1
    ann = O.synthetic_batch(cfgs, 3, 128, 256, seed=3)["gt_det"]
    ann[:, :, :4] = ann[:, :, :4]
    ann[1, 5:] = -1
    ann[2] = -1
    ones = torch.ones(3, 16, 5)
    res["det/anchors"], res["det/cls"], res["det/reg"], res["det/ann"] = _np(anc), _np(cls), _np(reg), _np(ann)
    for name, an in (("rand", ann), ("ones", ones)):
        cl, rl = FocalLoss()(cls, reg, anc, an)
        res[f"det/{name}/cls_loss"], res[f"det/{name}/reg_loss"] = _np(cl), _np(rl)
    res["det/iou"] = _np(calc_iou(anc[0, ::97], ann[0, :, :4]))
    # --- lane: cls_targets = ones; [:, 0:40, 1] = 0; loc_targets = ones (lanedetect.py:268-270) at 640x640
    hw, width = 400, 162
    cp = torch.randn(2, hw, 2, generator=g)
    lp = torch.randn(2, hw, width, generator=g)
    ct = torch.ones(2, hw, 2)
    ct[:, 0:40, 1] = 0
    lt = torch.ones(2, hw, width)
    res["lane/cls_pred"], res["lane/loc_pred"] = _np(cp), _np(lp)
    pos, neg, pmask, pnum = cal_loss_cls(ct, cp)
    res["lane/smoke/pos"], res["lane/smoke/neg"], res["lane/smoke/pnum"] = _np(pos), _np(neg), _np(pnum)
    res["lane/smoke/loc_default160"] = _np(cal_loss_regress(pmask, pnum, lt, lp))
    res["lane/smoke/loc_ppl80"] = _np(cal_loss_regress(pmask, pnum, lt, lp, points_per_line=80))
    b = O.synthetic_batch(cfgs, 2, 640, 640, seed=5)
    pos, neg, pmask, pnum = cal_loss_cls(b["gt_cls"], cp)
    res["lane/synth/pos"], res["lane/synth/neg"] = _np(pos), _np(neg)
    res["lane/synth/loc_default160"] = _np(cal_loss_regress(pmask, pnum, b["gt_loc"], lp))
    allbg = torch.zeros(2, hw, 2)
    allbg[..., 0] = 1
    pos, neg, pmask, pnum = cal_loss_cls(allbg, cp)
    res["lane/allbg/pos"], res["lane/allbg/neg"], res["lane/allbg/pnum"] = _np(pos), _np(neg), _np(pnum)
    # --- anchor tables (exact fp32 equality): sha256 + a strided sample
    for (h, w) in ((640, 640), (512, 1024)):
        t = _np(Anchors(anchor_scale=2.0, pyramid_levels=[3, 4, 5, 6, 7],
                        scales=[2 ** 0.0, 2 ** 0.333, 2 ** 0.667],
                        ratio=[(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)])(torch.zeros(1, 3, h, w), torch.float32))[0]
        res[f"anchors/{h}x{w}/shape"] = np.array(t.shape)
        res[f"anchors/{h}x{w}/sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest())
        res[f"anchors/{h}x{w}/sample"] = t[::997]
    # --- width derivation for the three shipped cfgs (INT equality)
    from net.regnet import RegNetY
    import net.anynet as anynet
    captured = {}
    orig = anynet.AnyNetX.__init__

    def spy(self, nb, bw, br, gw, stride, se):
        captured["v"] = (list(map(int, nb)), list(map(int, bw)), list(map(int, gw)))
        raise RuntimeError("captured")
    anynet.AnyNetX.__init__ = spy
    for name in ("hydranet_joint_big_backbone", "hydranet_joint_big_backbone_interview", "hydranet_joint_small_backbone"):
        c = yaml.safe_load(open(f"{REF}/cfgs/{name}.yml"))["backbone"]
        try:
            RegNetY(c["initial_width"], c["slope"], c["quantized_param"], c["network_depth"], c["bottleneck_ratio"],
                    c["group_width"], c["stride"], c["se_ratio"])
        except RuntimeError:
            pass
        nb, bw, gw = captured["v"]
        res[f"regnet/{name}/depths"], res[f"regnet/{name}/widths"], res[f"regnet/{name}/gw"] = \
            np.array(nb), np.array(bw), np.array(gw)
        res[f"regnet/{name}/args"] = np.array([c["initial_width"], c["slope"], c["quantized_param"], c["network_depth"],
                                               c["bottleneck_ratio"], c["group_width"]], dtype=np.float64)
    anynet.AnyNetX.__init__ = orig
    np.savez_compressed(os.path.join(HERE, "loss_kats.npz"), **res)
    print("loss KATs: %d arrays" % len(res))


def _digest(t):
    t = t.detach().double()
    return np.array([float(t.mean()), float(t.abs().max()), float(t.norm())], dtype=np.float64)


def big_digest(cond=False):
    """cond: the same digests on the WELL-CONDITIONED state (tests/helpers.conditioned_state: residual-branch BatchNorm scale x0.1) ->
    big_cond.npz: the state on which the HIP path is held to ABSOLUTE tolerances against fp32 at full size (tests/test_fullsize3_gpu.py).
    Big cfg (the reference's own hydranet_joint_big_backbone.yml, repo-default 640x640), B=1: state_dict key/shape list and NUMERIC
    digests (mean, abs-max, L2) of every feature map / head output / loss / parameter gradient of a training-mode step and of the
    eval-mode forward.  The 171 MB of weights cannot be committed, so they come from a seeded recipe
    (tests/helpers.synthetic_state) that the CPU test re-runs to feed the oracle the same values."""
    from model import HydraNet
    from head_lane.lanedetect_loss import cal_loss_regress  # noqa: F401  (the default points_per_line=160 is live at H=640)
    from tests.helpers import conditioned_state, synthetic_state
    cfgs = yaml.safe_load(open(f"{REF}/cfgs/hydranet_joint_big_backbone.yml"))
    ours = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
    for sec in ("backbone", "detection", "segment", "lane"):         # cfgs/hydranet_big.yml is the same model (paths blanked)
        assert cfgs[sec] == ours[sec], sec
    h = w = 640
    assert cfgs["dataloader"]["network_input_height"] == h and cfgs["dataloader"]["network_input_width"] == w
    torch.manual_seed(0)
    net = HydraNet(cfgs)
    keys = list(net.state_dict().keys())
    shapes = [tuple(v.shape) for v in net.state_dict().values()]
    res = {"keys": np.array(keys), "shapes": np.array([",".join(map(str, s)) for s in shapes]),
           "n_params": np.array(sum(p.numel() for p in net.parameters())),
           "param_keys": np.array([k for k, _ in net.named_parameters()])}
    sd = (conditioned_state if cond else synthetic_state)(keys, shapes, seed=11)
    net.load_state_dict(sd)
    res["digest/state_sha256"] = np.array(hashlib.sha256(b"".join(np.ascontiguousarray(_np(sd[k])).tobytes() for k in keys)).hexdigest())
    batch = O.synthetic_batch(cfgs, 1, h, w, seed=1)
    net.train()
    x = batch["image"]
    out = net(x)
    ld = net.cal_loss(out, batch)
    s, d, l = cfgs["segment"], cfgs["detection"], cfgs["lane"]
    total = ld["loss_seg"] * s["segment_weight"] \
        + (ld["loss_det_cls"] * d["loss_cls_weight"] + ld["loss_det_reg"] * d["loss_reg_weight"]) * d["detection_weight"] \
        + (ld["loss_lane_cls_pos"] + ld["loss_lane_cls_neg"] + ld["loss_lane_loc"]) * l["lane_weight"]
    total.backward()
    for k, v in ld.items():
        res["digest/loss/" + k] = _np(v).astype(np.float64)
    res["digest/loss/total"] = _np(total).astype(np.float64)
    res["digest/train/seg"] = _digest(out["seg"])
    res["digest/train/regression"] = _digest(out["detection"]["regression"])
    res["digest/train/classification"] = _digest(out["detection"]["classification"])
    res["digest/train/lane_cls"] = _digest(out["lane"]["predict_cls"])
    res["digest/train/lane_loc"] = _digest(out["lane"]["predict_loc"])
    gk, gv = [], []
    for k, p in net.named_parameters():
        if p.grad is not None:
            gk.append(k)
            gv.append(float(p.grad.double().norm()))
    res["digest/grad_keys"], res["digest/grad_l2"] = np.array(gk), np.array(gv, dtype=np.float64)
    # eval-mode forward with the post-step running statistics: feature maps, fused maps, head outputs
    net.eval()
    with torch.no_grad():
        feats = net.backbone(x)
        fused = net.neck(feats)
        dep = net(x, "deploy")
    for i, f in enumerate(feats):
        res[f"digest/eval/feat{i}"] = _digest(f)
    for i, f in enumerate(fused):
        res[f"digest/eval/fused{i}"] = _digest(f)
    res["digest/eval/regression"], res["digest/eval/classification"] = _digest(dep[2]), _digest(dep[3])
    res["digest/eval/lane_cls"], res["digest/eval/lane_loc"] = _digest(dep[4]), _digest(dep[5])
    cnt = torch.bincount(dep[0].flatten(), minlength=5)
    res["digest/eval/seg_argmax_hist"] = _np(cnt).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "big_cond.npz" if cond else "big_keys.npz"), **res)
    print("big cfg%s: %d state_dict entries, %d params; losses" % (" (conditioned state)" if cond else "", len(keys), int(res["n_params"])),
          {k: float(v) for k, v in ld.items()}, "argmax hist", cnt.tolist())


def lane_fixture():
    """Lane decode + lane NMS through the reference's own codec (LaneCodec.decode_lane, nms_with_pos via LaneHeader.decode): synthetic
    head outputs shaped like trained ones (a few smooth lanes seen by several neighbouring anchors + noise anchors), at 512x1024 and 640x640."""
    from head_lane.lane_codec import LaneCodec
    from head_lane.lanedetect import LaneHeader
    res = {}
    for tag, (h, w), seed in (("512x1024", (512, 1024), 21), ("640x640", (640, 640), 22), ("128x256", (128, 256), 23)):
        stride, interval = 32, 8
        ppl = h // interval
        coder = LaneCodec(input_width=w, input_height=h, anchor_stride=stride, points_per_line=ppl, do_interpolate=True, anchor_lane_num=1,
                          scale_invariance=True)
        fh, fw = h // stride, w // stride
        g = torch.Generator().manual_seed(seed)
        hw = fh * fw
        cls = torch.randn(hw, 2, generator=g)
        cls[:, 0] += 2.0                                            # mostly background
        loc = torch.randn(hw, 2 * ppl + 2, generator=g) * 0.3
        loc[:, ppl] = torch.rand(hw, generator=g) * ppl             # down length
        loc[:, ppl + 1] = torch.rand(hw, generator=g) * ppl         # up length
        # a few ground-truth-like lanes: x(pos) = x0 + slope * pos (in pixels), every anchor within 1.5 cells of the lane fires
        for li in range(5):
            x0 = float(torch.rand(1, generator=g)) * w
            slope = (float(torch.rand(1, generator=g)) - 0.5) * 6.0
            for ah in range(fh):
                pos0 = int((fh - 1 - ah) * (ppl / fh))
                xl = x0 + slope * pos0
                for aw in range(fw):
                    cx = (aw + 0.5) * stride
                    if abs(cx - xl) < 1.5 * stride and 0 <= xl < w:
                        idx = ah * fw + aw
                        cls[idx, 1] = cls[idx, 0] + 2.0 + 2.0 * float(torch.rand(1, generator=g))
                        up = torch.arange(ppl).float()
                        loc[idx, ppl + 2:] = ((x0 + slope * (pos0 + up)) - cx) / interval + torch.randn(ppl, generator=g) * 0.05
                        loc[idx, :ppl] = ((x0 + slope * (pos0 - 1 - up)) - cx) / interval + torch.randn(ppl, generator=g) * 0.05
                        loc[idx, ppl] = pos0 + 0.5
                        loc[idx, ppl + 1] = ppl - pos0 + 0.5
        res[f"{tag}/cls"], res[f"{tag}/loc"] = _np(cls), _np(loc)
        res[f"{tag}/geom"] = np.array([w, h, stride, ppl])
        for name, (thr, nms_thr, use_mean) in (("a", (0.5, 100, False)), ("b", (0.9, 40, False)), ("c", (0.3, 60, True))):
            lanes = LaneHeader.decode(cls, loc, coder, thr, nms_thr, use_mean)
            cand = coder.decode_lane(torch.softmax(cls, -1), loc, thr)
            k = f"{tag}/{name}"
            res[k + "/params"] = np.array([thr, nms_thr, float(use_mean)])
            res[k + "/n_candidates"] = np.array(len(cand))
            res[k + "/prob"] = np.array([float(l.prob) for l in lanes], np.float64)
            res[k + "/start_pos"] = np.array([l.start_pos for l in lanes], np.int64)
            res[k + "/end_pos"] = np.array([l.end_pos for l in lanes], np.int64)
            res[k + "/ax"] = np.array([float(l.ax) for l in lanes], np.float64)
            res[k + "/ay"] = np.array([float(l.ay) for l in lanes], np.float64)
            res[k + "/npts"] = np.array([len(l.lane) for l in lanes], np.int64)
            res[k + "/xs"] = np.array([float(p.x) for l in lanes for p in l.lane], np.float64)
            res[k + "/ys"] = np.array([float(p.y) for l in lanes for p in l.lane], np.float64)
            print("lane fixture", k, "candidates", len(cand), "kept", len(lanes))
    np.savez_compressed(os.path.join(HERE, "lane_decode.npz"), **res)


def aux_fixture():
    """(f3) ImageNet normalisation through the reference's own imagenet_normalize (dataset/utility.py:213-227; the resize is cv2 = absent
    third party, so only frames at the network size are recorded) and (f4) IoU statistics through head_seg/seg_metrics.py."""
    from dataset.utility import imagenet_normalize
    from head_seg.seg_metrics import IntersectionOverUnion, stat_scores_multiple_classes
    res = {}
    rng = np.random.RandomState(5)
    frame = rng.randint(0, 256, size=(24, 40, 3)).astype(np.uint8)               # BGR, already at the "network size"
    img = frame[:, :, ::-1].astype(np.float32)                                    # cv2.cvtColor(BGR2RGB) is a channel flip
    img = imagenet_normalize(img=img)
    res["pre/frame_bgr"] = frame
    res["pre/expected"] = np.transpose(img, (2, 0, 1)).astype(np.float32)          # .transpose + torch.tensor(...).float()
    g = torch.Generator().manual_seed(9)
    pred = torch.randint(0, 5, (3, 32, 48), generator=g)
    tgt = torch.randint(0, 5, (3, 32, 48), generator=g)
    tgt[0, :4] = 255                                                               # ignore pixels
    tgt[:, :, :6] = pred[:, :, :6]                                                 # some guaranteed hits
    tp, fp, tn, fn, sup = stat_scores_multiple_classes(pred.clone(), tgt.clone(), 5)
    res["iou/pred"], res["iou/target"] = _np(pred), _np(tgt)
    res["iou/tp"], res["iou/fp"], res["iou/fn"], res["iou/sup"] = _np(tp), _np(fp), _np(fn), _np(sup)
    for name, kw in (("plain", dict()), ("ignore0", dict(ignore_index=0)), ("absent", dict(absent_score=1.0))):
        m = IntersectionOverUnion(n_classes=5 if name != "absent" else 7, **kw)
        m.update(pred.clone(), tgt.clone())
        m.update(tgt.clone().clamp_max(4), tgt.clone())                            # a second batch (streaming)
        res[f"iou/{name}/scores"] = _np(m.compute())
    np.savez_compressed(os.path.join(HERE, "aux_stages.npz"), **res)
    print("aux fixture:", {k: v.shape for k, v in res.items()})


def schedule_fixture():
    """The head-wise fine-tuning schedule (train.py:441-515) by running the reference's own `main` -- extracted from its source with ast
    and executed against a recording stand-in for HydraTrainer (train.py itself cannot be imported: dataset / cv2 / pycocotools) -- for a
    few (epoch, epoch_tuning, tuning_turn) settings: which module's parameters() the optimizer's first param group holds in every epoch."""
    import ast
    import json
    src = open(f"{REF}/train.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    code = compile(ast.Module(body=[fn], type_ignores=[]), "reference_train_main", "exec")

    class _Mod:
        def __init__(self, tag):
            self.tag = tag

        def parameters(self):
            return [self.tag]

    class _Net(_Mod):
        def __init__(self):
            super().__init__("joint")
            self.laneheader, self.detectheader, self.segheader = _Mod("lane"), _Mod("det"), _Mod("seg")

    out = {}
    for epoch_all, epoch_tuning, turns, fine in ((16, 1, 2, True), (30, 1, 1, True), (24, 2, 2, True), (12, 1, 4, True), (5, 1, 1, False)):
        log = []

        class _Trainer:
            def __init__(self, cfgs, cfg_path):
                self.use_distribute = False
                self.hydranet = _Net()
                self.optimizer = types.SimpleNamespace(param_groups=[{"params": ["initial"]}])

            def train_one_epoch(self, epoch):
                log.append(self.optimizer.param_groups[0]["params"][0])

            def valid(self, epoch):
                pass

        cfg = {"train": {"epoch": epoch_all, "fine_tuning": fine, "epoch_tuning": epoch_tuning, "tuning_turn": turns}}
        ns = {"yaml": types.SimpleNamespace(safe_load=lambda f: cfg), "open": lambda p: p, "HydraTrainer": _Trainer, "print": lambda *a, **k: None}
        exec(code, ns)
        ns["main"]("cfg")
        out["%d,%d,%d,%d" % (epoch_all, epoch_tuning, turns, int(fine))] = log
        print("schedule", epoch_all, epoch_tuning, turns, fine, log)
    json.dump(out, open(os.path.join(HERE, "tuning_schedule.json"), "w"), indent=1)


def lane_metric_fixture():
    """(f4) lane F1 through the reference's own head_lane/lane_metric.py (spline_interp, calc_iou, evaluate_core, LaneMetric) on synthetic
    ground-truth / prediction sets; cv2.line is the stand-in above.  -> lane_metric.json"""
    from head_lane import lane_metric as LM
    rs = np.random.RandomState(31)
    H, W = 360, 640

    def lane(x0, slope, curve, n, y0=350.0, dy=-40.0, jitter=0.0):
        return [{"x": float(x0 + slope * i + curve * i * i + jitter * rs.randn()), "y": float(y0 + dy * i)} for i in range(n)]
    splines = [lane(100, 10, 0.3, 8), lane(300, -5, 0.0, 2), lane(500, 2, -1.0, 3), lane(50, 30, 0.5, 9, jitter=2.0)]
    out = {"H": H, "W": W, "splines": []}
    for ln in splines:
        ip = LM.spline_interp(lane=ln, step_t=1)
        out["splines"].append({"lane": ln, "x": [float(p["x"]) for p in ip], "y": [float(p["y"]) for p in ip]})
    images = []
    for k in range(4):
        gts = [lane(80 + 150 * j + 10 * k, 6 - 3 * j, 0.2 * (j - 1), 8) for j in range(3 + (k % 2))]
        prs = []
        for j, g in enumerate(gts):
            if (j + k) % 4 == 3:
                continue                                           # a missed lane
            off = (2.0, 7.0, 25.0, 4.0)[(j + k) % 4]               # IoU well above / above / far below the 0.5 threshold
            prs.append({"score": float(0.3 + 0.2 * ((j + 2 * k) % 4)), "points": [{"x": p["x"] + off, "y": p["y"]} for p in g]})
        if k == 2:
            prs.append({"score": 0.9, "points": lane(600, -20, 0.0, 6)})     # a false positive
        shape = {"width": W, "height": H}
        images.append(dict(pr_result={"Lines": prs, "Shape": shape}, gt_result={"Lines": gts, "Labels": [1] * len(gts), "Shape": shape}))
    out["images"] = images
    res = {}
    for lw in (30, 10):
        for thr in (0.5, 0.3):
            m = LM.LaneMetric(method="f1_measure", iou_thresh=0.5, lane_width=lw, thresh_list=[thr])
            [h.reset() for h in m.metric_handlers]
            m(output=images)
            h = m.metric_handlers[0]
            res["%d,%g" % (lw, thr)] = {"records": h.result_record, "summary": h.summary(), "f1": m.summary()}
            print("lane metric", lw, thr, h.summary())
    out["results"] = res
    # one IoU matrix in full (lane width 30, all predictions of image 0)
    g0 = images[0]["gt_result"]["Lines"]
    p0 = [l["points"] for l in images[0]["pr_result"]["Lines"]]
    out["iou_image0"] = [[float(LM.calc_iou(g, p, dict(eval_height=H, eval_width=W, lane_width=30))) for p in p0] for g in g0]
    json.dump(out, open(os.path.join(HERE, "lane_metric.json"), "w"))


if __name__ == "__main__":
    _install_shims()
    which = sys.argv[1:] or ["tiny", "tiny4", "kats", "big", "bigcond", "lane", "aux", "schedule", "lanemetric"]
    if "lanemetric" in which:
        lane_metric_fixture()
    if "tiny" in which:
        tiny_fixture()
    if "tiny4" in which:
        tiny_fixture("hydranet_tiny4.yml", "tiny4_hydranet.npz")
    if "schedule" in which:
        schedule_fixture()
    if "kats" in which:
        loss_kats()
    if "big" in which:
        big_digest()
    if "bigcond" in which:
        big_digest(cond=True)
    if "lane" in which:
        lane_fixture()
    if "aux" in which:
        aux_fixture()
