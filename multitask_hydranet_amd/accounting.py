"""Algorithmic work of HydraNet per image, per segment: conv MACs, conv input elements (sum X)
and conv output elements (sum Y) derived from the configuration alone -- the figures the reference-side forward hooks on nn.Conv2d count
(tests/test_host_cpu.py pins the 512x1024 and 640x640 totals to BASELINE.md).  bench.py prices its per-segment roofline floors with them:
seg decoder on the dense bf16 MFMA peak (~395 FLOP/B), every other segment on HBM bytes (<= ~90 FLOP/B): 3 * (X + Y) * 2 B per image
for forward + backward (read X write Y; read dY write dX; read X, dY)."""
from __future__ import annotations

from typing import Dict

from .model import regnet_stages

PEAK_BF16_FLOPS = 2.5e15        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_BYTES = 8.0e12


def conv_work(cfgs: dict, h: int, w: int) -> Dict[str, Dict[str, float]]:
    """-> {segment: {"macs", "x", "y"}} per image for backbone / neck / seg / det / lane"""
    b = cfgs["backbone"]
    widths, depths, _ = regnet_stages(b["initial_width"], b["slope"], b["quantized_param"], b["network_depth"], b["bottleneck_ratio"], b["group_width"])
    out = {k: dict(macs=0.0, x=0.0, y=0.0) for k in ("backbone", "neck", "seg", "det", "lane")}

    def conv(seg, cin, cout, k, hin, win, hout, wout, groups=1, padded=False):
        """padded: the reference pads OUTSIDE the nn.Conv2d (ReflectionPad2d(1) in the seg decoder, head_seg/segmentation.py:36-47; F.pad in
        Conv2dStaticSamePadding, net/common.py:57-72), so the conv's input -- what the survey's forward hooks counted -- is (h+2) x (w+2)"""
        out[seg]["macs"] += cout * (cin // groups) * k * k * hout * wout
        out[seg]["x"] += cin * (hin + 2 * padded) * (win + 2 * padded)
        out[seg]["y"] += cout * hout * wout

    conv("backbone", 3, 32, 3, h, w, h // 2, w // 2)
    prev, hi, wi = 32, h // 2, w // 2
    feats = []
    for wd, d in zip(widths, depths):
        for i in range(d):
            cin, s = (prev, b["stride"]) if i == 0 else (wd, 1)
            ho, wo = hi // s, wi // s
            conv("backbone", cin, wd, 1, hi, wi, hi, wi)
            conv("backbone", wd, wd, 3, hi, wi, ho, wo, groups=wd // 8)
            if b["se_ratio"] is not None:
                se = cin // b["se_ratio"]
                conv("backbone", wd, se, 1, 1, 1, 1, 1)
                conv("backbone", se, wd, 1, 1, 1, 1, 1)
            conv("backbone", wd, wd, 1, ho, wo, ho, wo)
            if s != 1 or cin != wd:
                conv("backbone", cin, wd, 1, hi, wi, ho, wo)
            hi, wi = ho, wo
        prev = wd
        feats.append((wd, hi, wi))
    f = b["fpn_num_filters"]
    lv = [(h >> (3 + l), w >> (3 + l)) for l in range(5)]          # P3 .. P7

    def sep(seg, cin, cout, hh, ww):
        conv(seg, cin, cin, 3, hh, ww, hh, ww, groups=cin, padded=True)
        conv(seg, cin, cout, 1, hh, ww, hh, ww)
    for cell in range(b["fpn_cell_repeats"]):
        for l in (3, 2, 1, 0, 1, 2, 3, 4):                         # conv6_up .. conv3_up, conv4_down .. conv7_down
            sep("neck", f, f, *lv[l])
        if cell == 0:
            if len(feats) == 5:
                (c3, *_), (c4, *_), (c5, *_), (c6, *_) = feats[1:]
                conv("neck", c6, f, 1, *lv[3], *lv[3])
            else:
                (c3, *_), (c4, *_), (c5, *_) = feats[1:]
                conv("neck", c5, f, 1, *lv[2], *lv[2])            # p5_to_p6 (then pooled)
            conv("neck", c3, f, 1, *lv[0], *lv[0])
            for _ in range(2):
                conv("neck", c4, f, 1, *lv[1], *lv[1])
                conv("neck", c5, f, 1, *lv[2], *lv[2])
    if cfgs["train"]["train_seg"]:
        sc = cfgs["segment"]
        enc, dec = sc["channel_dimension_seg_encode"], sc["channel_dimension_seg_decode"]
        n = len(enc)
        res = [(h >> 2, w >> 2)] + lv[:n - 1]                      # resolutions of the decoder inputs [feat0, P3, P4, P5]
        for i in range(n - 1, -1, -1):
            cin = enc[-1] if i == n - 1 else dec[i + 1]
            conv("seg", cin, dec[i], 3, *res[i], *res[i], padded=True)
            up = (2 * res[i][0], 2 * res[i][1])
            conv("seg", dec[i] + (enc[i - 1] if i > 0 else 0), dec[i], 3, *up, *up, padded=True)
        conv("seg", dec[0], len(sc["class_list"]), 3, h, w, h, w, padded=True)
    if cfgs["train"]["train_detect"]:
        d = cfgs["detection"]
        fd = d["fpn_num_filters_detect"]
        for cout in (9 * 4, 9 * d["num_classes"]):
            for l in range(d["pyramid_levels"]):
                for _ in range(d["box_class_repeats"]):
                    sep("det", fd, fd, *lv[l])
                sep("det", fd, cout, *lv[l])
    if cfgs["train"]["train_lane"]:
        l = cfgs["lane"]
        c = l["base_channel"]
        hh, ww = h // l["anchor_stride"], w // l["anchor_stride"]
        for cout in (l["num_classes"], h // l["interval"] + 1, h // l["interval"] + 1):
            conv("lane", c, c, 1, hh, ww, hh, ww)
            conv("lane", c, cout, 1, hh, ww, hh, ww)
    return out


def segment_floors_ms(cfgs: dict, h: int, w: int, n: int, elem_bytes: int = 2) -> Dict[str, float]:
    """roofline floor of forward + backward per STEP of n images, per segment (ms): seg decoder = 3 x its conv FLOPs on the MFMA peak, every
    other segment = 3 * (X + Y) * elem_bytes on the HBM peak (SURVEY 8(d) 'segment-wise roofline'); 'losses' carry no conv work (floor 0)"""
    wk = conv_work(cfgs, h, w)
    fl = {}
    for seg, v in wk.items():
        t_mfma = 3 * 2 * v["macs"] / PEAK_BF16_FLOPS
        t_hbm = 3 * (v["x"] + v["y"]) * elem_bytes / PEAK_HBM_BYTES
        fl[seg] = (t_mfma if seg == "seg" else t_hbm) * n * 1e3
    fl["losses"] = 0.0
    return fl
