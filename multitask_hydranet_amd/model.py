"""HydraNet nn.Module surface (model/model.py:26-264 of the reference) over the MI355X HIP kernels.

Same constructor (``HydraNet(cfgs: dict, onnx_export=False)``), same ``forward(x, mode)`` / ``cal_loss(pred, gt)`` contract, same
sub-module attributes (backbone / neck / segheader / detectheader / laneheader) and the same 1177 ``state_dict`` keys as the
reference, so checkpoints and train.py / demo.py style callers work unchanged.  Underneath, parameters live in a generic tree of
containers built from a declarative spec, activations are NHWC bf16, and every tensor op on the path is a kernel of
libhydranet_hip.so reached through multitask_hydranet_amd.ops.  There is no eager fallback: constructing the module on a machine
without the shared library raises.
"""
from __future__ import annotations

import contextlib
import itertools
import math
import sys
import weakref
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops as K
from .ops import ACT_ELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH

BN_STD = dict(eps=1e-5, momentum=0.1)        # nn.BatchNorm2d defaults: backbone + lane head
BN_FPN = dict(eps=1e-3, momentum=0.01)       # neck + detection towers (net/common.py:98)


def regnet_stages(initial_width, slope, quantized_param, network_depth, bottleneck_ratio, group_width):
    """Stage widths / depths / group widths of the RegNetY backbone (net/regnet.py:21-38)."""
    u = initial_width + slope * np.arange(network_depth)
    s = np.round(np.log(u / initial_width) / np.log(quantized_param))
    q = 8 * np.round(initial_width * np.power(quantized_param, s) / 8)
    widths, depths = np.unique(q.astype(np.int32), return_counts=True)
    gws = np.array([min(group_width, int(v) // bottleneck_ratio) for v in widths]).astype(np.int32) * bottleneck_ratio
    widths = (np.round(widths // bottleneck_ratio / group_width) * group_width).astype(np.int32)
    return widths.tolist(), depths.tolist(), gws.tolist()


class _Node(nn.Module):
    """Plain container; the five top-level ones get a callable bound to the owning HydraNet."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_fwd", None)

    def forward(self, *a, **k):
        if self._fwd is None:
            raise RuntimeError("this container is only a parameter holder")
        return self._fwd(*a, **k)


class _Spec:
    """Declares parameters/buffers by dotted name and materialises the container tree."""

    def __init__(self, root: nn.Module):
        self.root = root

    def _walk(self, name):
        parts = name.split(".")
        m = self.root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Node())
            m = m._modules[p]
        return m, parts[-1]

    def param(self, name, tensor):
        m, leaf = self._walk(name)
        m.register_parameter(leaf, nn.Parameter(tensor))

    def buffer(self, name, tensor):
        m, leaf = self._walk(name)
        m.register_buffer(leaf, tensor)

    # --- initialisers ------------------------------------------------------------------------------------
    def conv(self, name, cout, cin_g, k, bias, backbone_init=False):
        w = torch.empty(cout, cin_g, k, k)
        if backbone_init:                                  # net/anynet.py:124-128
            w.normal_(0.0, math.sqrt(2.0 / (k * k * cout)))
        else:                                              # nn.Conv2d default (kaiming_uniform(a=sqrt(5)))
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.param(name + ".weight", w)
        if bias:
            bound = 1.0 / math.sqrt(cin_g * k * k)
            self.param(name + ".bias", torch.empty(cout).uniform_(-bound, bound))

    def bn(self, name, c):
        self.param(name + ".weight", torch.ones(c))
        self.param(name + ".bias", torch.zeros(c))
        self.buffer(name + ".running_mean", torch.zeros(c))
        self.buffer(name + ".running_var", torch.ones(c))
        self.buffer(name + ".num_batches_tracked", torch.tensor(0, dtype=torch.long))


class HydraNet(nn.Module):
    def __init__(self, cfgs: dict, onnx_export: bool = False):
        super().__init__()
        K.lib()                                            # fail loudly if libhydranet_hip.so is absent
        self.cfgs = cfgs
        self.onnx_export = onnx_export
        self.net_input_width = cfgs["dataloader"]["network_input_width"]
        self.net_input_height = cfgs["dataloader"]["network_input_height"]
        b = cfgs["backbone"]
        self.widths, self.depths, self.group_widths = regnet_stages(b["initial_width"], b["slope"], b["quantized_param"],
                                                                    b["network_depth"], b["bottleneck_ratio"], b["group_width"])
        assert b["bottleneck_ratio"] == 1 and all(g == 8 for g in self.group_widths), "kernels are built for group width 8"
        self.backbone_stride = b["stride"]
        self.se_ratio = b["se_ratio"]
        self.fpn_num_filters = b["fpn_num_filters"]
        self.fpn_cell_repeats = b["fpn_cell_repeats"]
        self.conv_channel_coef = list(b["conv_channel_coef"])
        t = cfgs["train"]
        self.train_detect, self.train_seg, self.train_lane = t["train_detect"], t["train_seg"], t["train_lane"]
        self.check_finite = True                           # the reference exit()s on a zero / non-finite loss (model.py:212-258)
        self.lane_points_per_line = 160                    # cal_loss_regress default that model.py:246 never overrides
        self._anchor_cache = {}
        self._pending_nbt = []
        self.heads_on_side_stream = False          # measured: no gain over the single-stream graph on MI355X (kept for experiments)
        self._side_streams = {}
        self.seg_phase_output = True
        self.seg_fuse_elu_bwd = True       # ELU' of a decoder block applied by the next block's gradient fold
        self._pack_plan = None
        self._folded = None                 # prepare_inference(): conv name -> (packed folded weights, fp32 bias)
        self.deploy_postprocess = None      # (conf_thres, iou_thres): forward(x, "deploy") then appends the device-side detections
        self.pack_det_levels = True         # level-packed det towers when every level has a multiple of 128 rows
        self.levels_on_streams = False      # hipGraph branches cost more than they hide on gfx950 (55 vs 43 ms)
        self.grad_scope = None              # "lane" | "det" | "seg": head-only fine-tuning phase (see _forward)

        spec = _Spec(self)
        self._declare_backbone(spec)
        self._declare_neck(spec)
        if self.train_detect:
            self._declare_det(spec)
            self.loss_detect = K.det_loss_hip            # HIP only (losses.det_loss is the torch form the tests use as a reference)
        else:
            self.detectheader, self.loss_detect = None, None
        if self.train_seg:
            self._declare_seg(spec)
            s = cfgs["segment"]
            self.use_lovasz = s["use_lovasz"]
            assert not self.use_lovasz, "Lovasz loss is off in every shipped cfg and outside the hot path"
            # device-resident copy of the class weights (non-persistent: not part of the reference's state_dict)
            self.register_buffer("_seg_class_weight", torch.tensor(s["class_weight"], dtype=torch.float32), persistent=False)
            self._seg_cfg = (s["use_top_k"], s["top_k_ratio"], s["use_focal"])
            self.loss_seg = self._seg_loss
        else:
            self.segheader, self.loss_seg = None, None
        if self.train_lane:
            self._declare_lane(spec)
            self.loss_cls, self.loss_reg = K.lane_cls_loss_hip, K.lane_loc_loss_hip      # HIP; losses.py keeps the torch forms as test references
        else:
            self.laneheader, self.loss_cls, self.loss_reg = None, None, None
        self._bind_callables()
        self._idx: Dict[str, torch.Tensor] = {}
        self._reindex()

    # ------------------------------------------------------------------------------------------------------
    # parameter declaration (names, shapes and order = the reference's state_dict)
    # ------------------------------------------------------------------------------------------------------
    def _declare_backbone(self, s: _Spec):
        p = "backbone.net."
        s.conv(p + "stem.conv", 32, 3, 3, False, True)
        s.bn(p + "stem.bn", 32)
        prev = 32
        for k, (w, d) in enumerate(zip(self.widths, self.depths)):
            for i in range(d):
                cin = prev if i == 0 else w
                stride = self.backbone_stride if i == 0 else 1
                q = f"{p}stage_{k}.blocks.block_{i}."
                s.conv(q + "conv_block_1.0", w, cin, 1, False, True)
                s.bn(q + "conv_block_1.1", w)
                s.conv(q + "conv_block_2.0", w, 8, 3, False, True)
                s.bn(q + "conv_block_2.1", w)
                if self.se_ratio is not None:
                    se = cin // self.se_ratio
                    s.conv(q + "se.1", se, w, 1, True, True)
                    s.conv(q + "se.3", w, se, 1, True, True)
                s.conv(q + "conv_block_3.0", w, w, 1, False, True)
                s.bn(q + "conv_block_3.1", w)
                if stride != 1 or cin != w:
                    s.conv(q + "shortcut.0", w, cin, 1, False, True)
                    s.bn(q + "shortcut.1", w)
            prev = w

    def _sep(self, s: _Spec, name, cin, cout, norm):
        s.conv(name + ".depthwise_conv.conv", cin, 1, 3, False)
        s.conv(name + ".pointwise_conv.conv", cout, cin, 1, True)
        if norm:
            s.bn(name + ".bn", cout)

    def _declare_neck(self, s: _Spec):
        f = self.fpn_num_filters
        cc = self.conv_channel_coef
        for k in range(self.fpn_cell_repeats):
            p = f"neck.bifpn.{k}."
            for nm, n in (("p6_w1", 2), ("p5_w1", 2), ("p4_w1", 2), ("p3_w1", 2), ("p4_w2", 3), ("p5_w2", 3), ("p6_w2", 3), ("p7_w2", 2)):
                s.param(p + nm, torch.ones(n))
            for nm in ("conv6_up", "conv5_up", "conv4_up", "conv3_up", "conv4_down", "conv5_down", "conv6_down", "conv7_down"):
                self._sep(s, p + nm, f, f, True)
            if k == 0:
                def red(nm, cin):
                    s.conv(p + nm + ".0.conv", f, cin, 1, True)
                    s.bn(p + nm + ".1", f)
                red("p5_down_channel", cc[2])
                red("p4_down_channel", cc[1])
                red("p3_down_channel", cc[0])
                red("p5_to_p6", cc[2])
                if len(cc) == 4:
                    red("p6_down_channel", cc[3])
                red("p4_down_channel_2", cc[1])
                red("p5_down_channel_2", cc[2])

    def _declare_det(self, s: _Spec):
        d = self.cfgs["detection"]
        f, layers, levels = d["fpn_num_filters_detect"], d["box_class_repeats"], d["pyramid_levels"]
        self.num_anchors = 9
        for tower, cout in (("regressor", self.num_anchors * 4), ("classifier", self.num_anchors * d["num_classes"])):
            p = f"detectheader.{tower}."
            for i in range(layers):
                self._sep(s, f"{p}conv_list.{i}", f, f, False)
            for lv in range(levels):
                for i in range(layers):
                    s.bn(f"{p}bn_list.{lv}.{i}", f)
            self._sep(s, p + "header", f, cout, False)

    def _declare_seg(self, s: _Spec):
        sc = self.cfgs["segment"]
        enc, dec = sc["channel_dimension_seg_encode"], sc["channel_dimension_seg_decode"]
        idx = 0
        for i in range(len(enc) - 1, -1, -1):
            cin = enc[-1] if i == len(enc) - 1 else dec[i + 1]
            s.conv(f"segheader.decoder.{idx}.conv.conv", dec[i], cin, 3, True)
            cin = dec[i] + (enc[i - 1] if i > 0 else 0)
            s.conv(f"segheader.decoder.{idx + 1}.conv.conv", dec[i], cin, 3, True)
            idx += 2
        s.conv(f"segheader.decoder.{idx}.conv", len(sc["class_list"]), dec[0], 3, True)
        self._seg_layers = idx

    def _declare_lane(self, s: _Spec):
        l = self.cfgs["lane"]
        c = l["base_channel"]
        ppl = int(self.net_input_height / l["interval"])
        for nm, cout in (("conv_cls_conv", l["num_classes"]), ("conv_up_conv", ppl + 1), ("conv_down_conv", ppl + 1)):
            p = f"laneheader.{nm}."
            s.conv(p + "0", c, c, 1, False)
            s.bn(p + "1", c)
            s.conv(p + "3", cout, c, 1, True)

    # ------------------------------------------------------------------------------------------------------
    def _bind_callables(self):
        me = weakref.ref(self)
        def flushed(v):
            me()._flush_nbt()
            return v
        object.__setattr__(self.backbone, "_fwd", lambda x: flushed([K_to_nchw(t) for t in me()._backbone(x)]))
        object.__setattr__(self.neck, "_fwd", lambda feats: flushed(tuple(K_to_nchw(t) for t in me()._neck([K_to_nhwc(t) for t in feats]))))
        if self.train_seg:
            object.__setattr__(self.segheader, "_fwd", lambda feats: me()._seg([K_to_nhwc(t) for t in feats]))
            from .visual import seg_decode             # SegmentHeader.decode (head_seg/segmentation.py:107-125) on the device
            self.segheader.decode = seg_decode
        if self.train_detect:
            object.__setattr__(self.detectheader, "_fwd", lambda x, fused: flushed(me()._det(x, [K_to_nhwc(t) for t in fused])))
            self.detectheader.decode = _det_decode
            from .coco_json import invert_affine       # DetectionHeader.invert_affine (head_detect/detection.py:217-230; train.py:334-336)
            self.detectheader.invert_affine = invert_affine
            self.detectheader.display = _unavailable("detectheader.display (cv2 box / label drawing, head_detect/detection.py:247-252)")
        if self.train_lane:
            object.__setattr__(self.laneheader, "_fwd", lambda fused: flushed(me()._lane([K_to_nhwc(t) for t in fused])))
            from . import lane_codec as LC             # LaneHeader.decode / scale_to_org (head_lane/lanedetect.py:103-124) on the device
            self.laneheader.decode = LC.decode
            self.laneheader.decode_batch = LC.decode_batch
            self.laneheader.scale_to_org = LC.scale_to_org
            self.laneheader.visual = _unavailable("laneheader.visual (cv2 visualisation)")

    def _reindex(self):
        self._idx = {k: v for k, v in itertools.chain(self.named_parameters(), self.named_buffers()) if not k.startswith("_")}
        # raw-pointer writes to these tensors (this module's training forward, an optimizer that holds them) bump THIS module's epoch
        if getattr(self, "_mut_cell", None) is None:
            self._mut_cell = K.new_mutation_cell()
        K.tag_mutation_owner(self._idx.values(), self._mut_cell)
        # the parameters whose deferred gradients the flush node behind the backbone hands back (neck + det / lane heads): built once per
        # (re)index, not on every forward (~700 named_parameters() walks per step on the eager path)
        self._tail_params = [t for n_, t in self.named_parameters() if n_.startswith(("neck.", "detectheader.", "laneheader."))]

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._reindex()
        self._pack_plan = None
        self._folded = None
        K.clear_pack_cache()
        return r

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Accepts checkpoints written from a DistributedDataParallel wrapper as well: the reference saves `model.state_dict()` of the DDP
        module (train.py:437, keys "module.<name>") and strips the prefix when it loads them back (deparallel_model, train.py:96-109)."""
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = type(state_dict)((k[len("module."):], v) for k, v in state_dict.items())
        # new values: the BatchNorm-folded inference operands of prepare_inference() are stale.  In-place copies keep every parameter's
        # storage, so the per-step pack plan (a device table of raw pointers, re-validated by forward()) stays; assign=True swaps storage.
        self._folded = None
        K.clear_pack_cache()
        r = super().load_state_dict(state_dict, strict=strict, assign=assign)
        if assign:
            self._pack_plan = None
            self._reindex()
        return r

    # ------------------------------------------------------------------------------------------------------
    # inference with folded BatchNorm (BASELINE config 5)
    # ------------------------------------------------------------------------------------------------------
    def prepare_inference(self):
        """Fold every eval-mode BatchNorm that directly follows a convolution into that convolution's packed bf16 weights and an fp32 bias
        (backbone conv_block_1/2/3 and shortcuts, BiFPN channel reducers and separable blocks, lane branches).  Call after .eval() and after
        loading weights; forward() in eval mode then runs conv + BN + activation (+ the XBlock identity branch) as one launch per conv.
        .train(), load_state_dict() and device / dtype moves drop the folded operands (call again); an in-place parameter update in eval
        mode (an optimizer step without .train()) is detected through the parameters' version counters and raises in forward()."""
        assert not self.training, "prepare_inference() folds the RUNNING statistics: call .eval() first"
        P = self._idx
        folded = {}
        for name in P:
            if not name.endswith(".running_mean"):
                continue
            bn = name[:-len(".running_mean")]
            conv, kind, eps = None, "1x1", BN_STD["eps"]
            if bn.startswith("backbone.net.stage_"):
                conv = bn[:-1] + "0"                                  # conv_block_k.1 -> conv_block_k.0, shortcut.1 -> shortcut.0
                if ".conv_block_2." in bn:
                    blk0 = bn.split(".blocks.block_")[1].split(".")[0] == "0"
                    if blk0 and self.backbone_stride != 1:
                        continue                                      # stride-2 grouped conv: VALU stencil kernel, BN applied after it
                    kind = "g3x3"
            elif bn.startswith("neck.") and bn.endswith(".bn"):
                conv, eps = bn[:-3] + ".pointwise_conv.conv", BN_FPN["eps"]
            elif bn.startswith("neck.") and bn.endswith(".1"):
                conv, eps = bn[:-2] + ".0.conv", BN_FPN["eps"]
            elif bn.startswith("laneheader."):
                conv = bn[:-1] + "0"
            if conv is None or (conv + ".weight") not in P:
                continue
            w = P[conv + ".weight"]
            if not w.is_cuda:
                raise RuntimeError("prepare_inference() packs device operands: move the module to the GPU first")
            folded[conv] = K.fold_conv_bn(w, P.get(conv + ".bias"), P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"],
                                          P[bn + ".running_var"], eps, kind)
        self._folded = folded
        self._folded_versions = [(t, t._version) for n, t in P.items() if n.split(".")[-1] != "num_batches_tracked"]
        self._folded_epoch = K.mutation_epoch(self._mut_cell)
        return self

    def _check_folded(self):
        """the folded operands are constants of the parameters / running statistics as they were at prepare_inference(): any change since --
        visible to autograd (version counters) or made through raw pointers by this library (this module's mutation epoch: its own
        training forward, an optimizer step on its parameters) -- folds again (a few hundred small launches, once per such event), except
        inside a capture, where the operands must stay what the warm-up forwards used"""
        if self._folded is None:
            return
        if all(t._version == v for t, v in self._folded_versions) and self._folded_epoch == K.mutation_epoch(self._mut_cell):
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("parameters or BatchNorm statistics changed since prepare_inference(): call prepare_inference() again before "
                               "capturing the deploy forward")
        self.prepare_inference()

    def train(self, mode: bool = True):
        if mode:
            self._folded = None
        return super().train(mode)

    def _bn(self, name):
        P = self._idx
        if self.training:                                   # counters are bumped once per forward with ONE multi-tensor add
            self._pending_nbt.append(P[name + ".num_batches_tracked"])
        return (P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"], P[name + ".running_var"], None)

    def _flush_nbt(self):
        if self._pending_nbt:
            torch._foreach_add_(self._pending_nbt, 1)
            self._pending_nbt = []

    # ------------------------------------------------------------------------------------------------------
    # forward pieces (NHWC bf16 inside)
    # ------------------------------------------------------------------------------------------------------
    def _cba(self, x, conv, bn, bnkw, **kw):
        P = self._idx
        f = self._folded.get(conv) if (self._folded is not None and not self.training) else None
        if f is not None and x.dim() == 4 and x.dtype == torch.bfloat16:
            return K.conv_infer(x, f[0], f[1], P[conv + ".weight"].shape[0], kw.get("kind", "1x1"), kw.get("stride", 1), kw.get("act", ACT_NONE),
                                kw.get("res"), kw.get("gate"))
        if not x.requires_grad:
            kw.pop("slot", None)
        return K.conv_bn_act(x, P[conv + ".weight"], P.get(conv + ".bias"), self._bn(bn), training=self.training, **bnkw, **kw)

    def _xblock(self, q, x, stride, group=None):
        """XBlock.forward, net/anynet.py:65-76.  group: the stage's ops.WgradGroup (1x1 weight gradients deferred to the stage boundary)"""
        P = self._idx
        has_se, has_sc = (q + "se.1.weight") in P, (q + "shortcut.0.weight") in P
        if self._folded is not None and not self.training:            # inference: 7 launches per block, nothing but GEMM epilogues
            a = self._cba(x, q + "conv_block_1.0", q + "conv_block_1.1", BN_STD, act=ACT_RELU)
            b = self._cba(a, q + "conv_block_2.0", q + "conv_block_2.1", BN_STD, kind="g3x3", stride=stride, act=ACT_RELU)
            gate = None
            if has_se:      # the gate scales conv_block_3's weights per image where that form applies, else the activation (one launch)
                fold = K.gate_folds_into_weights(b)
                r = K.se_gate_infer(b, P[q + "se.1.weight"], P[q + "se.1.bias"], P[q + "se.3.weight"], P[q + "se.3.bias"], apply=not fold)
                b, gate = (b, r) if fold else (r, None)
            s = self._cba(x, q + "shortcut.0", q + "shortcut.1", BN_STD, stride=stride, act=ACT_NONE) if has_sc else x
            return self._cba(b, q + "conv_block_3.0", q + "conv_block_3.1", BN_STD, res=s, act=ACT_RELU, gate=gate)
        if K.xblock_fusable(x, P[q + "conv_block_1.0.weight"], stride, has_se, has_sc):     # one autograd node, 8 + 21 launches
            bn = [self._bn(q + f"conv_block_{i}.1")[:4] for i in (1, 2, 3)]
            sc = (P[q + "shortcut.0.weight"], *self._bn(q + "shortcut.1")[:4]) if has_sc else ()
            return K.XBlockFn.apply(x, P[q + "conv_block_1.0.weight"], *bn[0], P[q + "conv_block_2.0.weight"], *bn[1],
                                    P[q + "se.1.weight"], P[q + "se.1.bias"], P[q + "se.3.weight"], P[q + "se.3.bias"],
                                    P[q + "conv_block_3.0.weight"], *bn[2], BN_STD["eps"], BN_STD["momentum"], self.training, stride,
                                    *(sc if sc else (None,) * 5), group)
        a = self._cba(x, q + "conv_block_1.0", q + "conv_block_1.1", BN_STD, act=ACT_RELU)
        b = self._cba(a, q + "conv_block_2.0", q + "conv_block_2.1", BN_STD, kind="g3x3", stride=stride, act=ACT_RELU)
        if (q + "se.1.weight") in P:
            b = K.SEGate.apply(b, P[q + "se.1.weight"], P[q + "se.1.bias"], P[q + "se.3.weight"], P[q + "se.3.bias"])
        if (q + "shortcut.0.weight") in P:
            s = self._cba(x, q + "shortcut.0", q + "shortcut.1", BN_STD, stride=stride, act=ACT_NONE)
        else:
            s = x
        return self._cba(b, q + "conv_block_3.0", q + "conv_block_3.1", BN_STD, res=s, act=ACT_RELU)

    def _xblock_params(self, q):
        """the 19 parameter / buffer tensors of an identity XBlock in XBlockFn.forward's argument order (ops.xstage.PER_BLOCK)"""
        P = self._idx
        bn = [self._bn(q + f"conv_block_{i}.1")[:4] for i in (1, 2, 3)]
        return [P[q + "conv_block_1.0.weight"], *bn[0], P[q + "conv_block_2.0.weight"], *bn[1], P[q + "se.1.weight"], P[q + "se.1.bias"],
                P[q + "se.3.weight"], P[q + "se.3.bias"], P[q + "conv_block_3.0.weight"], *bn[2]]

    def _xstage_shape_ok(self, q, x):
        P = self._idx
        if (q + "se.1.weight") not in P or (q + "shortcut.0.weight") in P:
            return False
        w1 = P[q + "conv_block_1.0.weight"]
        return K.xblock_fusable(x, w1, 1, True, False) and K.xstage_ok(x, w1, P[q + "se.1.weight"].shape[0])

    def _xstage_run(self, q, x, group):
        """the blocks from `q` to the end of the stage can run as one persistent launch: training-mode identity blocks with SE whose shape the
        kernel covers (hn_xstage_supported), deferred weight gradients in place (the backward walks XBlockFn.backward)"""
        P = self._idx
        if not (self.training and torch.is_grad_enabled() and x.requires_grad and group is not None and (q + "se.1.weight") in P
                and (q + "shortcut.0.weight") not in P):
            return False
        w1 = P[q + "conv_block_1.0.weight"]
        return K.xblock_fusable(x, w1, 1, True, False) and K.xstage_ok(x, w1, P[q + "se.1.weight"].shape[0])

    def _backbone(self, x):
        """AnyNetX.forward, net/anynet.py:136-145: x NCHW fp32 -> list of NHWC bf16 stage outputs."""
        return [a[0] for a, _ in self._backbone_shared(x, (1,) * len(self.depths))]

    def _backbone_shared(self, x, ext, tail=None):
        """-> per stage (aliases, slot): ext[k] aliases of the stage output for its consumers OUTSIDE the backbone (BiFPN input convs, the seg
        decoder's skip operand); the next stage takes one more alias.  Consumers that know the slot add their gradient in place (ops.share)."""
        p = "backbone.net."
        x = x.contiguous().float()
        t = self._cba(x, p + "stem.conv", p + "stem.bn", BN_STD, kind="stem", act=ACT_RELU)
        feats = []
        P = self._idx
        for k, d in enumerate(self.depths):
            # The 1x1 weight gradients of the stage's XBlocks are deferred to the stage boundary and run as ONE grouped launch
            # (ops.DeferredGrads / WgradGroup): the identity node sits on the tensor entering the stage, so its backward runs after the
            # backward of every block of the stage.
            group = None
            names = [f"{p}stage_{k}.blocks.block_{i}.{c}" for i in range(d)
                     for c in ("conv_block_1.0.weight", "conv_block_2.0.weight", "conv_block_3.0.weight", "shortcut.0.weight", "se.1.weight",
                               "se.1.bias", "se.3.weight", "se.3.bias")]
            names = [nm for nm in names if nm in P]
            if (K.DEFER_WGRAD and self.training and torch.is_grad_enabled() and t.requires_grad and t.is_cuda and (self.se_ratio is not None) and
                    all(P[nm].requires_grad for nm in names)):
                group = K.WgradGroup()
                t = K.DeferredGrads.apply(t, group, *[P[nm] for nm in names])
            i = 0
            while i < d:
                q = f"{p}stage_{k}.blocks.block_{i}."
                if i > 0 and self.training and not (torch.is_grad_enabled() and t.requires_grad) and self._xstage_shape_ok(q, t):
                    # training-mode forward without autograd (the backbone under a head-only fine-tuning phase): the persistent forward alone
                    while i < d:
                        j1 = min(d, i + 16)
                        params = []
                        for j in range(i, j1):
                            params += self._xblock_params(f"{p}stage_{k}.blocks.block_{j}.")
                        with torch.no_grad():
                            t = K.xstage_forward_raw(t, params, BN_STD["eps"], BN_STD["momentum"])["out"][j1 - i - 1]
                        i = j1
                    break
                if i > 0 and i < d and self._xstage_run(q, t, group):
                    # blocks i..d-1 are stride-1 identity blocks of one shape: ONE persistent launch for their forward (ops/xstage.py)
                    while i < d:                                   # (a launch takes up to 16 blocks)
                        j1 = min(d, i + 16)
                        params = []
                        for j in range(i, j1):
                            params += self._xblock_params(f"{p}stage_{k}.blocks.block_{j}.")
                        t = K.xstage_apply(t, group, BN_STD["eps"], BN_STD["momentum"], params)
                        i = j1
                    break
                t = self._xblock(q, t, self.backbone_stride if i == 0 else 1, group)
                i += 1
            last = k == len(self.depths) - 1
            if last and tail is not None and K.DEFER_WGRAD and self.training and torch.is_grad_enabled() and t.requires_grad and t.is_cuda:
                # the neck's and the det / lane heads' deferred parameter gradients (ops.GradQueue) are flushed by this node: created after
                # every backbone node and before every neck / head node, so in backward it runs after all of the latter
                t = K.DeferredGrads.apply(t, tail[0], *tail[1])
                tail[2] = True
            al, slot = K.share(t, ext[k] + (0 if last else 1))
            feats.append((al[:ext[k]], slot))
            if not last:
                t = al[-1]
        return feats

    def _sepconv(self, name, x, act=ACT_NONE):
        """SeparableConvBlock with BN (net/common.py:104-114)."""
        P = self._idx
        d = K.DwConv.apply(x, P[name + ".depthwise_conv.conv.weight"])
        return self._cba(d, name + ".pointwise_conv.conv", name + ".bn", BN_FPN, act=act)

    def _fusew(self, name):
        return self._idx[name]                      # raw fusion parameter; relu / normalisation happen inside the Fuse op

    # consumers of a cell's five inputs inside the cell (fusion nodes): P3 feeds one node, P4..P7 two each (net/bifpn.py:186-231)
    CELL_IN_COUNTS = (1, 2, 2, 2, 2)

    def _cell_shared(self, p, inputs, first, ext):
        """BiFPN._forward_fast_attention, net/bifpn.py:156-233.  Every map with more than one consumer goes through ops.share(): the
        fusion nodes accumulate its gradient in place inside their backward kernels instead of leaving k-1 additions per k-consumer
        tensor to the autograd engine.  inputs: backbone features (first cell) or [(aliases, slot)] * 5 from the previous cell;
        ext[l] = consumers of output level l outside this cell; returns [(aliases, slot)] * 5."""
        red = lambda nm, src, i: self._cba(src[0][i], p + nm + ".0.conv", p + nm + ".1", BN_FPN, act=ACT_NONE, slot=src[1])
        sh = K.share
        if first:                                      # inputs: (aliases, slot) per backbone stage, see first_cell_counts()
            if len(inputs) == 4:                       # 4 backbone stages (small cfg): P6 is pooled from P5 (net/bifpn.py:158-160)
                p3, p4, p5 = inputs[-3:]
                p6_in = K.MaxPool.apply(red("p5_to_p6", p5, 2), 0)
            else:                                      # 5 stages (big cfg): the last stage is P6 (net/bifpn.py:162-165)
                p3, p4, p5, p6r = inputs[-4:]
                p6_in = red("p6_down_channel", p6r, 0)
            (p6a, p6b, p6c), s6 = sh(p6_in, 3)         # two fusion nodes + the pool that makes P7
            p7_in = K.MaxPool.apply(p6c, 0)
            p3a, s3 = red("p3_down_channel", p3, 0), None
            (p4a, p4b), s4a, s4b = (red("p4_down_channel", p4, 0), red("p4_down_channel_2", p4, 1)), None, None
            (p5a, p5b), s5a, s5b = (red("p5_down_channel", p5, 0), red("p5_down_channel_2", p5, 1)), None, None
        else:
            (p3a,), s3 = inputs[0]
            (p4a, p4b), s4a = inputs[1]
            (p5a, p5b), s5a = inputs[2]
            (p6a, p6b), s6 = inputs[3]
            s4b, s5b = s4a, s5a
        if first:
            (p7a, p7b), s7 = sh(p7_in, 2)
        else:
            (p7a, p7b), s7 = inputs[4]
        F = K.Fuse.apply
        w = lambda nm: self._fusew(p + nm)
        (u6a, u6b), t6 = sh(self._sepconv(p + "conv6_up", F(w("p6_w1"), 1, 2, 0, p6a, p7a, None, (s6, s7, None))), 2)
        (u5a, u5b), t5 = sh(self._sepconv(p + "conv5_up", F(w("p5_w1"), 1, 2, 0, p5a, u6a, None, (s5a, t6, None))), 2)
        (u4a, u4b), t4 = sh(self._sepconv(p + "conv4_up", F(w("p4_w1"), 1, 2, 0, p4a, u5a, None, (s4a, t5, None))), 2)
        o3, q3 = sh(self._sepconv(p + "conv3_up", F(w("p3_w1"), 1, 2, 0, p3a, u4a, None, (s3, t4, None))), 1 + ext[0])
        o4, q4 = sh(self._sepconv(p + "conv4_down", F(w("p4_w2"), 1, 1, 3, p4b, u4b, o3[0], (s4b, t4, q3))), 1 + ext[1])
        o5, q5 = sh(self._sepconv(p + "conv5_down", F(w("p5_w2"), 1, 1, 3, p5b, u5b, o4[0], (s5b, t5, q4))), 1 + ext[2])
        o6, q6 = sh(self._sepconv(p + "conv6_down", F(w("p6_w2"), 1, 1, 3, p6b, u6b, o5[0], (s6, t6, q5))), 1 + ext[3])
        o7, q7 = sh(self._sepconv(p + "conv7_down", F(w("p7_w2"), 1, 3, 0, p7b, o6[0], None, (s7, q6, None))), ext[4])
        return [(o3[1:], q3), (o4[1:], q4), (o5[1:], q5), (o6[1:], q6), (o7, q7)]

    def first_cell_counts(self):
        """consumers of every backbone stage output inside the first BiFPN cell (its 1x1 input convs, net/bifpn.py:156-197)"""
        n = len(self.depths)
        cnt = [0] * n
        if n == 4:
            cnt[-3:] = [1, 2, 3]                       # P3; P4 (two input convs); P5 (two input convs + p5_to_p6)
        else:
            cnt[-4:] = [1, 2, 2, 1]
        return cnt

    def _cell(self, p, inputs, first):
        """one BiFPN cell on plain tensors (the reference's module surface, per-segment tests): every output has one outside consumer"""
        ins = [((t, t, t), None) for t in inputs] if first else [K.share(t, c) for t, c in zip(inputs, self.CELL_IN_COUNTS)]
        return tuple(a[0] for a, _ in self._cell_shared(p, ins, first, (1, 1, 1, 1, 1)))

    def _neck_shared(self, feats, head_counts):
        """feats: (aliases, slot) per backbone stage (first_cell_counts() aliases each) -> per pyramid level a tuple of head_counts[l]
        aliases of the fused map (one per head that consumes it)"""
        x = list(feats)
        for k in range(self.fpn_cell_repeats):
            last = k == self.fpn_cell_repeats - 1
            x = self._cell_shared(f"neck.bifpn.{k}.", x, k == 0, head_counts if last else self.CELL_IN_COUNTS)
        return [aliases for aliases, _ in x]

    def _neck(self, feats):
        """BiFPN stack on plain tensors: the five fused maps"""
        return [a[0] for a in self._neck_shared([((t, t, t), None) for t in feats], (1, 1, 1, 1, 1))]

    def _seg(self, feats_seg, want_mask=False):
        """SegmentHeader.forward, head_seg/segmentation.py:84-105 -> fp32 logits, NCHW-shaped (channels-last memory); want_mask (deploy mode
        under no_grad): the int64 arg-max mask straight from the output conv's epilogue instead (the logits are never materialised)."""
        P = self._idx
        n = len(feats_seg)
        x = feats_seg[-1]
        p = "segheader.decoder."
        # chain flags: (x0 is the previous block's ELU output, the next block folds this block's ELU' into its data gradient)
        fuse = self.seg_fuse_elu_bwd
        s2d = False
        slot = None      # GradSlot between a phase-form block and the block behind it (the gradient a second time in operand order)
        for i in range(n):
            x = K.SegConv.apply(x, None, P[f"{p}{2 * i}.conv.conv.weight"], P[f"{p}{2 * i}.conv.conv.bias"], 0, ACT_ELU, False,
                                fuse and i > 0, fuse, slot)
            slot = None
            skip = feats_seg[n - 2 - i] if i < n - 1 else None
            wgt = P[f"{p}{2 * i + 1}.conv.conv.weight"]
            if K.seg_up_phase_ok(x, skip, wgt):        # conv over the up-sampled map in phase form on the low-resolution grid
                # last block in front of the phase-form output conv: that conv's data gradient arrives in this block's operand order
                s2d = bool(fuse and i == n - 1 and self.seg_phase_output and self.training and x.requires_grad
                           and not want_mask and K.seg_s2d_handover_ok(x, skip, wgt))
                slot = K.GradSlot() if (fuse and not s2d and self.training and x.requires_grad) else None
                x = K.SegConvUp.apply(x, skip, wgt, P[f"{p}{2 * i + 1}.conv.conv.bias"], fuse, fuse, s2d, slot)
            else:
                x = K.SegConv.apply(x, skip, wgt, P[f"{p}{2 * i + 1}.conv.conv.bias"], 1, ACT_ELU, False, fuse, fuse)
        last = 2 * n
        if want_mask and self.seg_phase_output and K.seg_out_argmax_ok(x, P[f"{p}{last}.conv.weight"]):
            return K.seg_out_argmax(x, P[f"{p}{last}.conv.weight"], P[f"{p}{last}.conv.bias"])
        if self.seg_phase_output:          # final 3x3 over the up-sampled map as a 4-phase conv on the low-resolution grid (ops.SegOutUp)
            # the loss may hand its gradient over in this node's operand form (no fp32 dlogits tensor): see _seg_loss
            slot = K.GradSlot() if (x.requires_grad and self.training) else None
            y = K.SegOutUp.apply(x, P[f"{p}{last}.conv.weight"], P[f"{p}{last}.conv.bias"], fuse, slot, s2d)
            out = y.permute(0, 3, 1, 2)
            # identified by the returned tensor OBJECT (weak reference): an address + shape match can be a recycled allocation
            self._seg_grad_slot = (slot, weakref.ref(out)) if slot is not None else None
            return out
        y = K.SegConv.apply(x, None, P[f"{p}{last}.conv.weight"], P[f"{p}{last}.conv.bias"], 1, ACT_NONE, True, fuse, False)
        return y.permute(0, 3, 1, 2)

    def anchors_for(self, h, w, device):
        """Anchors.forward, head_detect/detection.py:108-170 (host numpy, cached per shape/device)."""
        key = (h, w, str(device))
        if key not in self._anchor_cache:
            d = self.cfgs["detection"]
            r1, r2 = d["aspect_ratios_factor"]
            ratios = [(1.0, 1.0), (r1, r2), (r2, r1)]
            scales = [2 ** v for v in d["scales_factor"]]
            allb = []
            for lv in range(d["pyramid_levels"]):
                stride = 2 ** (lv + 3)
                if w % stride or h % stride:
                    raise ValueError("input size must be divided by the stride.")
                per = []
                for scale, ratio in itertools.product(scales, ratios):
                    base = d["anchor_scale"] * stride * scale
                    hx, hy = base * ratio[0] / 2.0, base * ratio[1] / 2.0
                    xv, yv = np.meshgrid(np.arange(stride / 2, w, stride), np.arange(stride / 2, h, stride))
                    xv, yv = xv.reshape(-1), yv.reshape(-1)
                    per.append(np.stack((yv - hy, xv - hx, yv + hy, xv + hx), axis=1)[:, None, :])
                allb.append(np.concatenate(per, axis=1).reshape(-1, 4))
            a = torch.from_numpy(np.vstack(allb).astype(np.float32)).to(device).unsqueeze(0)
            self._anchor_cache[key] = a
        return self._anchor_cache[key]

    def _streams(self, device, count):
        pool = self._side_streams.setdefault(device, [])
        while len(pool) < count:
            pool.append(torch.cuda.Stream(device=device))
        return pool[:count]

    def _det_tower(self, p, fused, k, act):
        """Regressor / Classifier (head_detect/detection.py:26-44, 63-83), level by level (dw 3x3 -> 1x1+BN statistics -> BN+Swish, three
        times per level).  Used when the levels cannot be packed (ops.levels_packable); levels_on_streams (off: hipGraph branches measured
        slower on gfx950) would put each level on its own HIP stream, forked from / joined to the caller's stream with events."""
        P = self._idx
        layers = self.cfgs["detection"]["box_class_repeats"]
        dev = fused[0].device
        multi = self.levels_on_streams and fused[0].is_cuda
        cur = torch.cuda.current_stream() if multi else None
        streams = self._streams(dev, len(fused)) if multi else [None] * len(fused)
        # pack the shared weights once, on the caller's stream, before the fork
        for i in range(layers):
            K.pack_dw_weight(P[f"{p}conv_list.{i}.depthwise_conv.conv.weight"])
            K.pack_conv_weight(P[f"{p}conv_list.{i}.pointwise_conv.conv.weight"])
        outs = []
        for lv, f in enumerate(fused):
            st = streams[lv]
            if st is not None:
                st.wait_stream(cur)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                for i in range(layers):
                    d = K.DwConv.apply(f, P[f"{p}conv_list.{i}.depthwise_conv.conv.weight"])
                    f = self._cba(d, f"{p}conv_list.{i}.pointwise_conv.conv", f"{p}bn_list.{lv}.{i}", BN_FPN, act=ACT_SWISH)
            outs.append(f)
        if multi:
            for st, f in zip(streams, outs):
                cur.wait_stream(st)
                f.record_stream(cur)
        return K.HeadOut.apply(P[p + "header.depthwise_conv.conv.weight"], P[p + "header.pointwise_conv.conv.weight"],
                               P[p + "header.pointwise_conv.conv.bias"], k, act, *outs)

    def _det_tower_packed(self, p, xp, geom, k, act, slot=None):
        """Regressor / Classifier on level-packed rows: one launch per op for all five levels (ops.TowerLayer)."""
        P = self._idx
        layers = self.cfgs["detection"]["box_class_repeats"]
        f = xp
        for i in range(layers):
            bn = []
            for lv in range(len(geom[1])):
                g, b, rm, rv, _ = self._bn(f"{p}bn_list.{lv}.{i}")
                bn += [g, b, rm, rv]
            f = K.TowerLayer.apply(f, P[f"{p}conv_list.{i}.depthwise_conv.conv.weight"], P[f"{p}conv_list.{i}.pointwise_conv.conv.weight"],
                                   P.get(f"{p}conv_list.{i}.pointwise_conv.conv.bias"), geom, ACT_SWISH, BN_FPN["eps"], BN_FPN["momentum"],
                                   self.training, slot if i == 0 else None, *bn)
        return K.HeadOutPacked.apply(P[p + "header.depthwise_conv.conv.weight"], P[p + "header.pointwise_conv.conv.weight"],
                                     P[p + "header.pointwise_conv.conv.bias"], k, act, geom, f)

    def _det(self, x, fused):
        anchors = self.anchors_for(x.shape[2], x.shape[3], x.device)
        ncls = self.cfgs["detection"]["num_classes"]
        if self.pack_det_levels and K.levels_packable(fused):
            geom = (fused[0].shape[0], tuple(f.shape[1] for f in fused), tuple(f.shape[2] for f in fused))
            (xr, xc), slot = K.share(K.PackLevels.apply(*fused), 2)       # both towers read the packed map
            reg = self._det_tower_packed("detectheader.regressor.", xr, geom, 4, ACT_NONE, slot)
            cls = self._det_tower_packed("detectheader.classifier.", xc, geom, ncls, ACT_SIGMOID, slot)
        else:
            reg = self._det_tower("detectheader.regressor.", fused, 4, ACT_NONE)
            cls = self._det_tower("detectheader.classifier.", fused, ncls, ACT_SIGMOID)
        return anchors, reg, cls

    def _lane(self, fused):
        """LaneHeader.forward, head_lane/lanedetect.py:66-96."""
        P = self._idx
        stride = self.cfgs["lane"]["anchor_stride"]
        assert stride == 32, "only the stride-32 lane fusion of the shipped cfgs is on the hot path"
        fl, slot = K.share(K.LaneConcat.apply(fused[0], fused[1], fused[2], fused[3]), 3)      # three branches read the fused map
        fl = list(fl)

        def trunk(nm):
            q = f"laneheader.{nm}."
            return self._cba(fl.pop(), q + "0", q + "1", BN_STD, act=ACT_RELU, slot=slot), P[q + "3.weight"], P[q + "3.bias"]
        t, w, b = trunk("conv_cls_conv")
        cls = K.HeadOut.apply(None, w, b, w.shape[0], ACT_NONE, t)
        tu, wu, bu = trunk("conv_up_conv")
        td, wd, bd = trunk("conv_down_conv")
        # predict_loc = cat([down, up], -1) (lanedetect.py:93): the two 1x1 convs write side by side into one tensor
        return dict(predict_cls=cls, predict_loc=K.HeadOutCat.apply(wd, bd, wu, bu, td, tu))

    # ------------------------------------------------------------------------------------------------------
    def forward(self, x, mode="train"):
        """HydraNet.forward, model/model.py:159-198."""
        if self._folded is not None and not self.training:
            self._check_folded()
        params = {id(t) for t in self.parameters()}
        if self.training:
            K.bump_mutation_epoch([self._mut_cell])   # the BatchNorm kernels update this module's running statistics through raw pointers
        if self._pack_plan is not None and not self._pack_plan.valid():
            self._pack_plan = None                 # a parameter's storage was replaced: re-record the weights on this forward
        # eval mode with unchanged parameters (serving): the packed operands of the last forward are still right -- no pack launches
        reuse = not self.training and self._pack_plan is not None and x.is_cuda and self._pack_plan.fresh()
        if not reuse:
            K.clear_pack_cache()
        if reuse:
            pass
        elif self._pack_plan is not None and x.is_cuda:
            self._pack_plan.run()                  # every dense conv weight -> bf16 operands, one launch
        elif x.is_cuda:
            K.start_pack_log()                     # first forward on this device: record which weights get packed
        try:
            return self._forward(x, mode)
        finally:
            if self._pack_plan is None and x.is_cuda:
                log = [e for e in K.stop_pack_log() if id(e[1]) in params]
                if log:
                    self._pack_plan = K.PackPlan(log)

    def _forward(self, x, mode):
        # grad_scope ("lane" | "det" | "seg", set by HydraTrainer.set_phase for the head-only phases of train.py:441-515): only that head's
        # parameters are being optimised, so everything else runs forward only (no autograd graph: no saved activations, no backward
        # launches); BatchNorm running statistics update as in any training-mode forward and every output is still produced.
        scope = self.grad_scope if (self.training and torch.is_grad_enabled()) else None
        off = lambda part: torch.no_grad() if (scope is not None and scope != part) else contextlib.nullcontext()
        neck_cnt = self.first_cell_counts()
        seg_skip = 1 if self.train_seg else 0          # the seg decoder's last skip operand is the stage-0 output
        # consumers of every fused pyramid level among the heads: det towers (all five), seg decoder (P3..P5), lane fusion (P3..P6)
        users = [[h for h, on, lv in (("det", self.train_detect, range(5)), ("seg", self.train_seg, range(3)), ("lane", self.train_lane, range(4)))
                  if on and l in lv] for l in range(5)]
        tail = [K.GradQueue(), self._tail_params, False]
        with off("shared"):
            bb = self._backbone_shared(x, tuple(c + (seg_skip if k == 0 else 0) for k, c in enumerate(neck_cnt)), tail)
            feats = [(a[(seg_skip if k == 0 else 0):], s) for k, (a, s) in enumerate(bb)]      # what the neck sees
            feat0_seg = bb[0][0][0] if seg_skip else None
            K.set_queue(tail[0] if tail[2] else None)    # the neck's weight / fusion gradients wait for the flush node behind the backbone
            try:
                al = self._neck_shared(feats, tuple(max(len(u), 1) for u in users))
            finally:
                K.set_queue(None)
        pick = lambda head: [al[l][users[l].index(head)] if head in users[l] else al[l][0] for l in range(5)]
        fused_det, fused_seg, fused_lane = pick("det"), pick("seg"), pick("lane")
        out = {}
        seg = anchors = reg = cls = lane_cls = lane_reg = None
        # The detection and lane heads are chains of small launches that are independent of the (large) segmentation decoder: they run on a
        # side HIP stream (forked / joined with events, so the fork is also legal inside hipGraph capture); autograd replays the same
        # stream assignment in backward.
        side = None
        if self.heads_on_side_stream and x.is_cuda and (self.train_detect or self.train_lane) and self.train_seg:
            cur = torch.cuda.current_stream()
            side = self._streams(x.device, 6)[5]
            side.wait_stream(cur)
        K.set_queue(tail[0] if (tail[2] and side is None) else None)
        try:
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if self.train_detect:
                    with off("det"):
                        anchors, reg, cls = self._det(x, fused_det)
                    out["detection"] = {"anchors": anchors, "regression": reg, "classification": cls}
                if self.train_lane:
                    with off("lane"):
                        lane = self._lane(fused_lane)
                    out["lane"] = lane
                    lane_cls, lane_reg = lane["predict_cls"], lane["predict_loc"]
        finally:
            K.set_queue(None)
        seg_mask = None
        if self.train_seg:
            with off("seg"):
                seg = self._seg([feat0_seg, fused_seg[0], fused_seg[1], fused_seg[2]], want_mask=(mode == "deploy" and not torch.is_grad_enabled()))
            if seg.dtype == torch.int64:                   # deploy + no_grad: the output conv already took the arg-max
                seg_mask, seg = seg, None
            out["seg"] = seg
        if side is not None:
            cur.wait_stream(side)
            for t in (reg, cls, lane_cls, lane_reg):
                if t is not None:
                    t.record_stream(cur)
        self._flush_nbt()
        if mode != "deploy":
            return out
        dep = (seg_mask if seg_mask is not None else K.argmax_channels(seg), anchors, reg, cls, lane_cls, lane_reg)
        if self.deploy_postprocess is not None and self.train_detect:
            # decode + clip + threshold + class-offset NMS + gather on the device, same stream, no host round trip;
            # a 7th element (dict of device tensors: rois / class_ids / scores [N, cap, ...], kept / total [N]) follows the reference's 6-tuple
            from .postprocess import postprocess_device
            conf, iou = self.deploy_postprocess[:2]
            dep = dep + (postprocess_device((x.shape[2], x.shape[3]), anchors, reg, cls, conf, iou,
                                            *(self.deploy_postprocess[2:3] or (4096,))),)
        return dep

    def _seg_loss(self, logits, target):
        """CrossEntropyLoss.forward (head_seg/segmentation_loss.py:27-65) on HIP kernels: the weighted-CE / top-k path of the big cfgs
        (hn_seg_loss_*) and the focal variant of the small cfg (hn_seg_focal_*).  No CPU fallback: raises off-device."""
        use_top_k, ratio, use_focal = self._seg_cfg
        key = getattr(self, "_seg_grad_slot", None)
        self._seg_grad_slot = None                                    # consumed (or dropped) by the first loss call after the forward
        if use_focal:
            # (the reference always hands gt_seg.long() to the loss, model.py:212; to_gpu delivers float32 class ids: both are accepted)
            return K.seg_focal_loss_hip(logits, target, self._seg_class_weight)
        slot = None
        if key is not None and key[1]() is logits:
            slot = key[0]                                             # this forward's own "seg" output
        return K.seg_loss_hip(logits, target, self._seg_class_weight, use_top_k, ratio, slot=slot)

    def _guard(self, value, what, allow_zero=False):
        if self.check_finite and ((not allow_zero and value == 0) or not torch.isfinite(value)):
            print(what)
            sys.exit()

    def cal_loss(self, pred_dict, gt_dict):
        """HydraNet.cal_loss, model/model.py:201-264 (same keys, same divergence guard)."""
        ld = {}
        if self.train_seg:
            gt_seg = gt_dict["gt_seg"]
            loss_seg = self.loss_seg(pred_dict["seg"], gt_seg if gt_seg.dtype == torch.float32 and gt_seg.is_cuda else gt_seg.long())
            self._guard(loss_seg, "cal segment loss diverge!")
            ld["loss_seg"] = loss_seg
        if self.train_detect:
            d = pred_dict["detection"]
            cl, rl = self.loss_detect(d["classification"], d["regression"], d["anchors"], gt_dict["gt_det"])
            # the reference takes .mean() of the [1]-shaped batch means (model.py:226-227): a reshape for one element
            cl, rl = (cl.reshape(()) if cl.numel() == 1 else cl.mean()), (rl.reshape(()) if rl.numel() == 1 else rl.mean())
            self._guard(cl, "cal det cls loss diverge!", allow_zero=True)
            self._guard(rl, "cal det reg loss diverge!", allow_zero=True)
            ld["loss_det_cls"], ld["loss_det_reg"] = cl, rl
        if self.train_lane:
            pos, neg, pmask, pnum = self.loss_cls(gt_dict["gt_cls"], pred_dict["lane"]["predict_cls"])
            loc = self.loss_reg(pmask, pnum, gt_dict["gt_loc"], pred_dict["lane"]["predict_loc"],
                                points_per_line=self.lane_points_per_line)
            self._guard(pos, "cal lane pos loss diverge!")
            self._guard(neg, "cal lane neg loss diverge!")
            self._guard(loc, "cal lane loc loss diverge!")
            ld["loss_lane_cls_pos"], ld["loss_lane_cls_neg"], ld["loss_lane_loc"] = pos, neg, loc
        return ld

    def total_loss(self, ld):
        """HydraTrainer.cal_total_loss, model/train.py:192-203."""
        c = self.cfgs
        if all(isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 for v in ld.values()):
            groups = []                                      # one launch (fwd) + one (bwd) instead of ~22 scalar torch kernels
            if self.train_seg:
                groups.append((1.0, [(ld["loss_seg"], c["segment"]["segment_weight"])]))
            if self.train_detect:
                d = c["detection"]
                groups.append((d["detection_weight"], [(ld["loss_det_cls"], d["loss_cls_weight"]), (ld["loss_det_reg"], d["loss_reg_weight"])]))
            if self.train_lane:
                l = c["lane"]
                groups.append((l["lane_weight"], [(ld["loss_lane_cls_pos"], l["loss_cls_pos_weight"]),
                                                  (ld["loss_lane_cls_neg"], l["loss_cls_neg_weight"]), (ld["loss_lane_loc"], l["loss_loc_weight"])]))
            return K.weighted_loss_sum(groups)
        tot = 0.0
        if self.train_seg:
            tot = tot + ld["loss_seg"] * c["segment"]["segment_weight"]
        if self.train_detect:
            d = c["detection"]
            tot = tot + (ld["loss_det_cls"] * d["loss_cls_weight"] + ld["loss_det_reg"] * d["loss_reg_weight"]) * d["detection_weight"]
        if self.train_lane:
            l = c["lane"]
            tot = tot + (ld["loss_lane_cls_pos"] * l["loss_cls_pos_weight"] + ld["loss_lane_cls_neg"] * l["loss_cls_neg_weight"]
                         + ld["loss_lane_loc"] * l["loss_loc_weight"]) * l["lane_weight"]
        return tot


def K_to_nchw(t):
    return t.permute(0, 3, 1, 2)


def K_to_nhwc(t):
    """accept an NCHW-shaped tensor (any memory format / dtype) and return dense NHWC bf16."""
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def _unavailable(what):
    def f(*a, **k):
        raise NotImplementedError(what + " is outside the forward/backward hot path (SURVEY.md section 8f)")
    return f


def _det_decode(imgs, regressions, classifications, anchors, conf_thres=0.6, iou_thres=0.3):
    from .postprocess import postprocess
    if imgs is None:
        return None
    return postprocess((imgs.shape[2], imgs.shape[3]), torch.stack([anchors[0]] * imgs.shape[0], 0).detach(), regressions.detach(),
                       classifications.detach(), conf_thres, iou_thres)
