"""Full-size (big cfg) parity, part 3: ABSOLUTE bf16-vs-fp32 statements on a well-conditioned state.

The random-weight state of parts 1-2 amplifies any perturbation ~1.5x per XBlock (x200 over the 30-block backbone), so end-to-end
comparisons there can only be made relative to what bf16 storage itself does.  Here every XBlock's last BatchNorm scale
(`conv_block_3.1.weight`, the residual branch) is multiplied by 0.1 -- the zero-init-residual practice of trained residual networks
(tests/helpers.conditioned_state) -- and the HIP path (bf16 storage, fp32 accumulation) is held to absolute tolerances:

  1. against the REFERENCE itself: big cfg at the repo-default 640x640, B = 1 -- the reference's recorded digests of this state
     (tests/golden/big_cond.npz, make_golden.py::big_digest(cond=True)): the six training-mode losses, the norms of the head outputs and
     of all 693 parameter gradients, the eval-mode feature / fused / head-output norms and the arg-max class histogram;
  2. against the UNMIRRORED fp32 oracle executed on the device, element by element, at BASELINE's 3x512x1024 (N = 2), in eval AND in
     training mode: every backbone feature map, fused map, seg logits, regression, classification, lane outputs in max-norm and relative
     L2, the six losses, all 693 parameter-gradient norms, the seg arg-max mask agreement.

What bounds the tolerances (measured with PyTorch's own CPU kernels, oracle bf16-mirror vs oracle fp32 on this state):
one bf16 rounding is 2^-9 relative; the backbone stores ~5 tensors per block (30 blocks), each BiFPN cell adds ~1e-2 on identical inputs
and passes input noise on with gain ~1.5, the heads add 3-9 more stored layers.  The bf16 noise floor itself is therefore ~1e-2 at
stages 0-2, ~3e-2 at stage 4, ~5e-2 behind the three BiFPN cells (SURVEY 8(c)'s 2e-2 guess holds for the shallow half only).
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import ROOT, conditioned_state, load_cfg, load_npz
from tests.test_fullsize_gpu import H, W, nchw, rel

pytestmark = pytest.mark.gpu

# ---- stated tolerances ------------------------------------------------------------------------------------------------------------
# element-wise, max|hip - fp32| / max|fp32| and relative L2, per tensor group
# (measured, gpurun_out/fullsize3_*.json: backbone features 0.8 / 1.3 / 0.9 / 1.8 / 2.6e-2 max-norm -- SURVEY 8(c)'s 2e-2 for stages 0-3;
#  fused maps 6-11e-2, seg / regression 1.0e-1, lane 7-8e-2, classification 3.4e-2 in L2 (its max-norm is set by single sigmoid outputs);
#  losses <= 1.5e-2; the oracle's own bf16-mirror run differs from its fp32 run by the same amounts)
# (feat4's max-norm is set by single elements: 2.1 ... 4.4e-2 across GEMM accumulation-order variants at an unchanged L2 of 2.1-2.2e-2)
TOL_MAX = dict(feat0=2e-2, feat1=2e-2, feat2=2e-2, feat3=3e-2, feat4=6e-2, fused=1.4e-1, seg=1.3e-1, regression=1.3e-1, lane_cls=1.1e-1,
               lane_loc=1.1e-1)
TOL_L2 = dict(feat0=1e-2, feat1=1.5e-2, feat2=2e-2, feat3=2.5e-2, feat4=3.5e-2, fused=1e-1, seg=1e-1, regression=1e-1, classification=5e-2,
              lane_cls=1e-1, lane_loc=1e-1)
TOL_LOSS = 3e-2                     # each of the six losses and the total
# | ||g_hip|| - ||g_fp32|| | / ||g_fp32|| over the parameter gradients (fusion weights and exactly-zero gradients aside):
TOL_GRAD_NORM_MEDIAN = 5e-2         # SURVEY 8(c)'s 5e-2 holds for the median (measured 1.3e-2 at 640x640 B=1, 3.5e-2 at 512x1024 N=2) ...
TOL_GRAD_NORM_P90 = 1.5e-1          # ... 90 % of the tensors stay within 1.5e-1 ...
TOL_GRAD_NORM = 7e-1                # ... and the worst of the rest within 7e-1
TOL_GRAD_NORM_SE0 = 1.0             # the 6..16-wide SE squeeze layers of the stages' FIRST blocks (se.1.* of block_0: cin / 4 hidden units, batch
                                    # of ONE image at 640x640): sums with cancellation, no stable norm under any change of a summation order
                                    # (measured 0.70 with stages 2-4 on the unfused XBlock composition, 0.89 on the fused node; median / p90 of
                                    # all tensors 1.2e-2 / 5.3e-2 either way)
GRAD_COS = 0.88                     # cosine of individual gradients vs fp32 (measured: heads 0.996-1.0, neck 0.98, backbone 0.91-0.95)
TOL_FUSION = 0.6                    # the 24 BiFPN fusion-weight gradients: | ||g_hip|| - ||g_fp32|| | / max_j ||g_fp32_j|| (see _is_fusion_weight)
MASK_AGREEMENT = 0.95               # fraction of pixels whose arg-max class equals the fp32 oracle's
MIRROR_FACTOR, MIRROR_FLOOR = 1.25, 5e-3   # HIP-vs-fp32 relative L2 <= 1.25 x (oracle bf16-mirror vs fp32) + 5e-3, per tensor, same state and batch
# digests vs the reference (norms only): |L2_hip - L2_ref| / L2_ref
TOL_DIGEST = dict(feat=2e-2, fused=4e-2, head=4e-2)


def dump(name, obj):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(obj, open(os.path.join(ROOT, "gpurun_out", f"fullsize3_{name}.json"), "w"), indent=1)
    print(name, json.dumps(obj)[:3000])


def _group(k):
    return "fused" if k.startswith("fused") else k


def _state(net=None):
    """net None: the 640x640 state the reference was run on (sha256-pinned); else the same seeded recipe on `net`'s own key / shape list
    (the lane head's output width follows the input height: 2 * H / 8 + 2 columns)"""
    z = load_npz("big_cond.npz")
    keys = z["keys"].tolist()
    if net is None:
        shapes = [tuple(int(v) for v in s.split(",")) if s else () for s in z["shapes"].tolist()]
    else:
        own = net.state_dict()
        assert list(own.keys()) == keys
        shapes = [tuple(v.shape) for v in own.values()]
    sd = conditioned_state(keys, shapes, seed=11)
    if net is None:
        sha = hashlib.sha256(b"".join(np.ascontiguousarray(sd[k].numpy()).tobytes() for k in keys)).hexdigest()
        assert sha == str(z["digest/state_sha256"])                    # exactly the state the reference was run on
    return z, sd


def _is_fusion_weight(name):
    """BiFPN fusion parameters p{3..7}_w{1,2}: d/dp_i = (dw_i - sum_j w_j dw_j) / (sum relu(p) + eps) is a difference of nearly equal sums,
    so the NORM of that gradient is not a stable quantity under any rounding (the terms themselves are checked at 3e-2 in
    tests/test_kernels_gpu.py::test_bifpn_fuse); they are reported, and bounded relative to the largest fusion-weight gradient instead"""
    import re
    return re.search(r"\.p\d_w\d$", name) is not None


def _grad_norm_report(names, ghip, gref):
    gerr = np.abs(ghip - gref) / np.maximum(gref, 1e-30)
    fus = np.array([_is_fusion_weight(k) for k in names])
    skip = (gref < 1e-6 * gref.max()) | fus                            # biases in front of BatchNorm: mathematically zero gradients
    fscale = float(gref[fus].max())
    se0 = np.array([".block_0.se.1." in k for k in names]) & ~skip
    rest = ~skip & ~se0
    order = np.argsort(-np.where(skip, 0, gerr))[:6]
    return dict(max=float(gerr[rest].max()), max_se_squeeze_first_blocks=float(gerr[se0].max()) if se0.any() else 0.0,
                median=float(np.median(gerr[~skip])), p90=float(np.quantile(gerr[~skip], 0.9)), n=int((~skip).sum()),
                worst=[(names[i], round(float(gerr[i]), 4)) for i in order],
                fusion_weights_abs_err_over_largest=float(np.abs(ghip[fus] - gref[fus]).max() / fscale))


def _check_grad_norms(r):
    assert r["median"] <= TOL_GRAD_NORM_MEDIAN and r["p90"] <= TOL_GRAD_NORM_P90 and r["max"] <= TOL_GRAD_NORM, r
    assert r["max_se_squeeze_first_blocks"] <= TOL_GRAD_NORM_SE0, r
    assert r["fusion_weights_abs_err_over_largest"] <= TOL_FUSION, r


def _net(h, w):
    import __graft_entry__ as g
    g.build()
    from multitask_hydranet_amd import HydraNet
    cfgs = load_cfg("hydranet_big.yml")
    cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
    net = HydraNet(cfgs)
    return net, cfgs


def test_conditioned_state_vs_reference_digests_640():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from oracle import hydranet_oracle as O
    z, sd = _state()
    net, cfgs = _net(640, 640)
    net.load_state_dict(sd)
    net = net.to("cuda:0").train()
    assert net.lane_points_per_line == 160                              # the reference's live default at 640x640
    batch = {k: v.to("cuda:0") for k, v in O.synthetic_batch(cfgs, 1, 640, 640, seed=1).items()}
    out = net(batch["image"])
    ld = net.cal_loss(out, batch)
    tot = net.total_loss(ld)
    tot.backward()
    torch.cuda.synchronize()
    res = {"loss": {k: (float(v), float(z["digest/loss/" + k])) for k, v in ld.items()}}
    res["loss"]["total"] = (float(tot), float(z["digest/loss/total"]))
    dig = lambda t: [float(t.detach().double().mean()), float(t.detach().abs().max()), float(t.detach().double().norm())]
    heads = {"seg": out["seg"], "regression": out["detection"]["regression"], "classification": out["detection"]["classification"],
             "lane_cls": out["lane"]["predict_cls"], "lane_loc": out["lane"]["predict_loc"]}
    res["train"] = {k: (dig(t), z["digest/train/" + k].tolist()) for k, t in heads.items()}
    P = dict(net.named_parameters())
    gk, gl = z["digest/grad_keys"].tolist(), z["digest/grad_l2"]
    assert {k for k, p in P.items() if p.grad is not None} == set(gk) and len(gk) == 693
    ghip = np.array([float(P[k].grad.double().norm()) for k in gk])
    res["grad_norm_rel_err"] = _grad_norm_report(gk, ghip, gl)
    # eval mode with the post-step running statistics, as the reference recorded it
    net.eval()
    with torch.no_grad():
        feats = net._backbone(batch["image"])
        fused = net._neck(feats)
        dep = net(batch["image"], "deploy")
    net.train()
    ev = {f"feat{i}": nchw(f) for i, f in enumerate(feats)}
    ev.update({f"fused{i}": nchw(f) for i, f in enumerate(fused)})
    ev.update(regression=dep[2], classification=dep[3], lane_cls=dep[4], lane_loc=dep[5])
    res["eval"] = {k: (dig(t), z["digest/eval/" + k].tolist()) for k, t in ev.items()}
    hist = torch.bincount(dep[0].flatten(), minlength=5).cpu().numpy()
    ref_hist = z["digest/eval/seg_argmax_hist"]
    res["eval_argmax_hist"] = (hist.tolist(), ref_hist.tolist())
    dump("digests_640", res)
    for k, (a, b) in res["loss"].items():
        assert abs(a - b) <= TOL_LOSS * abs(b), (k, a, b)
    for k, (got, ref) in res["train"].items():
        assert abs(got[2] - ref[2]) <= TOL_DIGEST["head"] * ref[2] and abs(got[1] - ref[1]) <= 2.5 * TOL_DIGEST["head"] * ref[1], (k, got, ref)
    for k, (got, ref) in res["eval"].items():
        tol = TOL_DIGEST["feat" if k.startswith("feat") else ("fused" if k.startswith("fused") else "head")]
        assert abs(got[2] - ref[2]) <= tol * ref[2], (k, got, ref)
    _check_grad_norms(res["grad_norm_rel_err"])
    assert np.abs(hist - ref_hist).sum() <= (1.0 - MASK_AGREEMENT) * 640 * 640


@pytest.mark.parametrize("training", [False, True])
def test_conditioned_state_elementwise_vs_fp32_oracle_512x1024(training):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import bench
    from oracle import hydranet_oracle as O
    net, cfgs = _net(H, W)
    _, sd0 = _state(net)
    batch = bench.synthetic_batch(cfgs, 2, H, W, seed=3, device="cuda:0")
    ppl = H // cfgs["lane"]["interval"]
    sd = {k: v.to("cuda:0") for k, v in sd0.items()}
    # running statistics = the statistics of this very batch (one oracle training pass with momentum 1): eval mode then sees normalised
    # inputs in every layer, as it would after training
    keep = (dict(O.BN_BACKBONE), dict(O.BN_NECK))
    try:
        O.BN_BACKBONE["momentum"] = 1.0
        O.BN_NECK["momentum"] = 1.0
        with torch.no_grad():
            O.hydranet_forward(sd, cfgs, batch["image"], training=True)
    finally:
        O.BN_BACKBONE.update(keep[0])
        O.BN_NECK.update(keep[1])
    net.load_state_dict({k: v.cpu() for k, v in sd.items()})
    net = net.to("cuda:0")
    net.check_finite = False
    net.lane_points_per_line = ppl
    net.train(training)
    osd = {k: v.clone() for k, v in sd.items()}
    if training:
        for k, v in osd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
    # The bf16 noise floor, measured in the same run instead of asserted in prose: the oracle -- PyTorch's own kernels --
    # in bf16-MIRROR mode (rounding wherever the HIP path stores bf16) on the same state and batch, and the same with the BiFPN cells kept
    # in fp32 (what fp32 storage of the 24 fusion nodes + fused maps would buy).  Forward only (running statistics restored afterwards).
    def mirror_run(fp32_parts):
        msd = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad(), O.bf16_mirror(fp32_parts):
            return O.hydranet_forward(msd, cfgs, batch["image"], training=training, want_features=True)
    mir, mir_neck32 = mirror_run(()), mirror_run(("neck",))
    with (torch.enable_grad() if training else torch.no_grad()):
        ref = O.hydranet_forward(osd, cfgs, batch["image"], training=training, want_features=True)          # UNMIRRORED fp32
        feats = net._backbone(batch["image"])
        fused = net._neck(feats)
        net._flush_nbt()
        net.load_state_dict({k: v.cpu() for k, v in sd.items()})
        out = net(batch["image"])
        res = {}
        if training:
            rld = O.hydranet_losses(cfgs, ref, batch, lane_points_per_line=ppl)
            rtot = O.total_loss(cfgs, rld)
            rtot.backward()
            ld = net.cal_loss(out, batch)
            tot = net.total_loss(ld)
            tot.backward()
            res["loss"] = {k: (float(ld[k]), float(rld[k])) for k in rld}
            res["loss"]["total"] = (float(tot), float(rtot))
            P = dict(net.named_parameters())
            names = [k for k, p in P.items() if p.grad is not None]
            assert len(names) == 693 and all(osd[k].grad is not None for k in names)
            gref = np.array([float(osd[k].grad.double().norm()) for k in names])
            ghip = np.array([float(P[k].grad.double().norm()) for k in names])
            cos = {k: float(torch.nn.functional.cosine_similarity(P[k].grad.flatten().double(), osd[k].grad.flatten().double(), dim=0))
                   for k in ("backbone.net.stem.conv.weight", "backbone.net.stage_2.blocks.block_0.conv_block_1.0.weight",
                             "backbone.net.stage_4.blocks.block_13.conv_block_3.0.weight", "neck.bifpn.2.conv3_up.pointwise_conv.conv.weight",
                             "segheader.decoder.8.conv.weight", "detectheader.classifier.header.pointwise_conv.conv.weight",
                             "laneheader.conv_up_conv.3.weight")}
            res["grad_norm_rel_err"] = _grad_norm_report(names, ghip, gref)
            res["grad_cosine"] = cos

    def tensors(o):
        t = {f"feat{i}": f for i, f in enumerate(o["_feats"])}
        t.update({f"fused{i}": f for i, f in enumerate(o["_fused"])})
        t.update(seg=o["seg"], regression=o["detection"]["regression"], classification=o["detection"]["classification"],
                 lane_cls=o["lane"]["predict_cls"], lane_loc=o["lane"]["predict_loc"])
        return t
    mine = dict(out)
    mine["_feats"], mine["_fused"] = [nchw(f) for f in feats], [nchw(f) for f in fused]
    tm, tr = tensors(mine), tensors(ref)
    l2 = lambda a, b: float((a.detach().float() - b.detach().float()).norm() / b.detach().float().norm().clamp(min=1e-20))
    res["tensors"] = {k: dict(max=rel(tm[k], tr[k]), l2=l2(tm[k], tr[k])) for k in tr}
    res["seg_mask_agreement"] = float((torch.argmax(out["seg"], 1) == torch.argmax(ref["seg"], 1)).float().mean())
    tmir, tn32 = tensors(mir), tensors(mir_neck32)
    res["mirror_vs_fp32"] = {k: dict(max=rel(tmir[k], tr[k]), l2=l2(tmir[k], tr[k])) for k in tr}
    res["mirror_fp32_neck_vs_fp32"] = {k: dict(max=rel(tn32[k], tr[k]), l2=l2(tn32[k], tr[k])) for k in tr}
    res["mirror_seg_mask_agreement"] = float((torch.argmax(mir["seg"], 1) == torch.argmax(ref["seg"], 1)).float().mean())
    res["mirror_fp32_neck_seg_mask_agreement"] = float((torch.argmax(mir_neck32["seg"], 1) == torch.argmax(ref["seg"], 1)).float().mean())
    res["hip_over_mirror_l2"] = {k: res["tensors"][k]["l2"] / max(res["mirror_vs_fp32"][k]["l2"], 1e-12) for k in tr}
    dump("elementwise_%s" % ("train" if training else "eval"), res)
    # the HIP path is no further from fp32 than PyTorch's own bf16-storage realisation of the same network: 1.25 x + a 5e-3 floor
    # (relative L2; the max-norm ratio of two chaotic bf16 realisations scatters and is only recorded)
    for k in tr:
        assert res["tensors"][k]["l2"] <= MIRROR_FACTOR * res["mirror_vs_fp32"][k]["l2"] + MIRROR_FLOOR, (k, res["tensors"][k], res["mirror_vs_fp32"][k])
    assert res["seg_mask_agreement"] >= res["mirror_seg_mask_agreement"] - 1.5e-2, (res["seg_mask_agreement"], res["mirror_seg_mask_agreement"])
    for k, v in res["tensors"].items():
        g = _group(k)
        if g in TOL_MAX:
            assert v["max"] <= TOL_MAX[g], (k, v)
        assert v["l2"] <= TOL_L2[g], (k, v)
    # (the eval-mode case of the conditioned state sits ON the floor: PyTorch's own bf16 mirror agrees with the fp32 oracle on 95.4 % of the
    # pixels, the HIP path on 95.0-95.3 % from build to build -- 0.9498 with DPP wave sums; the bound relative to the mirror above is the
    # parity statement, this one only catches a collapse)
    assert res["seg_mask_agreement"] >= MASK_AGREEMENT - 5e-3, res["seg_mask_agreement"]
    if training:
        for k, (a, b) in res["loss"].items():
            assert abs(a - b) <= TOL_LOSS * abs(b), (k, a, b)
        _check_grad_norms(res["grad_norm_rel_err"])
        assert min(res["grad_cosine"].values()) >= GRAD_COS, res["grad_cosine"]
