"""CPU: the oracle (oracle/hydranet_oracle.py) against vectors recorded from the reference itself."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import hydranet_oracle as O
from tests.helpers import assert_close, load_cfg, load_npz, tiny_state

FP32_TOL = 2e-5   # fp32 CPU oracle vs fp32 CPU reference: identical ATen ops, only graph-order differences


# tiny: 5 backbone stages, top-k CE (the big cfgs' family); tiny4: 4 stages -> p5_to_p6 path of the first BiFPN cell (net/bifpn.py:158-160),
# focal seg loss (segmentation_loss.py:31-46), the loss weights of the reference's hydranet_joint_small_backbone.yml
VARIANTS = {"tiny": ("tiny_hydranet.npz", "hydranet_tiny.yml"), "tiny4": ("tiny4_hydranet.npz", "hydranet_tiny4.yml")}


@pytest.fixture(scope="module", params=["tiny", "tiny4"])
def tiny(request):
    z = load_npz(VARIANTS[request.param][0])
    cfgs = load_cfg(VARIANTS[request.param][1])
    sd = tiny_state(z)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in/")}
    out = O.hydranet_forward(sd, cfgs, batch["image"], training=True, want_features=True)
    ld = O.hydranet_losses(cfgs, out, batch, lane_points_per_line=int(z["meta/lane_points_per_line"]))
    tot = O.total_loss(cfgs, ld)
    tot.backward()
    return z, cfgs, sd, batch, out, ld, tot


def test_tiny_features_and_outputs(tiny):
    z, cfgs, sd, batch, out, ld, tot = tiny
    for i, f in enumerate(out["_feats"]):
        assert_close(f, z[f"feat/{i}"], FP32_TOL, f"feat{i}")
    for i, f in enumerate(out["_fused"]):
        assert_close(f, z[f"fused/{i}"], FP32_TOL, f"fused{i}")
    assert_close(out["seg"], z["out/seg"], FP32_TOL, "seg")
    assert np.array_equal(out["detection"]["anchors"].numpy(), z["out/anchors"])
    assert_close(out["detection"]["regression"], z["out/regression"], FP32_TOL, "regression")
    assert_close(out["detection"]["classification"], z["out/classification"], FP32_TOL, "classification")
    assert_close(out["lane"]["predict_cls"], z["out/lane_cls"], FP32_TOL, "lane_cls")
    assert_close(out["lane"]["predict_loc"], z["out/lane_loc"], FP32_TOL, "lane_loc")


def test_tiny_losses(tiny):
    z, cfgs, sd, batch, out, ld, tot = tiny
    for k, v in ld.items():
        assert_close(v, z["loss/" + k], 1e-5, k)
    assert_close(tot, z["loss/total"], 1e-5, "total")


def test_tiny_gradients_and_unused_params(tiny):
    z, cfgs, sd, batch, out, ld, tot = tiny
    nograd = set(z["meta/nograd"].tolist())
    five = len(O.regnet_stages(**{k: cfgs["backbone"][k] for k in ("initial_width", "slope", "quantized_param", "network_depth",
                                                                   "bottleneck_ratio", "group_width")})[0]) == 5
    # 5 stages: the last stage IS P6 and p5_to_p6 never runs; 4 stages: P6 is pooled from p5_to_p6(P5) and every parameter is used
    assert nograd == ({"neck.bifpn.0.p5_to_p6.0.conv.weight", "neck.bifpn.0.p5_to_p6.0.conv.bias",
                       "neck.bifpn.0.p5_to_p6.1.weight", "neck.bifpn.0.p5_to_p6.1.bias"} if five else set())
    checked = 0
    for k in z.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        g = sd[name].grad
        assert g is not None, name
        assert_close(g, z[k], 2e-4, "grad " + name, atol=1e-6)
        checked += 1
    assert checked > 300
    for name in nograd:
        assert sd[name].grad is None


def test_tiny_running_stats(tiny):
    z, cfgs, sd, *_ = tiny
    for k in z.files:
        if k.startswith("sd_after/"):
            name = k[9:]
            if name.startswith("neck.bifpn.0.p5_to_p6") and "neck.bifpn.0.p6_down_channel.1.weight" in sd:
                continue                                             # unused in the 5-input path
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(z[k]), name
            else:
                assert_close(sd[name], z[k], 1e-5, name)


@pytest.mark.parametrize("variant", ["tiny", "tiny4"])
def test_tiny_deploy_and_postprocess(variant):
    z = load_npz(VARIANTS[variant][0])
    cfgs = load_cfg(VARIANTS[variant][1])
    sd = tiny_state(z)
    for k in z.files:                                                # eval uses the post-training running stats
        if k.startswith("sd_after/"):
            sd[k[9:]] = torch.from_numpy(z[k].copy())
    x = torch.from_numpy(z["in/image"])
    with torch.no_grad():
        dep = O.hydranet_forward(sd, cfgs, x, training=False, mode="deploy")
    seg_ref = z["deploy/seg_argmax"]
    agree = float((dep[0].numpy() == seg_ref).mean())
    assert agree > 0.9999, agree                                     # fp32 reorder can flip exact ties only
    assert_close(dep[2], z["deploy/regression"], FP32_TOL, "dep reg")
    assert_close(dep[3], z["deploy/classification"], FP32_TOL, "dep cls")
    # A15: bit-exact bookkeeping GIVEN IDENTICAL fp inputs (the reference's own tensors)
    reg = torch.from_numpy(z["deploy/regression"])
    cls = torch.from_numpy(z["deploy/classification"])
    anc = torch.from_numpy(z["out/anchors"])
    anchors = torch.stack([anc[0]] * x.shape[0], 0)
    pp = O.postprocess((x.shape[2], x.shape[3]), anchors, reg, cls, float(z["deploy/pp_thresh"]), 0.3)
    for i, o in enumerate(pp):
        assert np.array_equal(np.asarray(o["rois"], np.float32), z[f"deploy/pp{i}/rois"])
        assert np.array_equal(np.asarray(o["class_ids"], np.int64), z[f"deploy/pp{i}/class_ids"])
        assert np.array_equal(np.asarray(o["scores"], np.float32), z[f"deploy/pp{i}/scores"])


def test_loss_kats():
    z = load_npz("loss_kats.npz")
    t = lambda k: torch.from_numpy(z[k])
    cw = [0.1, 0.5, 1.0, 5.0, 5.0]
    for name, kw in (("topk", dict(use_top_k=True, top_k_ratio=0.3, use_focal=False)),
                     ("plain", dict(use_top_k=False, top_k_ratio=1.0, use_focal=False)),
                     ("focal", dict(use_top_k=False, top_k_ratio=0.3, use_focal=True))):
        assert_close(O.seg_loss(t("seg/logits"), t("seg/gt_ones"), cw, **kw), z[f"seg/{name}/ones"], 1e-6, name)
        if name != "focal":
            assert_close(O.seg_loss(t("seg/logits"), t("seg/gt_rand"), cw, **kw), z[f"seg/{name}/rand"], 1e-6, name)
    for name, ann in (("rand", t("det/ann")), ("ones", torch.ones(3, 16, 5))):
        cl, rl = O.det_loss(t("det/cls"), t("det/reg"), t("det/anchors"), ann)
        assert_close(cl, z[f"det/{name}/cls_loss"], 1e-6, "det cls " + name)
        assert_close(rl, z[f"det/{name}/reg_loss"], 1e-6, "det reg " + name)
    assert_close(O.box_iou_anchor_gt(t("det/anchors")[0, ::97], t("det/ann")[0, :, :4]), z["det/iou"], 1e-6, "iou")
    hw, width = 400, 162
    ct = torch.ones(2, hw, 2)
    ct[:, 0:40, 1] = 0
    lt = torch.ones(2, hw, width)
    pos, neg, pmask, pnum = O.lane_cls_loss(ct, t("lane/cls_pred"))
    assert_close(pos, z["lane/smoke/pos"], 1e-6)
    assert_close(neg, z["lane/smoke/neg"], 1e-6)
    assert int(pnum) == int(z["lane/smoke/pnum"])
    assert_close(O.lane_loc_loss(pmask, pnum, lt, t("lane/loc_pred")), z["lane/smoke/loc_default160"], 1e-6)
    assert_close(O.lane_loc_loss(pmask, pnum, lt, t("lane/loc_pred"), points_per_line=80), z["lane/smoke/loc_ppl80"], 1e-6)
    cfgs = load_cfg("hydranet_big.yml")
    b = O.synthetic_batch(cfgs, 2, 640, 640, seed=5)
    pos, neg, pmask, pnum = O.lane_cls_loss(b["gt_cls"], t("lane/cls_pred"))
    assert_close(pos, z["lane/synth/pos"], 1e-6)
    assert_close(neg, z["lane/synth/neg"], 1e-6)
    assert_close(O.lane_loc_loss(pmask, pnum, b["gt_loc"], t("lane/loc_pred")), z["lane/synth/loc_default160"], 1e-6)
    allbg = torch.zeros(2, hw, 2)
    allbg[..., 0] = 1
    pos, neg, pmask, pnum = O.lane_cls_loss(allbg, t("lane/cls_pred"))
    assert float(pos) == float(z["lane/allbg/pos"]) == 0.0
    assert_close(neg, z["lane/allbg/neg"], 1e-6)
    assert int(pnum) == int(z["lane/allbg/pnum"]) == 1


def test_lane_loc_default_crashes_below_161_columns():
    """The reference's cal_loss_regress default points_per_line=160 indexes column 161."""
    lp = torch.zeros(1, 4, 130)
    with pytest.raises(IndexError):
        O.lane_loc_loss(torch.ones(4, dtype=torch.bool), torch.tensor(1), lp.clone(), lp)


def test_anchor_tables_exact():
    z = load_npz("loss_kats.npz")
    cfgs = load_cfg("hydranet_big.yml")
    for (h, w) in ((640, 640), (512, 1024)):
        a = O.anchors_for(h, w, cfgs)
        assert list(a.shape) == z[f"anchors/{h}x{w}/shape"].tolist()
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(z[f"anchors/{h}x{w}/sha256"])
        assert np.array_equal(a[::997], z[f"anchors/{h}x{w}/sample"])
    assert O.anchors_for(640, 640, cfgs).shape[0] == 76725
    assert O.anchors_for(512, 1024, cfgs).shape[0] == 98208


def test_regnet_width_derivation():
    z = load_npz("loss_kats.npz")
    for name in ("hydranet_joint_big_backbone", "hydranet_joint_big_backbone_interview", "hydranet_joint_small_backbone"):
        a = z[f"regnet/{name}/args"]
        w, d, g = O.regnet_stages(int(a[0]), int(a[1]), float(a[2]), int(a[3]), int(a[4]), int(a[5]))
        assert w == z[f"regnet/{name}/widths"].tolist()
        assert d == z[f"regnet/{name}/depths"].tolist()
        assert g == z[f"regnet/{name}/gw"].tolist()


def test_nms_known_answers():
    """Authored KATs for the restated torchvision NMS (parity unpinned: torchvision absent, unpinned in the reference)."""
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert O.nms_greedy(boxes, scores, 0.5).tolist() == [3, 2]        # 0 (IoU 1.0) and 1 (IoU 0.68) suppressed
    assert O.nms_greedy(boxes, scores, 0.7).tolist() == [3, 1, 2]     # IoU(3,1)=0.68 is not > 0.7
    # IoU exactly at the threshold is kept (strict >): two 2x1 boxes sharing half -> IoU = 1/3
    b = torch.tensor([[0, 0, 2, 1], [1, 0, 3, 1.0]])
    assert O.nms_greedy(b, torch.tensor([0.9, 0.8]), 1.0 / 3.0).tolist() == [0, 1]
    # classes never suppress each other
    assert O.batched_nms(boxes, scores, torch.tensor([0, 0, 0, 1]), 0.5).tolist() == [3, 0, 2]
    assert O.nms_greedy(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0


@pytest.mark.parametrize("fixture", ["big_keys.npz", "big_cond.npz"])
def test_big_cfg_numeric_digest(fixture):
    """(big_cond.npz: the same on the well-conditioned state of tests/helpers.conditioned_state -- the state of tests/test_fullsize3_gpu.py)
    SURVEY 8(c) item 2: the oracle on the big cfg at the repo-default 640x640 (B=1) against digests (mean, abs-max, L2) recorded from
    the reference itself: training-mode losses, head outputs and all 693 parameter-gradient norms; eval-mode feature maps, fused maps,
    head outputs and the arg-max class histogram.  Weights come from the seeded recipe both sides use (tests/helpers.synthetic_state,
    pinned by a sha256 of the generated state).  The live `points_per_line = 160` default is exercised here (162 location columns)."""
    from tests.helpers import conditioned_state, synthetic_state
    z = load_npz(fixture)
    cfgs = load_cfg("hydranet_big.yml")
    cfgs["dataloader"]["network_input_height"] = cfgs["dataloader"]["network_input_width"] = 640
    keys = z["keys"].tolist()
    shapes = [tuple(int(v) for v in s.split(",")) if s else () for s in z["shapes"].tolist()]
    sd = (conditioned_state if fixture == "big_cond.npz" else synthetic_state)(keys, shapes, seed=11)
    sha = hashlib.sha256(b"".join(np.ascontiguousarray(sd[k].numpy()).tobytes() for k in keys)).hexdigest()
    assert sha == str(z["digest/state_sha256"])
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    batch = O.synthetic_batch(cfgs, 1, 640, 640, seed=1)
    out = O.hydranet_forward(sd, cfgs, batch["image"], training=True)
    ld = O.hydranet_losses(cfgs, out, batch)                      # default lane_points_per_line = 160, as the reference runs it
    tot = O.total_loss(cfgs, ld)
    tot.backward()
    dig = lambda t: np.array([float(t.detach().double().mean()), float(t.detach().abs().max()), float(t.detach().double().norm())])
    for k, v in ld.items():
        assert_close(v, z["digest/loss/" + k], 2e-5, k)
    assert_close(tot, z["digest/loss/total"], 2e-5, "total")
    pairs = {"seg": out["seg"], "regression": out["detection"]["regression"], "classification": out["detection"]["classification"],
             "lane_cls": out["lane"]["predict_cls"], "lane_loc": out["lane"]["predict_loc"]}
    for k, t in pairs.items():
        got, ref = dig(t), z["digest/train/" + k]
        assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] and abs(got[2] - ref[2]) <= 1e-4 * ref[2], (k, got, ref)
        assert abs(got[0] - ref[0]) <= 1e-4 * ref[1], (k, got, ref)
    gk = z["digest/grad_keys"].tolist()
    gl = z["digest/grad_l2"]
    assert len(gk) == 693 and {k for k, v in sd.items() if v.requires_grad and v.grad is not None} == set(gk)
    scale = float(gl.max())
    for k, ref in zip(gk, gl):
        got = float(sd[k].grad.double().norm())
        assert abs(got - ref) <= 5e-4 * ref + 1e-7 * scale, (k, got, ref)
    # eval mode with the post-step running statistics
    esd = {k: v.detach() for k, v in sd.items()}
    with torch.no_grad():
        eo = O.hydranet_forward(esd, cfgs, batch["image"], training=False, want_features=True)
    ev = {f"feat{i}": f for i, f in enumerate(eo["_feats"])}
    ev.update({f"fused{i}": f for i, f in enumerate(eo["_fused"])})
    ev.update(regression=eo["detection"]["regression"], classification=eo["detection"]["classification"],
              lane_cls=eo["lane"]["predict_cls"], lane_loc=eo["lane"]["predict_loc"])
    for k, t in ev.items():
        got, ref = dig(t), z["digest/eval/" + k]
        assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] and abs(got[2] - ref[2]) <= 1e-4 * ref[2], (k, got, ref)
    hist = torch.bincount(torch.argmax(eo["seg"], 1).flatten(), minlength=5).numpy()
    assert np.abs(hist - z["digest/eval/seg_argmax_hist"]).sum() <= 4          # a handful of near-tie pixels may flip between graph orders


@pytest.mark.parametrize("tag", ["512x1024", "640x640", "128x256"])
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_lane_decode_and_nms_vs_reference_codec(tag, case):
    """oracle lane decode + lane NMS against the reference's own LaneHeader.decode / LaneCodec.decode_lane / nms_with_pos outputs
    (tests/golden/lane_decode.npz): kept lanes, their order, start/end positions and point counts are exact (INT bookkeeping);
    probabilities and point coordinates to float32 precision."""
    z = load_npz("lane_decode.npz")
    w, h, stride, ppl = (int(v) for v in z[f"{tag}/geom"])
    thr, nms_thr, use_mean = z[f"{tag}/{case}/params"]
    geo = O.LaneGeometry(w, h, stride, ppl)
    cand = O.lane_decode(geo, z[f"{tag}/cls"], z[f"{tag}/loc"], float(thr))
    assert len(cand) == int(z[f"{tag}/{case}/n_candidates"])
    lanes = O.lane_nms(cand, float(nms_thr), bool(use_mean))
    k = f"{tag}/{case}"
    assert [l["start_pos"] for l in lanes] == z[k + "/start_pos"].tolist()
    assert [l["end_pos"] for l in lanes] == z[k + "/end_pos"].tolist()
    assert [len(l["xs"]) for l in lanes] == z[k + "/npts"].tolist()
    np.testing.assert_allclose([l["prob"] for l in lanes], z[k + "/prob"], rtol=1e-6)
    np.testing.assert_allclose([l["ax"] for l in lanes], z[k + "/ax"], rtol=0, atol=0)
    np.testing.assert_allclose([l["ay"] for l in lanes], z[k + "/ay"], rtol=0, atol=0)
    np.testing.assert_allclose(np.concatenate([l["xs"] for l in lanes]), z[k + "/xs"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(np.concatenate([l["ys"] for l in lanes]), z[k + "/ys"], rtol=0, atol=1e-9)


def test_preprocess_and_iou_vs_reference():
    """(f3) oracle pre-processing at the network size against the reference's imagenet_normalize output (bit-exact: float64 arithmetic, one
    final cast); (f4) oracle IoU statistics / scores against head_seg/seg_metrics.py, incl. ignore pixels, ignore_index and absent classes."""
    z = load_npz("aux_stages.npz")
    frame = z["pre/frame_bgr"]
    got = O.preprocess_bgr(frame, frame.shape[:2])
    assert got.dtype == np.float32 and np.array_equal(got, z["pre/expected"])
    # the restated cv2 fixed-point resize: identity at equal size, exact on a constant image, within 1 grey level of float bilinear
    const = np.full((10, 14, 3), 77, np.uint8)
    assert np.array_equal(O.resize_bilinear_u8(const, (25, 31)), np.full((25, 31, 3), 77, np.uint8))
    up = O.resize_bilinear_u8(frame, (48, 80)).astype(np.float64)
    ref = torch.nn.functional.interpolate(torch.from_numpy(frame.astype(np.float64)).permute(2, 0, 1)[None], size=(48, 80), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(up - ref).max() <= 1.0
    pred, tgt = torch.from_numpy(z["iou/pred"]), torch.from_numpy(z["iou/target"])
    tp, fp, fn, sup = O.seg_stat_scores(pred, tgt, 5)
    for name, v in (("tp", tp), ("fp", fp), ("fn", fn), ("sup", sup)):
        assert np.array_equal(v.numpy().astype(np.float32), z["iou/" + name]), name
    for name, nc, kw in (("plain", 5, {}), ("ignore0", 5, dict(ignore_index=0)), ("absent", 7, dict(absent_score=1.0))):
        a = O.seg_stat_scores(pred, tgt, nc)
        b = O.seg_stat_scores(tgt.clamp_max(4), tgt, nc)
        sc = O.seg_iou_scores(*(x + y for x, y in zip(a, b)), **kw)
        np.testing.assert_allclose(sc.numpy(), z[f"iou/{name}/scores"], rtol=1e-6)


def test_lane_metric_vs_reference_recording():
    """(f4) lane F1: the oracle's restatement of head_lane/lane_metric.py (spline_interp / calc_params, calc_iou, evaluate_core) against the
    recording made by the reference's own functions (tests/golden/lane_metric.json; cv2.line = the same restated rasteriser on both sides)."""
    import json
    import os
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lane_metric.json")))
    for s in z["splines"]:
        ip = O.lane_spline_interp(s["lane"], 1)
        assert len(ip) == len(s["x"])
        np.testing.assert_allclose([p["x"] for p in ip], s["x"], rtol=0, atol=1e-9)
        np.testing.assert_allclose([p["y"] for p in ip], s["y"], rtol=0, atol=1e-9)
    H, W = z["H"], z["W"]
    g0 = z["images"][0]["gt_result"]["Lines"]
    p0 = [l["points"] for l in z["images"][0]["pr_result"]["Lines"]]
    iou = [[O.lane_iou(g, p, H, W, 30) for p in p0] for g in g0]
    np.testing.assert_allclose(iou, z["iou_image0"], rtol=0, atol=1e-12)
    for key, rec in z["results"].items():
        lw, thr = key.split(",")
        recs = []
        for im in z["images"]:
            gts = [l for l in im["gt_result"]["Lines"] if len(l) > 0]
            prs = [l["points"] for l in im["pr_result"]["Lines"] if l["score"] > float(thr)]
            recs.append(O.lane_evaluate(gts, prs, H, W, 0.5, int(lw)))
        assert recs == rec["records"], key
