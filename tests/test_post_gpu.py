"""GPU parity of the stages either side of the hot path: device detection post-process, lane decode + lane NMS,
input pre-processing, streaming segmentation IoU -- against the oracle restatements and the fixtures recorded from the reference itself.
INT bookkeeping (kept indices, order, classes, lane positions, counts) is exact; fp32 values to float32 precision."""
import numpy as np
import pytest
import torch

from tests.helpers import load_cfg, load_npz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import __graft_entry__ as g
    g.build()
    import multitask_hydranet_amd as P
    from oracle import hydranet_oracle as O
    return P, O


# ------------------------------------------------------------------------------------------------------------------------------------
# (f1) detection post-process
# ------------------------------------------------------------------------------------------------------------------------------------
def _check_det(mine, ref):
    assert len(mine) == len(ref)
    total = 0
    for i, (o, r) in enumerate(zip(mine, ref)):
        assert np.array_equal(np.asarray(o["class_ids"], np.int64), np.asarray(r["class_ids"], np.int64)), i
        assert np.array_equal(np.asarray(o["scores"], np.float32), np.asarray(r["scores"], np.float32)), i
        # box corners go through exp(): the device expf and the host libm differ by an ulp
        np.testing.assert_allclose(np.asarray(o["rois"], np.float32).reshape(-1, 4), np.asarray(r["rois"], np.float32).reshape(-1, 4),
                                   rtol=5e-6, atol=1e-5)
        total += len(o["class_ids"])
    return total


def test_det_postprocess_vs_reference_recording(pkg):
    """the reference's own recorded postprocess output (tiny fixture, DetectionHeader.decode): identical kept boxes / classes / scores"""
    P, O = pkg
    from multitask_hydranet_amd.postprocess import postprocess
    z = load_npz("tiny_hydranet.npz")
    reg, cls = torch.from_numpy(z["deploy/regression"]), torch.from_numpy(z["deploy/classification"])
    anc = torch.from_numpy(z["out/anchors"])
    hw = tuple(z["in/image"].shape[2:])
    mine = postprocess(hw, anc, reg, cls, float(z["deploy/pp_thresh"]), 0.3)
    ref = [dict(rois=z[f"deploy/pp{i}/rois"], class_ids=z[f"deploy/pp{i}/class_ids"], scores=z[f"deploy/pp{i}/scores"]) for i in range(len(mine))]
    assert _check_det(mine, ref) > 0


@pytest.mark.parametrize("n,thr,iou,spread", [(4, 0.30, 0.3, 1.0), (2, 0.35, 0.5, 0.5), (3, 0.9999, 0.3, 1.0), (16, 0.2, 0.4, 2.0)])
def test_det_postprocess_fullsize_vs_oracle(pkg, n, thr, iou, spread):
    """98 208 anchors x 9 classes at 512x1024: clustered detections (many overlapping boxes per object so the NMS has work), an empty
    image, score ties; kept indices / classes / scores identical to the oracle's host post-process, whole batch in one pipeline"""
    P, O = pkg
    cfgs = load_cfg("hydranet_big.yml")
    H, W = 512, 1024
    a = torch.from_numpy(O.anchors_for(H, W, cfgs))[None]
    A = a.shape[1]
    g = torch.Generator().manual_seed(int(thr * 1000) + n)
    reg = torch.randn(n, A, 4, generator=g) * 0.2 * spread
    cls = torch.sigmoid(torch.randn(n, A, 9, generator=g) * 1.5 - 4.0)
    hot = torch.randint(0, A, (n, 300), generator=g)
    for i in range(n):
        cls[i, hot[i], torch.randint(0, 9, (300,), generator=g)] = 0.5 + 0.5 * torch.rand(300, generator=g)
    cls[0, 100:110, 3] = 0.75                                     # exact score ties: order by anchor index
    if n > 2:
        cls[2] = cls[2] * 0.01                                    # an image with nothing over the threshold
    from multitask_hydranet_amd.postprocess import postprocess
    mine = postprocess((H, W), a.cuda(), reg.cuda(), cls.cuda(), thr, iou)
    ref = O.postprocess((H, W), torch.stack([a[0]] * n), reg, cls, thr, iou)
    total = _check_det(mine, ref)
    if n > 2:
        assert len(mine[2]["class_ids"]) == 0
    assert total > 0 or thr > 0.999


def test_deploy_forward_appends_device_detections(pkg):
    P, O = pkg
    from tests.helpers import tiny_state
    z = load_npz("tiny_hydranet.npz")
    net = P.HydraNet(load_cfg("hydranet_tiny.yml"))
    net.load_state_dict(tiny_state(z))
    net = net.cuda().eval()
    thr = float(z["deploy/pp_thresh"])
    net.deploy_postprocess = (thr, 0.3)
    with torch.no_grad():
        dep = net(torch.from_numpy(z["in/image"]).cuda(), "deploy")
    assert len(dep) == 7 and dep[6]["rois"].is_cuda and dep[6]["kept"].shape == (2,)
    det = dep[6]
    ref = O.postprocess(tuple(z["in/image"].shape[2:]), torch.stack([dep[1][0].cpu()] * 2), dep[2].cpu(), dep[3].cpu(), thr, 0.3)
    for i, r in enumerate(ref):
        k = int(det["kept"][i])
        assert k == len(r["class_ids"])
        assert np.array_equal(det["class_ids"][i, :k].cpu().numpy(), np.asarray(r["class_ids"], np.int64))


# ------------------------------------------------------------------------------------------------------------------------------------
# (f2) lane decode + lane NMS
# ------------------------------------------------------------------------------------------------------------------------------------
def _check_lanes(lanes, ref_prob, ref_start, ref_end, ref_npts, ref_xs, ref_ys, ref_ax=None):
    assert [l.start_pos for l in lanes] == list(ref_start)
    assert [l.end_pos for l in lanes] == list(ref_end)
    assert [len(l.lane) for l in lanes] == list(ref_npts)
    np.testing.assert_allclose([float(l.prob) for l in lanes], ref_prob, rtol=2e-6)
    if ref_ax is not None:
        np.testing.assert_allclose([l.ax for l in lanes], ref_ax, rtol=0, atol=0)
    np.testing.assert_allclose([float(p.x) for l in lanes for p in l.lane], ref_xs, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose([float(p.y) for l in lanes for p in l.lane], ref_ys, rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", ["512x1024", "640x640", "128x256"])
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_lane_decode_nms_vs_reference_codec(pkg, tag, case):
    """HIP lane decode + NMS against the outputs of the reference's own LaneHeader.decode (tests/golden/lane_decode.npz)"""
    P, O = pkg
    from multitask_hydranet_amd import lane_codec as LC
    z = load_npz("lane_decode.npz")
    w, h, stride, ppl = (int(v) for v in z[f"{tag}/geom"])
    thr, nms_thr, use_mean = z[f"{tag}/{case}/params"]
    codec = LC.LaneCodec(w, h, stride, ppl, do_interpolate=True, anchor_lane_num=1, scale_invariance=True)
    cls, loc = torch.from_numpy(z[f"{tag}/cls"]).cuda(), torch.from_numpy(z[f"{tag}/loc"]).cuda()
    lanes = LC.decode(cls, loc, codec, float(thr), float(nms_thr), bool(use_mean))
    k = f"{tag}/{case}"
    _check_lanes(lanes, z[k + "/prob"], z[k + "/start_pos"], z[k + "/end_pos"], z[k + "/npts"], z[k + "/xs"], z[k + "/ys"], z[k + "/ax"])
    cand = codec.decode_lane(torch.softmax(cls, -1), loc, float(thr))
    assert len(cand) == int(z[k + "/n_candidates"])
    d = LC.scale_to_org(lanes, w, h, 1920, 1080)                       # host bookkeeping runs on the device result
    assert len(d["Lines"]) == len(lanes)


@pytest.mark.parametrize("n,w,h,thr", [(16, 1024, 512, 0.5), (2, 1920, 1152, 0.5), (1, 2560, 2048, 0.9)])
def test_lane_decode_nms_batch_vs_oracle(pkg, n, w, h, thr):
    """a random batch at 512x1024 (512 anchors, most of them firing: hundreds of candidates per image), at the BASELINE config-5 deploy
    resolution 1152x1920 (2160 anchors: more than one per thread of the workgroup) and at 5120 anchors (> 64 KiB of LDS bookkeeping) against
    the oracle, one launch each"""
    P, O = pkg
    from multitask_hydranet_amd import lane_codec as LC
    stride, ppl = 32, h // 8
    g = torch.Generator().manual_seed(77)
    hw = (w // stride) * (h // stride)
    cls = torch.randn(n, hw, 2, generator=g) * 2
    loc = torch.randn(n, hw, 2 * ppl + 2, generator=g) * 2.0
    loc[:, :, ppl] = torch.rand(n, hw, generator=g) * ppl
    loc[:, :, ppl + 1] = torch.rand(n, hw, generator=g) * ppl
    codec = LC.LaneCodec(w, h, stride, ppl)
    geo = O.LaneGeometry(w, h, stride, ppl)
    res = LC.decode_batch(cls.cuda(), loc.cuda(), codec, thr, 30.0, False)
    for i in range(n):
        ref = O.lane_postprocess(geo, cls[i].numpy(), loc[i].numpy(), thr, 30.0, False)
        _check_lanes(res[i], [l["prob"] for l in ref], [l["start_pos"] for l in ref], [l["end_pos"] for l in ref], [len(l["xs"]) for l in ref],
                     np.concatenate([l["xs"] for l in ref]) if ref else [], np.concatenate([l["ys"] for l in ref]) if ref else [])
        assert len(ref) > 3


# ------------------------------------------------------------------------------------------------------------------------------------
# (f3) pre-processing, (f4) streaming IoU
# ------------------------------------------------------------------------------------------------------------------------------------
def test_preprocess_bgr(pkg):
    P, O = pkg
    from multitask_hydranet_amd.preprocess import preprocess_bgr
    z = load_npz("aux_stages.npz")
    frame = z["pre/frame_bgr"]
    out = preprocess_bgr(frame, frame.shape[:2])
    assert out.shape == (1, 3) + frame.shape[:2] and np.array_equal(out[0].cpu().numpy(), z["pre/expected"])     # reference recording, bit-exact
    rng = np.random.RandomState(3)
    frames = rng.randint(0, 256, size=(2, 1080, 1920, 3)).astype(np.uint8)
    for hw in ((512, 1024), (640, 640), (1152, 1920), (1080, 1920)):
        got = preprocess_bgr(frames, hw).cpu().numpy()
        for i in range(2):
            assert np.array_equal(got[i], O.preprocess_bgr(frames[i], hw)), hw          # vs the oracle's restated cv2 fixed-point resize


def test_streaming_iou(pkg):
    P, O = pkg
    from multitask_hydranet_amd.metrics import IntersectionOverUnion
    z = load_npz("aux_stages.npz")
    pred, tgt = torch.from_numpy(z["iou/pred"]), torch.from_numpy(z["iou/target"])
    for name, nc, kw in (("plain", 5, {}), ("ignore0", 5, dict(ignore_index=0)), ("absent", 7, dict(absent_score=1.0))):
        m = IntersectionOverUnion(nc, **kw)
        m.update(pred.cuda(), tgt.cuda().float())                                       # float class ids, as to_gpu delivers them
        m.update(tgt.clamp_max(4).cuda(), tgt.cuda())
        np.testing.assert_allclose(m.compute().cpu().numpy(), z[f"iou/{name}/scores"], rtol=1e-6)
    m = IntersectionOverUnion(5)
    m.update(pred.cuda(), tgt.cuda())
    tp, fp, fn, sup = m.stats()
    for name, v in (("tp", tp), ("fp", fp), ("fn", fn), ("sup", sup)):
        assert np.array_equal(v.cpu().numpy().astype(np.float32), z["iou/" + name]), name
    # full-size mask (16 x 512 x 1024 = 8.4 M pixels): exact against the oracle's bincount
    g = torch.Generator().manual_seed(1)
    p2, t2 = torch.randint(0, 5, (16, 512, 1024), generator=g), torch.randint(0, 6, (16, 512, 1024), generator=g)
    t2[t2 == 5] = 255
    m = IntersectionOverUnion(5)
    m.update(p2.cuda(), t2.cuda())
    for a, b in zip(m.stats(), O.seg_stat_scores(p2, t2, 5)):
        assert torch.equal(a.cpu(), b)


def test_det_postprocess_capacity_overflow_raises(pkg):
    """more than 32 768 anchors over the threshold in one image exceed the device NMS capacity: loud error, never a silent truncation"""
    P, O = pkg
    from multitask_hydranet_amd.postprocess import postprocess
    a = torch.rand(1, 40000, 4) * 100
    a[..., 2:] += a[..., :2] + 1
    with pytest.raises(RuntimeError, match="32768"):
        postprocess((512, 1024), a.cuda(), torch.zeros(1, 40000, 4).cuda(), torch.full((1, 40000, 9), 0.9).cuda(), 0.5, 0.5)


def test_dispatcher_ops_match_the_module_path(pkg):
    """torch.ops.hydranet_hip.* (multitask_hydranet_amd/torch_ops.py) run the same C-ABI calls as the nn.Module path: identical outputs,
    gradients and running statistics for conv1x1 + BN + ReLU; the top-k CE loss op reproduces ops.SegLoss incl. its gradient"""
    import multitask_hydranet_amd.torch_ops as T
    from multitask_hydranet_amd import ops as K
    g = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.randn(4, 16, 24, 64, device="cuda", generator=g).to(torch.bfloat16)
    w0 = torch.randn(152, 64, 1, 1, device="cuda", generator=g) * 0.1
    up = torch.randn(4, 16, 24, 152, device="cuda", generator=g).to(torch.bfloat16)
    res = []
    for which in ("module", "dispatcher"):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        gm, bt = torch.ones(152, device="cuda", requires_grad=True), torch.zeros(152, device="cuda", requires_grad=True)
        rm, rv = torch.zeros(152, device="cuda"), torch.ones(152, device="cuda")
        K.clear_pack_cache()
        if which == "module":
            out = K.conv_bn_act(x, w, None, (gm, bt, rm, rv, None), act=K.ACT_RELU)
        else:
            out = T.conv1x1_bn_act(x, w, gm, bt, rm, rv, K.ACT_RELU, 1e-5, 0.1, True)
        out.backward(up)
        res.append((out.detach().float(), x.grad.float(), w.grad, gm.grad, bt.grad, rm, rv))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-3 * max(float(a.abs().max()), 1e-6)
    logits = torch.randn(2, 32, 64, 5, device="cuda", generator=g, requires_grad=True)
    tgt = torch.randint(0, 5, (2, 32, 64), device="cuda", generator=g)
    cw = torch.tensor([0.1, 0.5, 1.0, 5.0, 5.0], device="cuda")
    l1 = K.SegLoss.apply(logits, tgt, cw, True, 0.3, 255)
    g1, = torch.autograd.grad(l1, logits)
    l2, _ = torch.ops.hydranet_hip.seg_topk_ce_fwd(logits, tgt, cw, True, 0.3, 255)
    g2, = torch.autograd.grad(l2, logits)
    assert float(l1) == float(l2) and torch.equal(g1, g2)
    m = torch.ops.hydranet_hip.argmax_channels(logits.detach())
    assert torch.equal(m, torch.argmax(logits.detach(), 3))


# ------------------------------------------------------------------------------------------------------------------------------------
# SegmentHeader.decode on the device + the demo loop (demo.py:167-261)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("net_hw,org_hw", [((128, 256), (270, 480)), ((128, 256), (128, 256)), ((512, 1024), (1080, 1920)), ((96, 160), (50, 90))])
def test_seg_decode_vs_oracle(pkg, net_hw, org_hw):
    """segheader.decode (head_seg/segmentation.py:107-125): arg-max -> colour table -> cv2.resize (effectively INTER_LINEAR: the reference
    passes INTER_NEAREST as `dst`) -> cv2.addWeighted(0.8, 0.5), against the oracle's restatement of the two cv2 calls, bit for bit
    (parity with cv2 itself is unpinned: cv2 is absent).  Up-scaling, identity and down-scaling; logits and int64 masks; a class without
    a colour entry stays black."""
    P, O = pkg
    from multitask_hydranet_amd.visual import seg_decode
    rs = np.random.RandomState(5)
    n, c = 2, 5
    logits = torch.from_numpy(rs.randn(n, c, *net_hw).astype(np.float32))
    # piecewise-constant regions as well as noise: edges are where the interpolation matters
    logits[:, 2, : net_hw[0] // 2] += 3.0
    logits[:, 4, :, net_hw[1] // 3:] += 2.0
    frames = [rs.randint(0, 256, size=org_hw + (3,)).astype(np.uint8) for _ in range(n)]
    colors = {0: (0, 0, 0), 1: (128, 0, 128), 2: (255, 255, 255), 4: (0, 255, 0)}          # class 3 has no entry
    ref = O.seg_decode(frames, logits, (org_hw[1], org_hw[0]), colors)
    got = seg_decode(frames, logits.cuda(), (org_hw[1], org_hw[0]), colors)
    assert len(got) == n
    for a, b in zip(got, ref):
        assert a.dtype == np.uint8 and a.shape == org_hw + (3,) and np.array_equal(a, b)
    got2 = seg_decode(frames, torch.argmax(logits, 1).cuda(), (org_hw[1], org_hw[0]), colors)      # the deploy forward's int64 mask
    assert all(np.array_equal(a, b) for a, b in zip(got2, ref))
    assert any((g != f).any() for g, f in zip(got, frames))


def test_demo_loop_end_to_end(pkg):
    """demo.py:167-261 as one chain on the device: BGR frame -> pre-processing -> eval forward -> laneheader.decode + scale_to_org,
    segheader.decode, detectheader.decode.  Tiny cfg with the reference's recorded weights; every stage is checked against the oracle on
    the HIP path's own intermediate tensors (the stages themselves are pinned elsewhere), and the module surface has no unavailable
    decode left for the seg head."""
    P, O = pkg
    from multitask_hydranet_amd.demo import Demo, synthetic_frames
    from multitask_hydranet_amd.preprocess import preprocess_bgr
    from tests.helpers import tiny_state
    z = load_npz("tiny_hydranet.npz")
    cfgs = load_cfg("hydranet_tiny.yml")
    demo = Demo(cfgs, fold_batchnorm=False)
    demo.net.load_state_dict(tiny_state(z))
    demo.net.eval().prepare_inference()
    demo.lane_conf, demo.det_conf = 0.3, 0.3                        # (thresholds that let the tiny random-ish model produce something)
    frames = synthetic_frames(2, 270, 480, seed=4)
    for frame in frames:
        r = demo.process(frame)
        assert r["visual"].shape == frame.shape and r["visual"].dtype == np.uint8 and r["org_size"] == (480, 270)
        img = preprocess_bgr(frame, (demo.net_h, demo.net_w))
        assert np.array_equal(img[0].cpu().numpy(), O.preprocess_bgr(frame, (demo.net_h, demo.net_w)))
        with torch.no_grad():
            out = demo.net(img)
        ref_vis = O.seg_decode([frame], out["seg"].float().cpu(), (480, 270), demo.colors)[0]
        assert np.array_equal(r["visual"], ref_vis)
        # lanes: the reference's json structure, in source-frame pixels
        assert isinstance(r["lanes"], list) and len(r["lanes"]) == 1
        for ln in r["lanes"][0]:
            assert set(ln) == {"score", "points"} and all(set(p) == {"x", "y"} for p in ln["points"])
        geo = O.LaneGeometry(demo.net_w, demo.net_h, cfgs["lane"]["anchor_stride"], int(demo.net_h / cfgs["lane"]["interval"]))
        ref_lanes = O.lane_postprocess(geo, out["lane"]["predict_cls"][0].float().cpu().numpy(), out["lane"]["predict_loc"][0].float().cpu().numpy(),
                                       demo.lane_conf, demo.lane_nms, False)
        assert len(ref_lanes) >= len(r["lanes"][0])                 # (scale_to_org drops lanes with fewer than two points)
        # detections: the reference's dict per image
        det = r["detections"]
        assert len(det) == 1 and set(det[0]) == {"rois", "class_ids", "scores"}
        ref_det = O.postprocess((demo.net_h, demo.net_w), out["detection"]["anchors"].float().cpu(), out["detection"]["regression"].float().cpu(),
                                out["detection"]["classification"].float().cpu(), demo.det_conf, demo.det_iou)
        _check_det(det, ref_det)
    assert "NotImplemented" not in repr(demo.net.segheader.decode)


# ------------------------------------------------------------------------------------------------------------------------------------
# (f4) lane F1 (head_lane/lane_metric.py)
# ------------------------------------------------------------------------------------------------------------------------------------
def test_lane_metric_device_vs_reference_recording(pkg):
    """LaneMetric on the device (hn_lane_raster + hn_lane_iou) against the recording made by the reference's own lane_metric.py
    (tests/golden/lane_metric.json): spline samples, the full IoU matrix of one image, per-image (gt, pr, hit) counts and the F1 summary for
    two lane widths x two score thresholds -- identical (integer pixel counts on both sides).  SELF-CONSISTENCY ONLY for the fill rule:
    cv2 is absent from this image, so the recording was made with cv2.line replaced by the oracle's restatement ("pixel centre within
    lane_width / 2 of the segment", tests/golden/make_golden.py:40-44); OpenCV's fixed-point thick-line fill differs at boundary pixels,
    and hit / miss decisions near IoU 0.5 are NOT pinned against real OpenCV.  What the recording does pin is everything around the
    rasteriser: the spline, the pairing, the Hungarian assignment, the thresholds and the F1 arithmetic of the reference's own code."""
    import json
    import os
    P, O = pkg
    from multitask_hydranet_amd import lane_metric as LM
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lane_metric.json")))
    for s in z["splines"]:
        ip = LM.spline_interp(lane=s["lane"], step_t=1)
        np.testing.assert_allclose([p["x"] for p in ip], s["x"], rtol=0, atol=1e-9)
        np.testing.assert_allclose([p["y"] for p in ip], s["y"], rtol=0, atol=1e-9)
    H, W = z["H"], z["W"]
    g0 = z["images"][0]["gt_result"]["Lines"]
    p0 = [l["points"] for l in z["images"][0]["pr_result"]["Lines"]]
    np.testing.assert_allclose(LM.iou_matrix(g0, p0, H, W, 30), np.array(z["iou_image0"]), rtol=0, atol=1e-12)
    for key, rec in z["results"].items():
        lw, thr = key.split(",")
        m = LM.LaneMetric(method="f1_measure", iou_thresh=0.5, lane_width=int(lw), thresh_list=[float(thr)])
        m.reset()
        m(output=z["images"])
        h = m.metric_handlers[0]
        assert h.result_record == rec["records"], key
        assert h.summary() == rec["summary"] and m.summary() == rec["f1"], key
    # full-size frame, many lanes: device counts == the oracle's numpy rasteriser
    rs = np.random.RandomState(2)
    gts = [[{"x": float(200 + 300 * j + 15 * i + 0.8 * i * i * (j - 2)), "y": float(1070 - 90 * i)} for i in range(11)] for j in range(5)]
    prs = [[{"x": p["x"] + float(rs.randint(-20, 20)), "y": p["y"]} for p in g] for g in gts[:4]] + [[{"x": 50.0, "y": 1000.0}, {"x": 1900.0, "y": 300.0}]]
    got = LM.iou_matrix(gts, prs, 1080, 1920, 30)
    ref = np.array([[O.lane_iou(g, p, 1080, 1920, 30) for p in prs] for g in gts])
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    assert got.max() > 0.5 and got.min() == 0.0
    # more lanes than the pair kernel's 32 x 32 block (the reference has no limit): 37 ground truths x 41 predictions in blocks
    # == the same matrix assembled from single-block calls
    many_g = [[{"x": float(40 + 45 * j + 3 * i), "y": float(500 - 40 * i)} for i in range(8)] for j in range(37)]
    many_p = [[{"x": float(30 + 42 * j + 4 * i), "y": float(500 - 40 * i)} for i in range(8)] for j in range(41)]
    big = LM.iou_matrix(many_g, many_p, 512, 2048, 30)
    assert big.shape == (37, 41)
    for g0_, p0_ in ((0, 0), (32, 0), (0, 32), (32, 32)):
        blk = LM.iou_matrix(many_g[g0_:g0_ + 32], many_p[p0_:p0_ + 32], 512, 2048, 30)
        np.testing.assert_array_equal(big[g0_:g0_ + 32, p0_:p0_ + 32], blk)
    assert big.max() > 0.3
