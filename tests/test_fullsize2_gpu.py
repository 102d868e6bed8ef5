"""Full-size (big cfg) parity, part 2: the configurations and segments VERDICT r01 listed as unexercised.

  * teacher-forced deep backbone stages 2-4, all three BiFPN cells and the lane head at 3x512x1024 against the oracle (bf16-mirror mode,
    its own torch code executed on the device) -- same tolerances as tests/test_fullsize_gpu.py;
  * BASELINE config 3 as ONE training step (N = 16, 3x512x1024): the six losses against the oracle-on-device, run-to-run determinism;
  * BASELINE config 2 (backbone only, N = 8): features / loss against the oracle-on-device;
  * the repo-default 640x640 step (live `points_per_line = 160` columns, pyramid levels that are not multiples of 128 rows);
  * eval-mode END-TO-END against the UNMIRRORED fp32 oracle: with running-statistics BatchNorm the network is a fixed function, so
    there is no batch-statistics feedback to amplify bf16 rounding; every feature map, fused map, logits, regression, classification
    and lane output is held to the stated tolerances (EVAL_* below) and the arg-max mask agreement is reported.
Every run writes its measured errors to gpurun_out/fullsize2_*.json.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import ROOT, load_cfg
from tests.test_fullsize_gpu import ACT_TOL, GRAD_TOL, H, W, big, check_param_grads, nchw, nhwc, oracle_state, rel  # noqa: F401

pytestmark = pytest.mark.gpu

# eval-mode end-to-end tolerances (max|err| / max|ref| per tensor).  bf16 STORAGE alone moves the deep tensors of this untrained 30-block
# network by 25 % (stage 3) to 85 % (stage 4) in max-norm even in eval mode: the oracle in bf16-mirror mode (an independent torch
# implementation that rounds to bf16 where the HIP path stores bf16) differs from its own fp32 run by that much, on CPU and on the device
# alike, and two bf16 implementations decorrelate the same way (one flipped rounding is amplified block by block).
# So the end-to-end statement is three-way: with gap = (mirror oracle vs fp32 oracle), HIP vs the fp32 oracle <= EVAL_GAP_FACTOR x gap +
# EVAL_TOL_ABS and HIP vs the mirror oracle <= EVAL_GAP_FACTOR_PAIR x gap + EVAL_TOL_ABS; the shallow stages, where the gap is small, to an
# absolute EVAL_TOL_SHALLOW (max-norm).  The gap-relative statements are made in RELATIVE L2: measured, HIP / mirror = 0.95 ... 1.00 for
# every tensor in L2, while the max-norm ratio of two chaotic bf16 realisations scatters between 0.70 and 1.24 (it is decided by a single
# element) and crossed any fixed factor whenever a tile shape changed a summation order; the max-norm numbers are still recorded.
# Kernel-level correctness of the deep stages is established block by block (test_fullsize_backbone_deep_stage), and the end-to-end
# eval comparison against the REFERENCE's recorded fp32 outputs is tests/test_model_gpu.py (tiny cfg: 4e-2, measured 5e-3).
EVAL_GAP_FACTOR = 1.25
EVAL_GAP_FACTOR_PAIR = 1.6               # two decorrelated bf16 realisations differ by up to sqrt(2) x their own distance from fp32
EVAL_TOL_ABS = 2e-2
EVAL_TOL_SHALLOW = 2e-2                  # feat0 / feat1 vs the fp32 oracle


def dump(name, obj):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(obj, open(os.path.join(ROOT, "gpurun_out", f"fullsize2_{name}.json"), "w"), indent=1)
    print(name, obj)


def gen(seed):
    return torch.Generator(device="cuda:0").manual_seed(seed)


def cos_l2(a, b):
    """(cosine similarity, relative L2 error).  Used for INPUT gradients of residual blocks: d(out)/d(x) carries the ReLU mask of the block
    output element-wise, and a pre-activation within one bf16 ulp of zero flips that mask between two bf16 implementations -- a max-norm
    comparison then reports |upstream gradient| at a handful of elements, not a kernel error (the outputs themselves agree to 5e-3)."""
    a, b = a.detach().float().flatten(), b.detach().float().flatten()
    return float(F.cosine_similarity(a, b, dim=0)), float((a - b).norm() / b.norm().clamp(min=1e-20))


DIN_COS, DIN_L2 = 0.99, 0.12          # residual-block input gradients (see cos_l2)
PARAM_MAX = 0.2                       # parameter gradients: cosine >= 0.995 AND max-norm error <= PARAM_MAX (mask flips enter here too)


def param_grad_report(net, sd, prefix):
    """(worst cosine, worst scaled max error, list of offenders) over the parameters under `prefix`"""
    worst_cos, worst_err, bad, n = 1.0, 0.0, [], 0
    scale = max(float(v.grad.abs().max()) for k, v in sd.items() if k.startswith(prefix) and v.grad is not None)
    for name, prm in net.named_parameters():
        if not name.startswith(prefix):
            continue
        ref = sd[name].grad
        assert (prm.grad is None) == (ref is None), name
        if ref is None:
            continue
        n += 1
        g = prm.grad.float()
        if float(ref.abs().max()) < 1e-5 * scale:               # bias in front of BatchNorm: mathematically zero
            if float(g.abs().max()) >= 1e-4 * scale:
                bad.append((name, "nonzero", float(g.abs().max())))
            continue
        cos = float(F.cosine_similarity(g.flatten(), ref.flatten(), dim=0)) if g.numel() > 1 else 1.0
        e = rel(g, ref)
        worst_cos, worst_err = min(worst_cos, cos), max(worst_err, e)
        if cos < 0.995 or e > PARAM_MAX:
            bad.append((name, cos, e))
    assert n > 0
    return worst_cos, worst_err, bad


@pytest.mark.parametrize("stage", [2, 3, 4])
def test_fullsize_backbone_deep_stage(big, stage):
    """every XBlock of stage k (4 / 10 / 14 blocks at 32x64 / 16x32 / 8x16, 152 / 376 / 936 channels), N = 8, teacher-forced BLOCK BY
    BLOCK: block i gets the HIP output of block i-1 as its input on both sides and a fresh random upstream gradient, so each block's
    kernels are checked at full size without the chaotic amplification of a 14-block training-mode chain."""
    _deep_stage(big, stage, H, W, 8, f"stage{stage}")


@pytest.mark.parametrize("stage", [2, 3, 4])
def test_default_resolution_backbone_deep_stage(big, stage):
    """the same at the reference's default 640 x 640 input (hydranet_joint_big_backbone.yml:29-30), N = 16: stages 2-4 are 40 x 40 /
    20 x 20 / 10 x 10 maps -- 1600 / 400 / 100 pixels per image, no multiple of 128: SE partial rows, per-image gate rows and the GEMM /
    conv statistics epilogues all meet image boundaries inside their tiles.  Every block must take the fused one-node path."""
    from multitask_hydranet_amd import ops as K
    calls = []
    orig = K.XBlockFn.apply
    try:
        K.XBlockFn.apply = staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        _deep_stage(big, stage, 640, 640, 16, f"stage{stage}_640")
    finally:
        del K.XBlockFn.apply                       # back to torch.autograd.Function's own classmethod
    assert len(calls) == big[0].depths[stage], (len(calls), big[0].depths[stage])


def _deep_stage(big, stage, H, W, n, tag):
    net, cfgs, O = big
    p = "backbone.net."
    b = cfgs["backbone"]
    widths, depths, gws = O.regnet_stages(b["initial_width"], b["slope"], b["quantized_param"], b["network_depth"], b["bottleneck_ratio"],
                                          b["group_width"])
    s = stage + 1                                                            # stage k-1 output lives at stride 2^(k+1)
    x = torch.relu(torch.randn(n, widths[stage - 1], H >> s, W >> s, device="cuda:0", generator=gen(20 + stage))).to(torch.bfloat16).float()
    res, bad_all = {}, []
    cur = nhwc(x)
    for i in range(depths[stage]):
        q = f"{p}stage_{stage}.blocks.block_{i}"
        sd = oracle_state(net, q + ".")
        xin = nchw(cur).clone().requires_grad_(True)
        with O.bf16_mirror():
            t = O.xblock(sd, q, xin, b["stride"] if i == 0 else 1, widths[stage] // gws[stage], True)
        up = torch.randn(t.shape, device="cuda:0", generator=gen(1000 * stage + i))
        t.backward(up)
        net.zero_grad(set_to_none=True)
        a = cur.detach().clone().requires_grad_(True)
        o = net._xblock(q + ".", a, 2 if i == 0 else 1)
        o.backward(up.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
        wc, we, bad = param_grad_report(net, sd, q + ".")
        dc, dl = cos_l2(nchw(a.grad), xin.grad)
        res[f"block_{i}"] = dict(out=rel(nchw(o), t), din_cos=dc, din_rel_l2=dl, din_maxnorm=rel(nchw(a.grad), xin.grad),
                                 worst_param_cos=wc, worst_param_err=we)
        bad_all += bad
        cur = o.detach()
    dump(tag, res)
    for k, v in res.items():
        assert v["out"] <= ACT_TOL and v["din_cos"] >= DIN_COS and v["din_rel_l2"] <= DIN_L2, (k, v)
    assert not bad_all, bad_all


@pytest.mark.parametrize("cell", [0, 1, 2])
def test_fullsize_bifpn_cell(big, cell):
    """BiFPN cell k at full size (N = 4): cell 0 from four random backbone maps, cells 1-2 from five random 112-channel maps"""
    net, cfgs, O = big
    n = 4
    c = net.fpn_num_filters
    g = gen(40 + cell)
    if cell == 0:                             # the big cfg hands all five backbone maps to cell 0, which uses the last four
        ins = [torch.relu(torch.randn(n, net.widths[k], H >> (k + 2), W >> (k + 2), device="cuda:0", generator=g)) for k in (0, 1, 2, 3, 4)]
    else:
        ins = [torch.randn(n, c, H >> s, W >> s, device="cuda:0", generator=g) for s in (3, 4, 5, 6, 7)]
    ins = [t.to(torch.bfloat16).float() for t in ins]
    sd = oracle_state(net, f"neck.bifpn.{cell}.")
    rin = [t.clone().requires_grad_(True) for t in ins]
    with O.bf16_mirror():
        routs = O.bifpn_cell(sd, f"neck.bifpn.{cell}", rin, cell == 0, True)
    ups = [torch.randn(t.shape, device="cuda:0", generator=g) for t in routs]
    torch.autograd.backward(list(routs), ups)
    net.zero_grad(set_to_none=True)
    xin = [nhwc(t) for t in ins]
    outs = net._cell(f"neck.bifpn.{cell}.", xin, cell == 0)
    torch.autograd.backward(list(outs), [u.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16) for u in ups])
    res = {f"out{i}": rel(nchw(o), r) for i, (o, r) in enumerate(zip(outs, routs))}
    for i, (a, r) in enumerate(zip(xin, rin)):
        if r.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, i
        else:                                       # max-pool routing / Swish gradients: a flipped arg-max moves whole entries (see cos_l2)
            res[f"din{i}_cos"], res[f"din{i}_rel_l2"] = cos_l2(nchw(a.grad), r.grad)
    # parameter gradients; the fusion weights are a difference of nearly equal sums (see tests/test_model_gpu.py::check_params)
    worst, bad = 1.0, []
    gscale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for name, prm in net.named_parameters():
        if not name.startswith(f"neck.bifpn.{cell}.") or name.startswith("neck.bifpn.0.p5_to_p6"):
            continue
        ref = sd[name].grad
        gq = prm.grad.float()
        if float(ref.abs().max()) < 1e-5 * gscale:
            continue
        cos = float(F.cosine_similarity(gq.flatten(), ref.flatten(), dim=0)) if gq.numel() > 1 else 1.0
        fusion_w = name.split(".")[-1].startswith("p") and "_w" in name.split(".")[-1]
        worst = min(worst, cos)
        ok = (cos >= 0.98 and rel(gq, ref) <= 0.5) if fusion_w else (cos >= 0.995 and rel(gq, ref) <= PARAM_MAX)
        if not ok:
            bad.append((name, cos, rel(gq, ref)))
    res["worst_param_grad_cos"] = worst
    dump(f"bifpn{cell}", res)
    for k, v in res.items():
        if k.startswith("out"):
            assert v <= ACT_TOL, (k, v)
        elif k.endswith("_cos"):
            assert v >= DIN_COS, (k, v)
        elif k.endswith("_rel_l2"):
            assert v <= DIN_L2, (k, v)
    assert not bad, bad


def test_fullsize_lane_head(big):
    """lane head on four full-size fused maps, N = 16 (max-pool cascade + concat + three 448-channel branches)"""
    net, cfgs, O = big
    n = 16
    c = net.fpn_num_filters
    g = gen(50)
    ins = [torch.randn(n, c, H >> s, W >> s, device="cuda:0", generator=g).to(torch.bfloat16).float() for s in (3, 4, 5, 6)]
    sd = oracle_state(net, "laneheader.")
    rin = [t.clone().requires_grad_(True) for t in ins]
    with O.bf16_mirror():
        ref = O.lane_forward(sd, cfgs, rin + [None], True)
    wc = torch.randn(ref["predict_cls"].shape, device="cuda:0", generator=g)
    wl = torch.randn(ref["predict_loc"].shape, device="cuda:0", generator=g)
    ((ref["predict_cls"] * wc).sum() + (ref["predict_loc"] * wl).sum()).backward()
    net.zero_grad(set_to_none=True)
    xin = [nhwc(t) for t in ins]
    out = net._lane(xin + [None])
    ((out["predict_cls"] * wc).sum() + (out["predict_loc"] * wl).sum()).backward()
    res = dict(cls=rel(out["predict_cls"], ref["predict_cls"]), loc=rel(out["predict_loc"], ref["predict_loc"]))
    res.update({f"din{i}": rel(nchw(a.grad), r.grad) for i, (a, r) in enumerate(zip(xin, rin))})
    res["worst_param_grad_cos"] = check_param_grads(net, sd, "laneheader.")
    dump("lane", res)
    assert res["cls"] <= ACT_TOL and res["loc"] <= ACT_TOL
    assert all(res[f"din{i}"] <= GRAD_TOL for i in range(4))


def _oracle_step(net, cfgs, O, batch, ppl, training=True, mirror=True):
    sd = oracle_state(net, "")
    ctx = O.bf16_mirror() if mirror else torch.no_grad()
    with ctx:
        out = O.hydranet_forward(sd, cfgs, batch["image"], training=training, want_features=True)
        ld = O.hydranet_losses(cfgs, out, batch, lane_points_per_line=ppl) if training else None
    return sd, out, ld


# The tight statement about the loss path lives HERE, on the full-size batch (16 images, 98 208 anchors: nothing is normalised over a handful
# of samples).  Measured: seg 3e-5, det_cls 1.5e-3, det_reg 4.5e-3, lane_cls_pos 5.4e-3, lane_cls_neg 5.5e-3, lane_loc 3.5e-3, total 1.5e-3.
# The 2-image 128x128 fixtures of test_model_gpu.py move by +-0.6 % per change of an fp32 summation order and keep looser bounds (2.5e-2 / 6e-2).
# (loss_lane_cls_pos -- hard-example mining over the deepest features -- is the one term whose ORACLE value moves between runs of the
# torch-on-device mirror: 8.80 ... 8.98 against 9.03 here; it keeps 6e-2)
LOSS_TOL = dict(loss_seg=5e-3, loss_det_cls=1e-2, loss_det_reg=1e-2, loss_lane_cls_pos=6e-2, loss_lane_cls_neg=2e-2, loss_lane_loc=2e-2)


def test_fullsize_training_step_n16(big):
    """BASELINE config 3 as one step: N = 16, 3x512x1024, forward + multitask loss + backward.  Losses against the oracle-on-device
    (bf16-mirror), finite gradients for all 693 parameters, and a second identical step is bit-identical."""
    net, cfgs, O = big
    import bench
    batch = bench.synthetic_batch(cfgs, 16, H, W, seed=1, device="cuda:0")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    sd, rout, rld = _oracle_step(net, cfgs, O, batch, net.lane_points_per_line)
    runs = []
    for _ in range(2):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        out = net(batch["image"])
        ld = net.cal_loss(out, batch)
        tot = net.total_loss(ld)
        tot.backward()
        runs.append((float(tot), {k: float(v) for k, v in ld.items()}, [p.grad.clone() for p in net.parameters() if p.grad is not None]))
    net.load_state_dict(state)
    assert runs[0][0] == runs[1][0] and len(runs[0][2]) == 693
    assert all(torch.equal(a, b) and bool(torch.isfinite(a).all()) for a, b in zip(runs[0][2], runs[1][2]))
    res = {k: (runs[0][1][k], float(rld[k])) for k in rld}
    res["total"] = (runs[0][0], float(O.total_loss(cfgs, rld)))
    agree = float((torch.argmax(out["seg"], 1) == torch.argmax(rout["seg"], 1)).float().mean())
    res["seg_mask_agreement_vs_mirror_oracle"] = agree
    dump("step_n16", res)
    for k, tol in LOSS_TOL.items():
        a, b = res[k]
        assert abs(a - b) <= tol * abs(b), (k, a, b)
    assert abs(res["total"][0] - res["total"][1]) <= 1e-2 * abs(res["total"][1])


def test_fullsize_backbone_only_n8(big):
    """BASELINE config 2: backbone forward + backward at N = 8 (loss = sum of the five feature means, as bench.py --backbone-only)"""
    net, cfgs, O = big
    x = torch.randn(8, 3, H, W, device="cuda:0", generator=gen(60))
    state = {k: v.clone() for k, v in net.state_dict().items()}
    sd = oracle_state(net, "backbone.")
    with O.bf16_mirror():
        rf = O.backbone_forward(sd, cfgs, x, True)
    rloss = sum(f.mean() for f in rf)
    rloss.backward()
    net.zero_grad(set_to_none=True)
    feats = net._backbone(x)
    net._flush_nbt()
    loss = sum(f.float().mean() for f in feats)
    loss.backward()
    net.load_state_dict(state)
    res = {"loss": (float(loss), float(rloss))}
    res.update({f"feat{i}_maxnorm": rel(nchw(f), r) for i, (f, r) in enumerate(zip(feats, rf))})
    res.update({f"feat{i}_rel_l2": float((nchw(f) - r).norm() / r.norm()) for i, (f, r) in enumerate(zip(feats, rf))})
    dump("backbone_n8", res)
    assert abs(res["loss"][0] - res["loss"][1]) <= 1e-2 * abs(res["loss"][1])
    assert res["feat0_maxnorm"] <= ACT_TOL and res["feat1_maxnorm"] <= ACT_TOL
    assert all(res[f"feat{i}_rel_l2"] <= 5e-2 for i in range(3))             # stages 3-4 of a 30-block training-mode chain: chaotic
    # (reported above; their kernels are checked block by block in test_fullsize_backbone_deep_stage)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for n_, p in net.named_parameters() if n_.startswith("backbone."))


def test_repo_default_640x640_step():
    """repo default resolution (model/cfgs/hydranet_joint_big_backbone.yml:29-30): N = 4 at 3x640x640.  P7 is 5x5 = 25 rows per image, so
    the pyramid levels are not multiples of 128 rows; the location targets have 162 columns, so the reference's hard-coded
    points_per_line = 160 weights are live (columns 160 / 161)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import bench
    from multitask_hydranet_amd import HydraNet
    from oracle import hydranet_oracle as O
    cfgs = load_cfg("hydranet_big.yml")
    cfgs["dataloader"]["network_input_height"] = cfgs["dataloader"]["network_input_width"] = 640
    torch.manual_seed(0)
    net = HydraNet(cfgs).to("cuda:0").train()
    net.check_finite = False
    assert net.lane_points_per_line == 160
    batch = bench.synthetic_batch(cfgs, 4, 640, 640, seed=2, device="cuda:0")
    assert batch["gt_loc"].shape[-1] == 162
    sd, rout, rld = _oracle_step(net, cfgs, O, batch, 160)
    O.total_loss(cfgs, rld).backward()
    out = net(batch["image"])
    ld = net.cal_loss(out, batch)
    tot = net.total_loss(ld)
    tot.backward()
    res = {k: (float(ld[k]), float(rld[k])) for k in rld}
    res["regression_shape"] = list(out["detection"]["regression"].shape)
    dump("step_640", res)
    assert out["detection"]["regression"].shape == (4, 76725, 4) and out["lane"]["predict_loc"].shape == (4, 400, 162)
    for k, tol in LOSS_TOL.items():                   # N = 4: P6 / P7 BatchNorms normalise over 400 / 100 samples -> wider than at N = 16
        a, b = res[k]
        assert abs(a - b) <= max(tol, 6e-2) * abs(b), (k, a, b)
    n = 0
    for name, p in net.named_parameters():
        assert (p.grad is None) == (sd[name].grad is None), name
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), name
            n += 1
    assert n == 693
    # shallow-layer gradients are comparable element-wise (deep ones are covered by the teacher-forced segment tests)
    for name in ("segheader.decoder.8.conv.weight", "segheader.decoder.7.conv.conv.weight", "laneheader.conv_cls_conv.3.weight"):
        g, r = net._idx[name].grad.float(), sd[name].grad
        assert float(F.cosine_similarity(g.flatten(), r.flatten(), dim=0)) >= 0.97, name


def test_eval_end_to_end_vs_unmirrored_fp32_oracle(big):
    """north_star: "within a stated fp32 tolerance for feature maps, logits and loss".  Eval mode (running-statistics BatchNorm), big cfg,
    N = 2 at 3x512x1024: the HIP path against the UNMIRRORED fp32 oracle.  The running statistics are first set to the statistics of this
    very batch (one oracle training pass with momentum 1), so every layer sees normalised inputs as it would after training."""
    net, cfgs, O = big
    import bench
    batch = bench.synthetic_batch(cfgs, 2, H, W, seed=3, device="cuda:0")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    sd = {k: v.detach().clone().float() if v.is_floating_point() else v.detach().clone() for k, v in net.state_dict().items()}
    keep = (dict(O.BN_BACKBONE), dict(O.BN_NECK))
    try:
        O.BN_BACKBONE["momentum"] = 1.0
        O.BN_NECK["momentum"] = 1.0
        with torch.no_grad():
            O.hydranet_forward(sd, cfgs, batch["image"], training=True)
    finally:
        O.BN_BACKBONE.update(keep[0])
        O.BN_NECK.update(keep[1])
    try:
        net.load_state_dict(sd)
        net.eval()
        with torch.no_grad():
            ref = O.hydranet_forward(sd, cfgs, batch["image"], training=False, want_features=True)
            with O.bf16_mirror():
                mir = O.hydranet_forward(sd, cfgs, batch["image"], training=False, want_features=True)
            feats = net._backbone(batch["image"])
            fused = net._neck(feats)
            out = net(batch["image"])
            dep = net(batch["image"], "deploy")

        def tensors(o):
            t = {f"feat{i}": f for i, f in enumerate(o["_feats"])}
            t.update({f"fused{i}": f for i, f in enumerate(o["_fused"])})
            t.update(seg=o["seg"], regression=o["detection"]["regression"], classification=o["detection"]["classification"],
                     lane_cls=o["lane"]["predict_cls"], lane_loc=o["lane"]["predict_loc"])
            return t
        mine = dict(out)
        mine["_feats"], mine["_fused"] = [nchw(f) for f in feats], [nchw(f) for f in fused]
        tm, tr, tmi = tensors(mine), tensors(ref), tensors(mir)
        l2 = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-20))
        res = {k: dict(hip_vs_fp32=rel(tm[k], tr[k]), hip_vs_mirror=rel(tm[k], tmi[k]), mirror_vs_fp32=rel(tmi[k], tr[k]),
                       l2_hip_vs_fp32=l2(tm[k], tr[k]), l2_hip_vs_mirror=l2(tm[k], tmi[k]), l2_mirror_vs_fp32=l2(tmi[k], tr[k])) for k in tr}
        res["seg_mask_agreement"] = dict(hip_vs_fp32=float((dep[0] == torch.argmax(ref["seg"], 1)).float().mean()),
                                         hip_vs_mirror=float((dep[0] == torch.argmax(mir["seg"], 1)).float().mean()),
                                         mirror_vs_fp32=float((torch.argmax(mir["seg"], 1) == torch.argmax(ref["seg"], 1)).float().mean()))
        dump("eval_end_to_end", res)
    finally:
        net.train()
        net.load_state_dict(state)
    for k, v in res.items():
        if k == "seg_mask_agreement":
            continue
        gap = v["l2_mirror_vs_fp32"]
        assert v["l2_hip_vs_fp32"] <= EVAL_GAP_FACTOR * gap + EVAL_TOL_ABS, (k, v)
        assert v["l2_hip_vs_mirror"] <= EVAL_GAP_FACTOR_PAIR * gap + EVAL_TOL_ABS, (k, v)
    assert res["feat0"]["hip_vs_fp32"] <= EVAL_TOL_SHALLOW and res["feat1"]["hip_vs_fp32"] <= EVAL_TOL_SHALLOW
    assert res["seg_mask_agreement"]["hip_vs_fp32"] >= res["seg_mask_agreement"]["mirror_vs_fp32"] - 0.05     # (two chaotic realisations: 0.80 +- 0.02)
    assert torch.equal(dep[0], torch.argmax(out["seg"], 1))                   # the HIP arg-max is bit-exact on the HIP logits
