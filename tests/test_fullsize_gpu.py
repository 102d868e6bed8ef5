"""Parity at BASELINE.json's full sizes (big cfg, 3x512x1024).  The CPU oracle needs seconds per image at this size, so here the
ORACLE's own torch code (oracle/hydranet_oracle.py, bf16-mirror mode, fp32) is executed on the GPU device as the checker for teacher-forced
segments of the network, the loss kernels are checked against the oracle's loss functions on full-size random tensors, and the whole
step is checked for the size-independent properties the domain offers: run-to-run bit determinism (no float atomics anywhere) and
finite, identical gradients between the eager step and its hipGraph replay.
Tolerances as in test_model_gpu.py (activations 3e-2 * max|ref|, gradients cosine >= 0.995 and 6e-2; fp32 loss kernels 2e-5)."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import load_cfg

pytestmark = pytest.mark.gpu
H, W = 512, 1024
ACT_TOL, GRAD_TOL, GRAD_COS = 3e-2, 6e-2, 0.995


def rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-20))


@pytest.fixture(scope="module")
def big():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import __graft_entry__ as g
    g.build()
    from multitask_hydranet_amd import HydraNet
    from oracle import hydranet_oracle as O
    cfgs = load_cfg("hydranet_big.yml")
    cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = H, W
    torch.manual_seed(0)
    net = HydraNet(cfgs).to("cuda:0").train()
    net.check_finite = False
    net.lane_points_per_line = H // cfgs["lane"]["interval"]
    return net, cfgs, O


def oracle_state(net, prefix):
    sd = {k: (v.detach().clone().float() if v.is_floating_point() else v.detach().clone()) for k, v in net.state_dict().items()
          if k.startswith(prefix)}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return sd


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).requires_grad_(True)


def nchw(t):
    return t.detach().float().permute(0, 3, 1, 2)


def check_param_grads(net, sd, prefix):
    worst, n = 1.0, 0
    scale = max(float(v.grad.abs().max()) for k, v in sd.items() if k.startswith(prefix) and v.grad is not None)
    for name, p in net.named_parameters():
        if not name.startswith(prefix):
            continue
        ref = sd[name].grad
        assert (p.grad is None) == (ref is None), name
        if ref is None:
            continue
        n += 1
        g = p.grad.float()
        if float(ref.abs().max()) < 1e-5 * scale:               # bias in front of BatchNorm: mathematically zero (fp32 noise in the oracle)
            assert float(g.abs().max()) < 1e-4 * scale, name
            continue
        cos = float(F.cosine_similarity(g.flatten(), ref.flatten(), dim=0)) if g.numel() > 1 else 1.0
        worst = min(worst, cos)
        assert cos >= GRAD_COS and rel(g, ref) <= GRAD_TOL, (name, cos, rel(g, ref))
    assert n > 0
    return worst


def test_fullsize_backbone_high_resolution_stages(big):
    """stem + stage_0 + stage_1 at 3x512x1024 (the layers with 2M / 524k / 131k pixel rows per image), N = 2"""
    net, cfgs, O = big
    n = 2
    p = "backbone.net."
    x = torch.randn(n, 3, H, W, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    sd = oracle_state(net, p)
    b = cfgs["backbone"]
    widths, depths, gws = O.regnet_stages(b["initial_width"], b["slope"], b["quantized_param"], b["network_depth"], b["bottleneck_ratio"],
                                          b["group_width"])
    with O.bf16_mirror():
        t = O._r(F.conv2d(x, sd[p + "stem.conv.weight"], None, 2, 1))
        t = O._r(F.relu(O._bn(sd, p + "stem.bn", t, True, **O.BN_BACKBONE)))
        for k in (0, 1):
            for i in range(depths[k]):
                t = O.xblock(sd, f"{p}stage_{k}.blocks.block_{i}", t, b["stride"] if i == 0 else 1, widths[k] // gws[k], True)
    up = torch.randn(t.shape, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))
    t.backward(up)
    net.zero_grad(set_to_none=True)
    o = net._cba(x, p + "stem.conv", p + "stem.bn", dict(eps=1e-5, momentum=0.1), kind="stem", act=1)
    for k in (0, 1):
        for i in range(depths[k]):
            o = net._xblock(f"{p}stage_{k}.blocks.block_{i}.", o, 2 if i == 0 else 1)
    o.backward(up.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
    assert rel(nchw(o), t) <= ACT_TOL
    for pre in (p + "stem.", p + "stage_0.", p + "stage_1."):
        check_param_grads(net, sd, pre)


def test_fullsize_seg_head(big):
    """the whole decoder at full size (N = 2): direct 3x3 convs, ELU-folded gradients, 4-phase output conv at 512x1024"""
    net, cfgs, O = big
    n = 2
    c = net.fpn_num_filters
    g = torch.Generator(device="cuda:0").manual_seed(3)
    feats = [torch.randn(n, net.widths[0], H // 4, W // 4, device="cuda:0", generator=g)] + \
            [torch.randn(n, c, H >> s, W >> s, device="cuda:0", generator=g) for s in (3, 4, 5)]
    feats = [f.to(torch.bfloat16).float() for f in feats]
    sd = oracle_state(net, "segheader.")
    ins = [f.clone().requires_grad_(True) for f in feats]
    with O.bf16_mirror():
        ref = O.seg_forward(sd, ins)
    up = torch.randn(ref.shape, device="cuda:0", generator=g)
    ref.backward(up)
    net.zero_grad(set_to_none=True)
    xin = [nhwc(f) for f in feats]
    out = net._seg(xin)
    out.backward(up)
    assert rel(out, ref) <= ACT_TOL
    for a, r in zip(xin, ins):
        assert rel(nchw(a.grad), r.grad) <= GRAD_TOL
    check_param_grads(net, sd, "segheader.")
    # bit-exact argmax bookkeeping on identical fp inputs at full size
    from multitask_hydranet_amd import ops as K
    assert torch.equal(K.argmax_channels(out.detach()), torch.argmax(out.detach(), 1))


def test_fullsize_det_head_level_packed(big):
    """both towers on the five full-size pyramid levels with N = 16 (every level a multiple of 128 rows -> the level-packed path)"""
    net, cfgs, O = big
    n = 16
    c = net.fpn_num_filters
    g = torch.Generator(device="cuda:0").manual_seed(4)
    fused = [torch.randn(n, c, H >> s, W >> s, device="cuda:0", generator=g).to(torch.bfloat16).float() for s in (3, 4, 5, 6, 7)]
    img = torch.zeros(n, 3, H, W, device="cuda:0")
    sd = oracle_state(net, "detectheader.")
    ins = [f.clone().requires_grad_(True) for f in fused]
    with O.bf16_mirror():
        anchors_ref, reg_ref, cls_ref = O.det_forward(sd, cfgs, img, ins, True)
    wr = torch.randn(reg_ref.shape, device="cuda:0", generator=g)
    wc = torch.randn(cls_ref.shape, device="cuda:0", generator=g)
    ((reg_ref * wr).sum() + (cls_ref * wc).sum()).backward()
    net.zero_grad(set_to_none=True)
    from multitask_hydranet_amd import ops as K
    xin = [nhwc(f) for f in fused]
    assert net.pack_det_levels and K.levels_packable(xin)
    anchors, reg, cls = net._det(img, xin)
    ((reg * wr).sum() + (cls * wc).sum()).backward()
    assert torch.equal(anchors.cpu(), torch.as_tensor(anchors_ref).reshape(anchors.shape).float().cpu())
    assert rel(reg, reg_ref) <= ACT_TOL and rel(cls, cls_ref) <= ACT_TOL
    for a, r in zip(xin, ins):
        assert rel(nchw(a.grad), r.grad) <= GRAD_TOL
    check_param_grads(net, sd, "detectheader.")


def test_fullsize_loss_kernels_vs_oracle(big):
    """seg top-k CE (524 288 pixels per image), det focal/smooth-L1 over 98 208 anchors, lane OHEM + Huber: fp32 kernels vs the oracle"""
    net, cfgs, O = big
    import bench
    n = 2
    batch = bench.synthetic_batch(cfgs, n, H, W, seed=5, device="cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(6)
    a = net.anchors_for(H, W, torch.device("cuda:0"))
    A = a.shape[1]
    hw = (H // 32) * (W // 32)
    L = 2 * (H // cfgs["lane"]["interval"]) + 2
    raw = dict(seg=torch.randn(n, H, W, 5, device="cuda:0", generator=g).permute(0, 3, 1, 2) * 2,
               reg=torch.randn(n, A, 4, device="cuda:0", generator=g) * 0.3,
               cls=torch.sigmoid(torch.randn(n, A, 9, device="cuda:0", generator=g) * 2 - 3),
               lc=torch.randn(n, hw, 2, device="cuda:0", generator=g), ll=torch.randn(n, hw, L, device="cuda:0", generator=g))
    res = []
    for which in ("hip", "oracle"):
        t = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        pred = {"seg": t["seg"], "detection": {"anchors": a, "regression": t["reg"], "classification": t["cls"]},
                "lane": {"predict_cls": t["lc"], "predict_loc": t["ll"]}}
        if which == "hip":
            ld = net.cal_loss(pred, batch)
            tot = net.total_loss(ld)
        else:
            ld = O.hydranet_losses(cfgs, pred, batch, lane_points_per_line=net.lane_points_per_line)
            tot = O.total_loss(cfgs, ld)
        tot.backward()
        res.append(({k: v.detach().clone() for k, v in ld.items()}, {k: v.grad.clone() for k, v in t.items()}))
    for k in res[1][0]:
        assert rel(res[0][0][k], res[1][0][k]) <= 2e-5, (k, float(res[0][0][k]), float(res[1][0][k]))
    for k in res[1][1]:
        assert rel(res[0][1][k], res[1][1][k]) <= 1e-4, ("grad", k, rel(res[0][1][k], res[1][1][k]))


def test_fullsize_step_deterministic_and_graph_equals_eager(big):
    """size-independent properties of the whole step at 3x512x1024 (N = 4): two eager steps are bit-identical (deterministic reductions),
    and a hipGraph replay of the step reproduces the eager loss and every gradient bit for bit"""
    net, cfgs, O = big
    import bench
    n = 4
    batch = bench.synthetic_batch(cfgs, n, H, W, seed=7, device="cuda:0")
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def step():
        net.load_state_dict(state)                      # same running statistics / counters before every step
        net.zero_grad(set_to_none=True)
        loss = net.total_loss(net.cal_loss(net(batch["image"]), batch))
        loss.backward()
        return loss
    # every pre-capture step runs on the side stream that also hosts the capture warm-up, and no loss tensor outlives its step: a live
    # autograd graph keeps the parameters' AccumulateGrad nodes bound to the stream they were created on, and an accumulate on a
    # non-capturing stream during capture breaks the capture (segfault in hipStreamEndCapture on this stack)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    runs = []
    with torch.cuda.stream(s):
        for _ in range(2):
            lv = float(step().detach())
            runs.append((lv, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0] and runs[0][0] == runs[0][0]
    assert set(runs[0][1]) == set(runs[1][1]) and len(runs[0][1]) == 693
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
        assert bool(torch.isfinite(runs[0][1][k]).all()), k
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        gl = net.total_loss(net.cal_loss(net(batch["image"]), batch))
        gl.backward()
    ggrads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    for _ in range(2):
        net.load_state_dict(state)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert float(gl) == runs[0][0]
        for k in runs[0][1]:
            assert torch.equal(ggrads[k], runs[0][1][k]), k


def test_fullsize_eval_deploy_packed_equals_per_level(big):
    """eval-mode (running-statistics BatchNorm) deploy forward at N = 16: the level-packed det towers reproduce the per-level path, the
    6-tuple has the reference's shapes and the seg mask is the arg-max of the logits"""
    net, cfgs, O = big
    import bench
    batch = bench.synthetic_batch(cfgs, 16, H, W, seed=9, device="cuda:0")
    net.eval()
    try:
        outs = {}
        with torch.no_grad():
            for packed in (True, False):
                net.pack_det_levels = packed
                outs[packed] = net(batch["image"], "deploy")
    finally:
        net.pack_det_levels = True
        net.train()
    a, b = outs[True], outs[False]
    assert len(a) == 6 and a[0].dtype == torch.int64 and tuple(a[0].shape) == (16, H, W)
    assert torch.equal(a[0], b[0])                                   # seg masks identical (same kernels)
    assert torch.equal(a[1], b[1])                                   # anchors
    assert rel(a[2], b[2]) <= 1e-2 and rel(a[3], b[3]) <= 1e-2       # regression / classification: same arithmetic, eval-mode BN
    assert a[2].shape == (16, 98208, 4) and a[3].shape == (16, 98208, 9)
    assert a[4].shape == (16, (H // 32) * (W // 32), 2) and a[5].shape[2] == 2 * (H // 8) + 2
