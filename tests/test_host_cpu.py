"""CPU: host logic of the product package (no kernel is launched here): the C-ABI library loads and exports every symbol
include/hydranet_hip.h declares, the module reproduces the reference's state_dict contract, and the static-shape losses equal
the oracle's per-image loops."""
import ctypes
import os
import sys
import time

import numpy as np
import pytest
import torch

from oracle import hydranet_oracle as O
from tests.helpers import ROOT, assert_close, load_cfg, load_npz


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from multitask_hydranet_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    sig = built.parse_header()
    assert len(sig) >= 40
    dll = ctypes.CDLL(built.SO_PATH)
    for name in sig:
        assert hasattr(dll, name), f"{name} declared in include/hydranet_hip.h but not exported"
    # planning helpers are pure host code and may run without a GPU
    l = built.lib()
    assert l.query("hn_nt_stat_rows", 1000, 64) == 8        # one partial statistic row per 128-row pixel tile
    s, r, w = ctypes.c_int(), ctypes.c_long(), ctypes.c_long()
    assert l.query("hn_wgrad_plan", 0, 16, 512, 1024, 16 * 512 * 1024, 64, 64, 1, ctypes.addressof(s), ctypes.addressof(r), ctypes.addressof(w)) == 0
    assert s.value >= 1 and r.value % 64 == 0 and s.value * r.value >= 16 * 512 * 1024
    assert w.value == s.value * 64 * 1 * 64 * 4 + s.value * 64 * 4        # fp32 slabs + the bias-gradient partial rows


def test_bad_arguments_are_rejected_without_a_gpu(built):
    l = built.lib()
    # null pointers / misaligned channel counts must come back as status 1 before any launch
    assert l.raw("hn_bn_act")(None, 8, None, None, None, 0, None, None, 0, None, 8, 4, 8, None) == 1
    assert l.raw("hn_dwconv_fwd")(1, 12, 1, 1, 12, 1, 4, 4, 12, None) == 1


def test_state_dict_contract_matches_reference(built):
    from multitask_hydranet_amd import HydraNet
    z = load_npz("big_keys.npz")
    net = HydraNet(load_cfg("hydranet_big.yml"))
    sd = net.state_dict()
    keys = list(sd.keys())
    assert keys == z["keys"].tolist()                                 # names AND order
    assert [",".join(map(str, v.shape)) for v in sd.values()] == z["shapes"].tolist()
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"]) == 42715747
    assert [k for k, _ in net.named_parameters()] == z["param_keys"].tolist()
    for attr in ("backbone", "neck", "segheader", "detectheader", "laneheader", "loss_seg", "loss_detect", "loss_cls", "loss_reg"):
        assert getattr(net, attr) is not None
    assert net.widths == [24, 64, 152, 376, 936] and net.depths == [1, 1, 4, 10, 14]
    # train.py:469-503 hands sub-module parameter lists to the optimizer
    assert len(list(net.laneheader.parameters())) == 15 and len(list(net.segheader.parameters())) == 18
    # static helpers the callers reach through the instance: train.py:334-336, demo.py:230-244, lanedetect.py:103-178
    import torch
    for head, names in ((net.detectheader, ("decode", "invert_affine", "display")), (net.segheader, ("decode",)),
                        (net.laneheader, ("decode", "scale_to_org", "visual"))):
        for nm in names:
            assert callable(getattr(head, nm)), nm
    preds = [{"rois": torch.tensor([[10.0, 20.0, 30.0, 40.0]]).numpy()}, {"rois": torch.zeros(0, 4).numpy()}]
    out = net.detectheader.invert_affine([[640, 640, 1280, 320, 0, 0]] * 2, preds)
    assert out[0]["rois"].tolist() == [[20.0, 10.0, 60.0, 20.0]]
    for fn in (net.detectheader.display, net.laneheader.visual):         # cv2 drawing: out of scope, says so instead of an AttributeError
        with pytest.raises(NotImplementedError):
            fn()


@pytest.mark.parametrize("fixture,cfg", [("tiny_hydranet.npz", "hydranet_tiny.yml"), ("tiny4_hydranet.npz", "hydranet_tiny4.yml")])
def test_tiny_state_dict_loads_reference_checkpoint(built, fixture, cfg):
    """5-stage and 4-stage (small-backbone family) tiny cfgs: the reference's own state_dict loads strictly, key ORDER included"""
    from multitask_hydranet_amd import HydraNet
    from tests.helpers import tiny_state
    z = load_npz(fixture)
    net = HydraNet(load_cfg(cfg))
    ref = tiny_state(z)
    missing, unexpected = net.load_state_dict(ref, strict=True)
    assert not missing and not unexpected
    assert list(net.state_dict().keys()) == [k[3:] for k in z.files if k.startswith("sd/")]


def test_small_cfg_is_the_references_four_stage_configuration(built):
    """cfgs/hydranet_small.yml = model/cfgs/hydranet_joint_small_backbone.yml: RegNet derivation pinned by loss_kats.npz, 4 stages ->
    p5_to_p6 is USED (no parameter excluded from the gradient exchange), focal seg loss"""
    from multitask_hydranet_amd import HydraNet
    from multitask_hydranet_amd.ddp import unused_parameters
    z = load_npz("loss_kats.npz")
    cfgs = load_cfg("hydranet_small.yml")
    net = HydraNet(cfgs)
    assert net.widths == z["regnet/hydranet_joint_small_backbone/widths"].tolist() == [24, 64, 152, 376]
    assert net.depths == z["regnet/hydranet_joint_small_backbone/depths"].tolist() == [1, 1, 4, 10]
    assert net.first_cell_counts() == [0, 1, 2, 3] and net._seg_cfg == (False, 0.3, True)
    assert "neck.bifpn.0.p5_to_p6.0.conv.weight" in net.state_dict() and "neck.bifpn.0.p6_down_channel.0.conv.weight" not in net.state_dict()
    assert unused_parameters(net) == ()
    big = HydraNet(load_cfg("hydranet_big.yml"))
    assert set(unused_parameters(big)) == {"neck.bifpn.0.p5_to_p6.0.conv.weight", "neck.bifpn.0.p5_to_p6.0.conv.bias",
                                           "neck.bifpn.0.p5_to_p6.1.weight", "neck.bifpn.0.p5_to_p6.1.bias"}


def test_anchor_table_matches_oracle(built):
    from multitask_hydranet_amd import HydraNet
    cfgs = load_cfg("hydranet_big.yml")
    net = HydraNet(cfgs)
    for (h, w) in ((640, 640), (512, 1024), (128, 256)):
        assert np.array_equal(net.anchors_for(h, w, "cpu")[0].numpy(), O.anchors_for(h, w, cfgs))


def test_static_losses_equal_oracle_loops():
    from tests import torch_losses as L
    z = load_npz("loss_kats.npz")
    t = lambda k: torch.from_numpy(z[k])
    for ann in (t("det/ann"), torch.ones(3, 16, 5)):
        a = L.det_loss(t("det/cls"), t("det/reg"), t("det/anchors"), ann)
        b = O.det_loss(t("det/cls"), t("det/reg"), t("det/anchors"), ann)
        assert_close(a[0], b[0], 1e-6, "det cls")
        assert_close(a[1], b[1], 1e-6, "det reg")
    cfgs = load_cfg("hydranet_big.yml")
    batch = O.synthetic_batch(cfgs, 2, 640, 640, seed=5)
    for tgt in (batch["gt_cls"], torch.cat([torch.ones(2, 400, 1), torch.zeros(2, 400, 1)], -1)):
        a = L.lane_cls_loss(tgt, t("lane/cls_pred"))
        b = O.lane_cls_loss(tgt, t("lane/cls_pred"))
        assert_close(a[0], b[0], 1e-6)
        assert_close(a[1], b[1], 1e-6)
        assert torch.equal(a[2], b[2]) and int(a[3]) == int(b[3])
        assert_close(L.lane_loc_loss(a[2], a[3], batch["gt_loc"], t("lane/loc_pred")),
                     O.lane_loc_loss(b[2], b[3], batch["gt_loc"], t("lane/loc_pred")), 1e-6)
    cw = torch.tensor([0.1, 0.5, 1.0, 5.0, 5.0])
    for kw in (dict(use_top_k=True, top_k_ratio=0.3, use_focal=False), dict(use_top_k=False, top_k_ratio=1.0, use_focal=True)):
        # channels-last logits (what the HIP path hands over) give the same value as contiguous ones
        lg = t("seg/logits").permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        assert_close(L.seg_loss(lg, t("seg/gt_ones"), cw, **kw), O.seg_loss(t("seg/logits"), t("seg/gt_ones"), cw.tolist(), **kw), 1e-6)
    # gradients of the static det loss equal the loop's
    c1 = t("det/cls").clone().requires_grad_(True)
    r1 = t("det/reg").clone().requires_grad_(True)
    c2 = t("det/cls").clone().requires_grad_(True)
    r2 = t("det/reg").clone().requires_grad_(True)
    a = L.det_loss(c1, r1, t("det/anchors"), t("det/ann"))
    b = O.det_loss(c2, r2, t("det/anchors"), t("det/ann"))
    (a[0].sum() + 50 * a[1].sum()).backward()
    (b[0].sum() + 50 * b[1].sum()).backward()
    assert_close(c1.grad, c2.grad, 1e-5, "dcls")
    assert_close(r1.grad, r2.grad, 1e-5, "dreg")


def test_phase_weights_algebra_cpu():
    """The 4-phase low-resolution form used for the final seg conv (ops.SegOutUp): Conv3x3(reflect_pad(nearest_up2(x)), W) equals
    depth_to_space(Conv3x3(replicate_pad(x), W_eff)) with W_eff built from ops._phase_matrix -- checked with plain torch on the CPU."""
    import torch
    import torch.nn.functional as F
    from multitask_hydranet_amd import ops as K
    torch.manual_seed(0)
    k, c, n, h, w = 5, 7, 2, 6, 9
    x = torch.randn(n, c, h, w, dtype=torch.float64)
    wt = torch.randn(k, c, 3, 3, dtype=torch.float64)
    ref = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), [1, 1, 1, 1], mode="reflect"), wt)
    T = K._phase_matrix(torch.device("cpu")).double()
    w_eff = (wt.reshape(k * c, 9) @ T.t()).view(k, c, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c, 3, 3)
    y4 = F.conv2d(F.pad(x, [1, 1, 1, 1], mode="replicate"), w_eff)            # [n, (py,px,o), h, w]
    out = y4.view(n, 2, 2, k, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, k, 2 * h, 2 * w)
    assert torch.allclose(out, ref, atol=1e-12)
    # and the transposed map used for the weight gradient: dW = dW_eff (phase-major) @ T
    g = torch.randn_like(w_eff)
    dw = (g.view(2, 2, k, c, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(k * c, 36) @ T).view(k, c, 3, 3)
    wt2 = wt.clone().requires_grad_(True)
    ((wt2.reshape(k * c, 9) @ T.t()).view(k, c, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c, 3, 3) * g).sum().backward()
    assert torch.allclose(dw, wt2.grad, atol=1e-12)


def test_dispatcher_ops_are_registered_with_schemas():
    """the HIP entry points are visible to the PyTorch dispatcher (torch.ops.hydranet_hip.*) with schemas and fake kernels
    (shape propagation works without a GPU); the forward / backward pairs are tied together with register_autograd."""
    import torch
    import multitask_hydranet_amd.torch_ops  # noqa: F401
    ns = torch.ops.hydranet_hip
    for name in ("conv1x1_bn_act_fwd", "conv1x1_bn_act_bwd", "seg_topk_ce_fwd", "seg_topk_ce_bwd", "argmax_channels", "det_postprocess",
                 "preprocess_bgr"):
        assert hasattr(ns, name), name
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 8, 8, 32, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(64, 32, 1, 1, device="cuda")
        v = torch.empty(64, device="cuda")
        out, z, coef = ns.conv1x1_bn_act_fwd(x, w, v, v, v.clone(), v.clone(), 1, 1e-5, 0.1, True)
        assert out.shape == (2, 8, 8, 64) and coef.shape == (4, 64)
        m = ns.argmax_channels(torch.empty(2, 8, 8, 5, device="cuda"))
        assert m.shape == (2, 8, 8) and m.dtype == torch.int64


def test_tuning_schedule_matches_reference_main():
    """the head-wise fine-tuning schedule (train.py:441-515) against what the reference's own main() did, epoch by epoch
    (tests/golden/tuning_schedule.json, recorded by make_golden.py::schedule_fixture from the reference source)"""
    import json
    import os
    from multitask_hydranet_amd.train import tuning_phase
    from tests.helpers import GOLD
    ref = json.load(open(os.path.join(GOLD, "tuning_schedule.json")))
    checked = 0
    for key, phases in ref.items():
        epoch_all, epoch_tuning, turns, fine = (int(v) for v in key.split(","))
        if not fine:
            assert set(phases) == {"initial"}                            # fine_tuning off: the param group is never touched
            continue
        period = len(phases) // turns
        for e, ph in enumerate(phases):
            turn, mine = tuning_phase(e, epoch_all, epoch_tuning, turns)
            assert mine == ph and turn == e // period, (key, e, mine, ph)
            checked += 1
    assert checked > 60
    with pytest.raises(AssertionError):
        tuning_phase(0, 5, 1, 2)                                         # 3 * epoch_tuning * tuning_turn > epoch: the reference asserts too


def test_coco_json_records(tmp_path):
    """COCO-side bookkeeping of HydraTrainer.valid (train.py:308-364, 412-421; head_detect/detection.py:217-229; gen_val_json.py:29-117)"""
    import json
    from multitask_hydranet_amd.coco_json import coco_ground_truth, detections_to_coco, invert_affine, write_results
    preds = [dict(rois=np.array([[64., 32., 128., 96.], [0., 0., 640., 640.]], np.float32), class_ids=np.array([7, 0]), scores=np.array([0.9, 0.31], np.float32)),
             dict(rois=np.array(()), class_ids=np.array(()), scores=np.array(())),
             dict(rois=np.array([[10., 20., 30., 50.]], np.float32), class_ids=np.array([2]), scores=np.array([0.5], np.float32))]
    metas = [[640, 640, 1920, 1080, 0, 0]] * 3
    preds = invert_affine(metas, preds)
    np.testing.assert_allclose(preds[0]["rois"][0], [192., 54., 384., 162.])      # x / (640/1920), y / (640/1080)
    rec = detections_to_coco(preds, first_image_id=9)
    assert [r["image_id"] for r in rec] == [9, 9, 11] and [r["category_id"] for r in rec] == [8, 1, 3]
    np.testing.assert_allclose(rec[0]["bbox"], [192., 54., 192., 108.])           # x, y, w, h
    np.testing.assert_allclose(rec[2]["bbox"], [30., 33.75, 60., 50.625])
    assert abs(rec[1]["score"] - 0.31) < 1e-6
    path = write_results(rec, str(tmp_path))
    assert json.load(open(path)) == rec and write_results([], str(tmp_path)) is None
    gt = coco_ground_truth([dict(file_name="a.jpg", height=1080, width=1920, annos=["100,200,300,260,7\n", "5.5,6.5,9.5,9.9,0"]),
                            dict(file_name="empty.jpg", height=1080, width=1920, annos=[]),
                            dict(file_name="b.jpg", height=720, width=1280, annos=[(1, 2, 3, 4, 8)])])
    assert [im["id"] for im in gt["images"]] == [1, 2] and [im["file_name"] for im in gt["images"]] == ["a.jpg", "b.jpg"]
    assert [a["image_id"] for a in gt["annotations"]] == [1, 1, 2] and [a["id"] for a in gt["annotations"]] == [1, 2, 3]
    assert gt["annotations"][0]["bbox"] == [100.0, 200.0, 200, 60] and gt["annotations"][0]["area"] == 12000
    assert gt["annotations"][1]["bbox"] == [5.5, 6.5, 4, 3] and len(gt["categories"]) == 9 and gt["categories"][7]["name"] == "vehicle"


def test_conv_work_accounting_matches_survey_totals():
    """multitask_hydranet_amd.accounting (the per-segment floors of bench.py's `segments` table) against the figures the survey measured with
    forward hooks on the reference: MACs, conv input / output elements, per-segment FLOP split; and the resulting
    segment-wise roofline floor (0.177 ms per image at 512x1024)"""
    from multitask_hydranet_amd.accounting import conv_work, segment_floors_ms
    cfgs = load_cfg("hydranet_big.yml")
    for (h, w), (gflop, sx, sy, split) in {(512, 1024): (81.13, 128.33, 82.50, (12.02, 1.48, 64.85, 2.10, 0.68)),
                                          (640, 640): (63.40, 100.38, 64.48, (9.39, 1.15, 50.67, 1.64, 0.54))}.items():
        wk = conv_work(cfgs, h, w)
        assert abs(2 * sum(v["macs"] for v in wk.values()) / 1e9 - gflop) < 0.006
        assert abs(sum(v["x"] for v in wk.values()) / 1e6 - sx) < 0.1 and abs(sum(v["y"] for v in wk.values()) / 1e6 - sy) < 0.02
        for k, g in zip(("backbone", "neck", "seg", "det", "lane"), split):
            assert abs(2 * wk[k]["macs"] / 1e9 - g) < 0.006, k
    fl = segment_floors_ms(cfgs, 512, 1024, 16)
    assert abs(sum(fl.values()) / 16 - 0.177) < 0.004 and fl["seg"] > fl["backbone"] > fl["det"] > fl["neck"] > fl["lane"]


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 8` without torch.distributed.run must start its own 8 ranks or FAIL -- never measure one GPU and report it
    (VERDICT r3 missing 1).  This container has no GPU: the parent refuses before touching the runtime, exit code 2, no JSON line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HN_BENCH_ONE_DEVICE")}
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node would run the bench")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert not any(l.startswith("{") for l in r.stdout.splitlines())
    assert "refusing" in r.stderr


def test_live_traffic_sampling_fails_soft():
    """bench.py samples roofline.traffic with two rocprofv3 child passes; whatever goes wrong there (no profiler, no GPU as in this container,
    a pass over its time limit) must come back as {"error": ...} -- the line then keeps the committed profile's figure and says so -- and
    must never raise or hang.  The figure's bookkeeping: a live result replaces the committed one only for the workload it was taken on."""
    import argparse
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the passes succeed (covered by the bench run itself)")
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(batch=16, res="512x1024", cfg=os.path.join(ROOT, "cfgs", "hydranet_big.yml"))
    t0 = time.perf_counter()
    r = bench.sample_traffic_live(args, timeout_s=60)
    assert isinstance(r, dict) and "error" in r and "bytes" not in r, r
    assert time.perf_counter() - t0 < 90
    committed, src = bench.measured_traffic(16, 512, 1024, "phase")
    assert committed and src.startswith("profiles/r06_")
    assert bench.measured_traffic(8, 512, 1024, "phase") == (None, None)


def test_policy_switches_are_constants_without_the_tuning_flag():
    """the package reads no HN_* environment variable unless HN_TUNING=1 / ab is set (_lib.policy): a stray HN_FUSED_BN=0
    or HN_LIB_AB in a user's environment must not change what the product runs"""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); "
            "from multitask_hydranet_amd import _lib, ops; "
            "print(int(ops.FUSED_BN), int(ops.FUSED_XBLOCK), _lib.policy('HN_LIB_AB', ''))" % ROOT)
    base = {k: v for k, v in os.environ.items() if not k.startswith("HN_")}
    plain = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(base, HN_FUSED_BN="0", HN_FUSED_XBLOCK="0", HN_LIB_AB="/nonexistent.so"))
    assert plain.returncode == 0 and plain.stdout.split() == ["1", "1"], (plain.stdout, plain.stderr[-500:])
    ab = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(base, HN_TUNING="ab", HN_FUSED_BN="0", HN_LIB_AB="/nonexistent.so"))
    assert ab.returncode == 0 and ab.stdout.split() == ["0", "1", "/nonexistent.so"], (ab.stdout, ab.stderr[-500:])
