"""CPU: host logic of the product package (no kernel is launched here): the C-ABI library loads and exports every symbol
include/hydranet_hip.h declares, the module reproduces the reference's state_dict contract, and the static-shape losses equal
the oracle's per-image loops."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import hydranet_oracle as O
from tests.helpers import assert_close, load_cfg, load_npz


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


def test_tiny_state_dict_loads_reference_checkpoint(built):
    from multitask_hydranet_amd import HydraNet
    from tests.helpers import tiny_state
    z = load_npz("tiny_hydranet.npz")
    net = HydraNet(load_cfg("hydranet_tiny.yml"))
    missing, unexpected = net.load_state_dict(tiny_state(z), strict=True)
    assert not missing and not unexpected


def test_anchor_table_matches_oracle(built):
    from multitask_hydranet_amd import HydraNet
    cfgs = load_cfg("hydranet_big.yml")
    net = HydraNet(cfgs)
    for (h, w) in ((640, 640), (512, 1024), (128, 256)):
        assert np.array_equal(net.anchors_for(h, w, "cpu")[0].numpy(), O.anchors_for(h, w, cfgs))


def test_static_losses_equal_oracle_loops():
    from multitask_hydranet_amd import losses as L
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
