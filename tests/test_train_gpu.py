"""The training-loop surface (multitask_hydranet_amd.train.HydraTrainer, reference model/train.py:31-269) on the device: a few optimizer
steps on the tiny fixture batch through the HIP path -- Adam + cosine LR, to_gpu, cal_total_loss (one launch), the seg validation with the
device mIoU, checkpoint save / reload.  The first step's losses must equal the reference's recorded values (same state_dict, same batch);
the following steps must keep the loss finite and move every trainable parameter that received a gradient."""
import os
import tempfile

import pytest
import torch

from tests.helpers import load_cfg, load_npz, tiny_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import __graft_entry__ as g
    g.build()
    z = load_npz("tiny_hydranet.npz")
    cfgs = load_cfg("hydranet_tiny.yml")
    cfgs["train"].update(dict(continue_train=False, weight_file="", epoch=1, lr=1e-4, weight_decay=0.0))
    batch = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("in/")}
    return z, cfgs, batch


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_trainer_steps_and_validation(setup):
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    loader = [dict(batch), dict(batch), dict(batch)]
    tr = HydraTrainer(cfgs, trainloader=loader, validloader=[dict(batch)], iters_per_epoch=len(loader))
    tr.hydranet.load_state_dict(tiny_state(z))
    tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
    before = {n: p.detach().clone() for n, p in tr.hydranet.named_parameters()}
    losses = []
    for b in loader:
        ld = tr.train_step({k: v.clone() for k, v in b.items()})
        losses.append({k: float(v.detach()) for k, v in ld.items()})
    # step 1 == the reference's recorded losses of the same state / batch (1e-2; lane terms 6e-2, as in the model tests)
    for k in ("loss_seg", "loss_det_cls", "loss_det_reg", "loss_lane_cls_pos", "loss_lane_cls_neg", "loss_lane_loc"):
        ref = float(z["loss/" + k])
        tol = 6e-2 if "lane" in k else (2.5e-2 if k == "loss_det_reg" else 1e-2)     # (det_reg: see test_model_gpu's end-to-end test)
        assert abs(losses[0][k] - ref) <= tol * max(abs(ref), 1e-6), (k, losses[0][k], ref)
    assert all(v == v and abs(v) < 1e9 for step in losses for v in step.values())
    # the weighted total is the reference's formula (train.py:192-203)
    s, d, l = cfgs["segment"], cfgs["detection"], cfgs["lane"]
    want = losses[0]["loss_seg"] * s["segment_weight"] + (losses[0]["loss_det_cls"] * d["loss_cls_weight"] +
            losses[0]["loss_det_reg"] * d["loss_reg_weight"]) * d["detection_weight"] + \
        (losses[0]["loss_lane_cls_pos"] * l["loss_cls_pos_weight"] + losses[0]["loss_lane_cls_neg"] * l["loss_cls_neg_weight"] +
         losses[0]["loss_lane_loc"] * l["loss_loc_weight"]) * l["lane_weight"]
    assert abs(losses[0]["total_loss"] - want) <= 1e-5 * abs(want)
    # every parameter whose gradient is not identically zero moved (exactly-zero gradients -- a conv bias that feeds BatchNorm, layers behind
    # a dead ReLU, the 1x1-pixel pyramid level of the tiny cfg -- stay put under Adam with weight_decay = 0, as in the reference)
    with_grad = [n for n, p in tr.hydranet.named_parameters() if p.grad is not None]
    nonzero = [n for n, p in tr.hydranet.named_parameters() if p.grad is not None and bool((p.grad != 0).any())]
    moved = [n for n in nonzero if not torch.equal(dict(tr.hydranet.named_parameters())[n].detach(), before[n])]
    assert len(with_grad) >= 0.95 * len(before) and len(nonzero) >= 0.85 * len(with_grad) and moved == nonzero, \
        (len(before), len(with_grad), len(nonzero), len(moved))
    assert tr.scheduler.last_epoch == len(loader)
    iou = tr.valid()
    assert iou is not None and torch.isfinite(torch.as_tensor(iou)).all()
    assert tr.hydranet.training                      # valid() switches back to train mode
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ckpt.pth")
        tr.save(path)
        sd = torch.load(path, map_location="cpu")
        assert set(sd.keys()) == set(tr.hydranet.state_dict().keys())
        from multitask_hydranet_amd import HydraNet
        net2 = HydraNet(cfgs).cuda()
        net2.load_state_dict({"module." + k: v for k, v in sd.items()})      # DDP-style prefixes are accepted (train.py:96-109)
        for (n1, p1), (n2, p2) in zip(tr.hydranet.state_dict().items(), net2.state_dict().items()):
            assert n1 == n2 and torch.equal(p1.cpu(), p2.cpu()), n1


def test_trainer_captured_step_equals_eager(setup):
    """HydraTrainer(capture_step=True): from the third iteration on the forward + loss + backward of an iteration is one hipGraph replay on
    static input buffers -- the same launches in the same order, so losses and parameters must equal the eager trainer's bit for bit."""
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    g = torch.Generator().manual_seed(3)
    loader = []
    for i in range(5):                                       # different images per iteration, same targets
        b = dict(batch)
        b["image"] = batch["image"] + 0.05 * torch.randn(batch["image"].shape, generator=g)
        loader.append(b)
    runs = []
    for capture in (False, True):
        tr = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader), capture_step=capture)
        tr.hydranet.load_state_dict(tiny_state(z))
        tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
        losses = []
        for b in loader:
            ld = tr.train_step({k: v.clone() for k, v in b.items()})
            losses.append({k: float(v.detach()) for k, v in ld.items()})
        assert (tr._cap is not None) == capture
        runs.append((losses, {n: p.detach().clone() for n, p in tr.hydranet.named_parameters()},
                     {n: b_.detach().clone() for n, b_ in tr.hydranet.named_buffers()}))
    (l0, p0, b0), (l1, p1, b1) = runs
    for step, (a, b) in enumerate(zip(l0, l1)):
        assert a == b, (step, a, b)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n
    for n in b0:
        assert torch.equal(b0[n], b1[n]), n


def test_rccl_gradient_exchange_world1_in_graph():
    """bench.py --ddp-world1: RCCL process group of one rank, bucketed ncclAvg all-reduce of the gradients captured INSIDE the hipGraph
    on the reducer's side stream (the N > 1 code path, executed on one GPU).  The step must report the in-graph exchange and reproduce
    the loss of the run without it (the average over one rank is the identity)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-optimizer", "--steps", "2", "--warmup", "1",
            "--batch", "2", "--res", "256x512"]
    outs = []
    for extra in ([], ["--ddp-world1"], ["--ddp-world1", "--grad-payload", "bf16"]):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", HN_BENCH_GRAD_NORM="1")
        r = subprocess.run(base + extra, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]))
    plain, ddp, ddp16 = outs
    assert plain["config"]["grad_allreduce"] is None
    for o in (ddp, ddp16):
        assert "inside the hipGraph" in o["config"]["grad_allreduce"], o["config"]["grad_allreduce"]
        assert o["config"]["hipgraph"] is True
        assert o["loss"] == plain["loss"]
    # fp32 payload: the deferred weight gradients are written into the buckets by the backward kernels themselves (ops.grad_out), the
    # gather only moves the rest (85 % in place at the bench's 512 x 1024; at this size the deep stages' XBlocks take the unfused path)
    import re
    assert int(re.search(r"(\d+)% of the elements written in place", ddp["config"]["grad_allreduce"]).group(1)) >= 5
    # the exchanged gradients: fp32 payload = the plain run's gradients exactly, bf16 payload = to bf16 rounding
    assert ddp["grad_norm"] == plain["grad_norm"], (ddp["grad_norm"], plain["grad_norm"])
    assert abs(ddp16["grad_norm"] - plain["grad_norm"]) <= 5e-3 * plain["grad_norm"]
    assert "bfloat16" in ddp16["config"]["grad_allreduce"]


def test_two_rank_control_flow_on_one_gpu():
    """the N > 1 launch contract of bench.py (torch.distributed.run, one process per rank, barrier + max-over-ranks timing, rank 0 prints
    the JSON line) with the two test hooks that make it runnable on a 1-GPU box: both ranks on cuda:0, gloo instead of RCCL (which refuses
    two ranks on one device).  The gradient exchange then runs after each replay (only RCCL collectives are captured in the graph)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ["--gpus", "2", "--no-cpu-baseline", "--no-optimizer", "--steps", "2", "--warmup", "1", "--batch", "2", "--res", "256x512"]
    env = dict(os.environ, HN_BENCH_ONE_DEVICE="1", HN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # (1) the driver's N > 1 command line; (2) plain `python bench.py --gpus 2`: bench.py starts its own two ranks as a child process
    for cmd in ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + tail,
                [sys.executable, os.path.join(root, "bench.py")] + tail):
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stdout.strip().splitlines()[-1].startswith("{"), "the JSON line must be the last line on stdout"
        o = json.loads(r.stdout.strip().splitlines()[-1])
        assert o["n_gpus"] == 2 and o["config"]["global_batch"] == 4 and o["config"]["parallelism"] == "dp2" and o["scaling"] == "weak"
        assert "gloo" in o["config"]["grad_allreduce"]
        assert o["value"] > 0 and o["loss"] == o["loss"]
        assert o["dist_ranks"] == 2 and o["dist_backend"] == "gloo" and len(o["per_rank_ms_per_step"]) == 2
    # a world size that does not match --gpus is refused (exit code 2, no JSON line) instead of measured under the wrong label
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + tail, capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_two_rank_gradients_are_the_mean_of_the_single_rank_gradients():
    """DDP semantics (model/train.py:130-137) with the real HIP model at world size 2: two gloo ranks on cuda:0, a different batch per rank
    (tests/ddp_two_rank_worker.py): the exchanged gradients of each rank == the mean of the two single-rank gradients (fp32 payload, 1e-6;
    in-place bucket slots on the first step, accumulation into the bucket views on the second), and HydraTrainer's captured data-parallel
    step == its eager hook form bit for bit, identical parameters on both ranks after five iterations."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "ddp_two_rank_worker.py")], stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, env=env))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % rank) in o, o[-3000:]


def test_trainer_captured_data_parallel_step_rccl_world1(setup):
    """HydraTrainer(capture_step=True) with the data-parallel machinery on (force_distribute: RCCL process group of one rank): two eager
    hook-mode iterations, then the captured step with every bucket's gather + ncclAvg all-reduce IN the hipGraph (ddp.capture_exchange_step,
    the recipe bench.py times).  The average over one rank is the identity: losses, parameters and buffers after five iterations equal the
    plain single-GPU eager trainer's bit for bit."""
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    g = torch.Generator().manual_seed(5)
    loader = []
    for i in range(5):
        b = dict(batch)
        b["image"] = batch["image"] + 0.05 * torch.randn(batch["image"].shape, generator=g)
        loader.append(b)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    runs = []
    for ddp in (False, True):
        tr = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader), capture_step=ddp, force_distribute=ddp)
        tr.hydranet.load_state_dict(tiny_state(z))
        tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
        losses = []
        for b in loader:
            ld = tr.train_step({k: v.clone() for k, v in b.items()})
            losses.append({k: float(v.detach()) for k, v in ld.items()})
        torch.cuda.synchronize()
        if ddp:
            assert tr._cap is not None and tr._cap[4], "the exchange was not captured inside the hipGraph"
            assert tr.reducer.captured and "inside the hipGraph" in tr.reducer.describe()
        runs.append((losses, {n: p.detach().clone() for n, p in tr.hydranet.named_parameters()},
                     {n: b_.detach().clone() for n, b_ in tr.hydranet.named_buffers()}))
    (l0, p0, b0), (l1, p1, b1) = runs
    assert l0 == l1, (l0, l1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n
    for n in b0:
        assert torch.equal(b0[n], b1[n]), n


def test_trainer_short_last_batch_and_phase_switch_under_captured_data_parallel_step(setup):
    """with the exchange captured in the step's hipGraph, a batch of another shape (the loaders' short last batch) must not
    trigger a second capture on the spot.  It and the next iteration run on the eager hook path, the step is captured again after that --
    and a fine-tuning phase switch does the same.  The whole sequence equals the plain eager trainer bit for bit (one rank: the average
    is the identity)."""
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    g = torch.Generator().manual_seed(9)
    n = batch["image"].shape[0]
    assert n >= 2
    loader = []
    for i in range(9):
        b = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        b["image"] = batch["image"] + 0.05 * torch.randn(batch["image"].shape, generator=g)
        if i == 4:                                            # the short batch
            b = {k: (v[:n - 1].clone() if isinstance(v, torch.Tensor) and v.shape[:1] == (n,) else v) for k, v in b.items()}
        loader.append(b)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    runs = []
    for ddp in (False, True):
        tr = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader), capture_step=ddp, force_distribute=ddp)
        tr.hydranet.load_state_dict(tiny_state(z))
        tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
        losses, captured = [], []
        for i, b in enumerate(loader):
            if i == 7:
                tr.set_phase("seg")                           # head-only phase: its own reducer and its own captured step
            ld = tr.train_step({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
            losses.append({k: float(v.detach()) for k, v in ld.items()})
            captured.append(tr._cap is not None)
        torch.cuda.synchronize()
        if ddp:
            # iterations 0, 1 eager; 2, 3 replayed; 4 (short) and 5 eager; 6 replayed; 7, 8 eager (new phase)
            assert captured == [False, False, True, True, False, False, True, False, False], captured
        runs.append((losses, {k: p.detach().clone() for k, p in tr.hydranet.named_parameters()}))
    (l0, p0), (l1, p1) = runs
    assert l0 == l1, (l0, l1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_eval_after_training_steps_sees_the_updated_weights(setup):
    """hn_adam_step and the training-mode BatchNorm kernels write parameters / running statistics through raw pointers, which
    torch's version counters do not see.  The eval-mode caches (PackPlan.fresh, the det towers' per-level coefficient rows) must still
    notice: eval -> train steps -> eval gives what a module with freshly packed operands gives."""
    from multitask_hydranet_amd import HydraNet
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    tr = HydraTrainer(cfgs, trainloader=[dict(batch)] * 3, validloader=None, iters_per_epoch=3)
    net = tr.hydranet
    net.load_state_dict(tiny_state(z))
    net.lane_points_per_line = int(z["meta/lane_points_per_line"])
    x = batch["image"].cuda()

    def eval_out():
        net.eval()
        with torch.no_grad():
            o = net(x)
            o2 = net(x)                                   # the second eval forward reuses the packed operands (PackPlan.fresh)
        net.train()
        flat = lambda d: [d["seg"], d["detection"]["regression"], d["detection"]["classification"], d["lane"]["predict_cls"], d["lane"]["predict_loc"]]
        for a, b in zip(flat(o), flat(o2)):
            assert torch.equal(a, b)
        return [t.float().clone() for t in flat(o)]
    first = eval_out()
    for _ in range(3):
        tr.train_step({k: v.clone() for k, v in batch.items()})
    after = eval_out()
    assert any(not torch.equal(a, b) for a, b in zip(first, after)), "three optimizer steps must change the eval outputs"
    fresh = HydraNet(cfgs).cuda()
    fresh.load_state_dict(net.state_dict())
    fresh.eval()
    with torch.no_grad():
        o = fresh(x)
    want = [o["seg"], o["detection"]["regression"], o["detection"]["classification"], o["lane"]["predict_cls"], o["lane"]["predict_loc"]]
    for a, b in zip(after, want):
        assert torch.equal(a, b.float()), "an eval-mode cache served operands from before the optimizer steps"


def test_fine_tuning_schedule_two_turns(setup):
    """main()'s head-wise fine-tuning schedule (train.py:441-515) through run_training on the tiny cfg: epoch = 8, epoch_tuning = 1,
    tuning_turn = 2 -> per turn joint, lane, det, seg.  After every epoch only the parameters of that phase's param group may have moved
    (joint: the whole model); Adam's state survives the swaps; head-only phases run NO backward outside their head (no gradient appears on
    any other parameter) while the BatchNorm running statistics of the shared trunk keep updating and all six losses are reported.
    A head-only step moves its head's parameters exactly like a full-backward step does (same gradients for those parameters)."""
    import copy
    from multitask_hydranet_amd.train import HydraTrainer, run_training, tuning_phase
    z, cfgs0, batch = setup
    cfgs = copy.deepcopy(cfgs0)
    cfgs["train"].update(dict(epoch=8, fine_tuning=True, epoch_tuning=1, tuning_turn=2, lr=1e-3))
    loader = [dict(batch), dict(batch)]
    tr = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader))
    tr.hydranet.load_state_dict(tiny_state(z))
    tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
    net = tr.hydranet
    prefix = {"lane": "laneheader.", "det": "detectheader.", "seg": "segheader."}
    log = []
    orig = tr.train_one_epoch

    def spy(epoch):
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        stats = net.state_dict()["backbone.net.stem.bn.running_mean"].clone()
        orig(epoch)
        torch.cuda.synchronize()
        moved = {n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n])}
        turn, phase = tuning_phase(epoch, 8, 1, 2)
        assert tr.phase == phase and net.grad_scope == (None if phase == "joint" else phase)
        if phase == "joint":
            assert len(moved) > 0.8 * len(before)
        else:
            assert moved and all(n.startswith(prefix[phase]) for n in moved), (phase, sorted(moved)[:5])
            own = [n for n, _ in net.named_parameters() if n.startswith(prefix[phase])]
            assert len(moved) >= 0.7 * len(own)
            # no backward ran outside the head: nothing else holds a gradient
            assert all(p.grad is None for n, p in net.named_parameters() if not n.startswith(prefix[phase])), phase
        assert not torch.equal(net.state_dict()["backbone.net.stem.bn.running_mean"], stats)       # training-mode forward everywhere
        log.append(phase)
    tr.train_one_epoch = spy
    run_training(tr, valid_every_epoch=False, log=lambda *a: None)
    assert log == ["joint", "lane", "det", "seg"] * 2
    st = tr.optimizer.state
    lane_w = dict(net.named_parameters())["laneheader.conv_cls_conv.0.weight"]
    stem_w = dict(net.named_parameters())["backbone.net.stem.conv.weight"]
    assert int(st[lane_w]["step"]) == 2 * 2 * 2 and int(st[stem_w]["step"]) == 2 * 2       # (joint + lane) x 2 turns x 2 iterations / joint only
    # a head-only step == the same step with the full backward, for the head's parameters (the reference's semantics)
    res = []
    for scoped in (True, False):
        t2 = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader))
        t2.hydranet.load_state_dict(tiny_state(z))
        t2.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
        t2.set_phase("det")
        if not scoped:
            t2.hydranet.grad_scope = None                    # the reference: full backward, optimizer only holds the head
        ld = t2.train_step({k: v.clone() for k, v in batch.items()})
        res.append(({k: float(v) for k, v in ld.items()},
                    {n: p.detach().clone() for n, p in t2.hydranet.named_parameters() if n.startswith("detectheader.")}))
    assert res[0][0] == res[1][0]                            # all six losses reported, identical
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


def test_valid_writes_coco_results_and_lane_json(setup, tmp_path):
    """HydraTrainer.valid (train.py:271-438) on the device: per-batch losses, streaming mIoU, detection records in COCO-json form out of
    hn_det_postprocess (+ invert_affine to the source-image size), lane prediction json out of hn_lane_decode_nms"""
    import json
    from multitask_hydranet_amd.lane_codec import LaneCodec
    from multitask_hydranet_amd.train import HydraTrainer
    z, cfgs, batch = setup
    vb = dict(batch)
    vb["src_image_shape"] = [{"width": 1920, "height": 1080}] * batch["image"].shape[0]
    tr = HydraTrainer(cfgs, trainloader=[dict(batch)], validloader=[vb, dict(vb)], iters_per_epoch=1)
    tr.hydranet.load_state_dict(tiny_state(z))
    tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
    h, w = batch["image"].shape[2], batch["image"].shape[3]
    coder = LaneCodec(w, h, cfgs["lane"]["anchor_stride"], h // cfgs["lane"]["interval"])
    iou = tr.valid(0, eval_dir=str(tmp_path), lane_coder=coder, det_conf_thres=0.02)
    lv = tr.last_valid
    assert torch.isfinite(torch.as_tensor(iou)).all() and len(lv["losses"]) == 2 and set(lv["losses"][0]) >= {"loss_seg", "loss_det_cls", "total_loss"}
    assert lv["losses"][0] == lv["losses"][1]                      # eval mode: the same batch twice gives the same losses
    assert len(lv["lane_result"]) == 2 * batch["image"].shape[0] and "Lines" in lv["lane_result"][0]["pr_result"]
    rec = lv["detect_result"]
    assert rec and lv["detect_json"] and json.load(open(lv["detect_json"])) == rec
    n = batch["image"].shape[0]
    ids = {r["image_id"] for r in rec}
    assert ids <= set(range(1, 2 * cfgs["train"]["batch_size_valid"] + 1)) and all(1 <= r["category_id"] <= 9 for r in rec)
    # boxes are in SOURCE-image pixels (1920 x 1080) after invert_affine, x/y/w/h
    assert max(r["bbox"][0] + r["bbox"][2] for r in rec) <= 1920 + 1e-3 and max(r["bbox"][1] + r["bbox"][3] for r in rec) <= 1080 + 1e-3
    assert max(r["bbox"][0] + r["bbox"][2] for r in rec) > w


def test_eager_hook_exchange_world1_keeps_first_step_gradients(setup):
    """GradReducer in eager hook mode (the HydraTrainer multi-GPU path) at world size 1 with RCCL: the side-stream gather reads the gradients
    autograd allocated on the main stream; they must stay alive until the join (stream lifetime).  Two small buckets, first
    step and a second one: the exchanged gradients equal those of a run without the reducer, bit for bit (average over one rank)."""
    import torch.distributed as dist
    from multitask_hydranet_amd import HydraNet
    from multitask_hydranet_amd.ddp import GradReducer, unused_parameters
    z, cfgs, batch = setup
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    gb = {k: v.cuda() for k, v in batch.items()}
    grads = []
    for use in (False, True):
        net = HydraNet(cfgs).cuda().train()
        net.load_state_dict(tiny_state(z))
        net.lane_points_per_line = int(z["meta/lane_points_per_line"])
        red = GradReducer(list(net.named_parameters()), world_size=1, skip=unused_parameters(net), bucket_bytes=64 << 10,
                          force_collectives=True) if use else None
        if red is not None:
            assert len(red.buckets) >= 4 and red.active
        per_step = []
        for step in range(2):
            net.zero_grad(set_to_none=(red is None or step == 0))
            # allocator pressure between backward and the join: recycled blocks would be overwritten here
            out = net(gb["image"])
            net.total_loss(net.cal_loss(out, gb)).backward()
            junk = [torch.full((1 << 18,), 7.0, device="cuda") for _ in range(8)]
            if red is not None:
                red.finish()
                # step 0 (.grad was None): the deferred HIP weight gradients were written straight into the bucket slots (ops.grad_out);
                # step 1: autograd accumulates into the bucket views of step 0
                assert red.direct_fraction() > 0.0, (step, red.direct_fraction())
            torch.cuda.synchronize()
            per_step.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
            del junk
        grads.append(per_step)
    for step in range(2):
        assert grads[0][step].keys() == grads[1][step].keys()
        for n in grads[0][step]:
            assert torch.equal(grads[0][step][n], grads[1][step][n]), (step, n)
