"""Training-loop surface around the HIP hot path (reference: model/train.py:31-269 HydraTrainer).

Same roles and names as the reference's trainer for the part that touches the hot path: model construction (+ `module.`-prefixed
checkpoint loading, train.py:96-126), the data-parallel wrap (train.py:130-137 -> multitask_hydranet_amd.ddp.GradReducer, one process per
GPU over RCCL), Adam + per-iteration CosineAnnealingLR (train.py:147-150), `cal_total_loss` (train.py:192-203), `to_gpu` (train.py:228-239),
`train_one_epoch` (train.py:241-269), `valid` (train.py:271-438: losses, streaming mIoU on the device, detection results in COCO json form
from the device post-process, lane decode on the device) and `main`'s head-wise fine-tuning schedule (train.py:441-515: run_training /
tuning_phase / HydraTrainer.set_phase), the lane F1 metric of train.py:188,397,433 (lane_metric.LaneMetric: rasterisation + IoU counts on the
device).  The dataset / augmentation pipeline (cv2 + imgaug) and COCOeval (pycocotools) stay outside (out of scope); any
iterable of batch dicts with the Collater contract
(dataset/dataloader.py:557-633) drives the loop.

Launch for N GPUs of one node:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 your_script.py
(the trainer reads RANK / LOCAL_RANK / WORLD_SIZE; world size 1 needs no launcher).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

from .ddp import GradReducer, broadcast_state, capture_exchange_step, unused_parameters
from .metrics import IntersectionOverUnion
from .model import HydraNet
from .optim import Adam


PHASES = ("joint", "lane", "det", "seg")


def tuning_phase(epoch: int, epoch_all: int, epoch_tuning: int, tuning_turn: int):
    """main(), train.py:445-505: -> (turn, phase) of `epoch` under the head-wise fine-tuning schedule: every turn is `epoch_joint` joint
    epochs followed by `epoch_tuning` epochs each of lane-only, det-only and seg-only optimisation."""
    assert 3 * epoch_tuning * tuning_turn <= epoch_all
    epoch_joint = int(epoch_all / tuning_turn) - epoch_tuning * 3
    period = epoch_joint + epoch_tuning * 3
    turn, e = int(epoch / period), epoch % period
    if e < epoch_joint:
        return turn, "joint"
    if e < epoch_joint + epoch_tuning:
        return turn, "lane"
    if e < epoch_joint + 2 * epoch_tuning:
        return turn, "det"
    return turn, "seg"


def run_training(trainer: "HydraTrainer", valid_every_epoch: bool = True, log=print):
    """main(), train.py:441-515: the epoch loop with the fine-tuning schedule of cfgs["train"] (fine_tuning / epoch_tuning / tuning_turn)"""
    t = trainer.cfgs["train"]
    epoch_all = t["epoch"]
    fine = t.get("fine_tuning", False)
    for epoch in range(epoch_all):
        if fine:
            turn, phase = tuning_phase(epoch, epoch_all, t["epoch_tuning"], t["tuning_turn"])
            log("======= TURN %i %s TRAINING =======" % (turn, phase.upper()))
            trainer.set_phase(phase)
        trainer.train_one_epoch(epoch)
        if valid_every_epoch and trainer.validloader is not None:
            log("=========================== VALIDATION %i ===========================" % epoch)
            trainer.valid(epoch)
    log("============== finish training ==============")


from .ops.xstage import xstage_assert_ok as K_xstage_assert_ok


class HydraTrainer:
    def __init__(self, cfgs: dict, trainloader: Optional[Iterable] = None, validloader: Optional[Iterable] = None, iters_per_epoch: Optional[int] = None,
                 grad_payload: torch.dtype = torch.float32, capture_step: bool = False, hip_adam: bool = True, force_distribute: bool = False):
        """capture_step: after two eager iterations the forward + loss + backward of an iteration is captured as ONE hipGraph and replayed
        on static input buffers (875 vs ~500 img/s on the bench workload: ~1200 launches per step are host-bound when issued one by one);
        Adam / LR steps stay eager.  Needs batches of one fixed shape; a different shape re-captures.  With more than one rank the
        gradient exchange is part of the captured step (RCCL: every bucket's gather + all-reduce captured in line where its last gradient
        appears, ddp.capture_exchange_step -- the form bench.py times; other backends, whose collectives cannot be captured: the graph holds
        forward + loss + backward and the buckets are exchanged right after each replay).
        force_distribute: run the data-parallel machinery (process group, bucketed all-reduce, in-graph exchange) at world size 1 too --
        the N > 1 code path on a single GPU (tests; the average over one rank is the identity)."""
        self.cfgs = cfgs
        self.capture_step = capture_step
        self._cap = None                       # (shape key, graph, static batch, static loss dict)
        self._eager_iters = 0
        self._stream = None                    # data-parallel captured steps: ONE stream for the eager iterations and the capture (ddp.py)
        t = cfgs["train"]
        self.train_detect, self.train_seg, self.train_lane = t["train_detect"], t["train_seg"], t["train_lane"]
        self.print_interval = t.get("print_interval", 10)
        self.trainloader, self.validloader = trainloader, validloader
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(self.device)
        self._force_distribute = bool(force_distribute)
        if (self.world > 1 or force_distribute) and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if "MASTER_PORT" not in os.environ:          # a free port: a constant collides with a second trainer / a socket in TIME_WAIT
                    import socket
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)

        self.hydranet = HydraNet(cfgs=cfgs).to(self.device)
        if t.get("continue_train") and t.get("weight_file"):
            # files written from a DDP wrapper carry "module." prefixes (train.py:96-109 deparallel_model): load_state_dict strips them
            self.hydranet.load_state_dict(torch.load(t["weight_file"], map_location=self.device))
        broadcast_state(self.hydranet)                               # what DDP does at construction (train.py:137)
        self.use_distribute = self.world > 1 or self._force_distribute
        self.reducer = None
        self.phase = "joint"
        self._grad_payload = grad_payload
        self._reducers = {}                                          # one bucket plan per fine-tuning phase (the set of live gradients differs)
        if self.use_distribute:
            self.reducer = self._reducers["joint"] = self._make_reducer("joint")
            if capture_step:
                self._stream = torch.cuda.Stream(device=self.device)

        self.lr, self.weight_decay, self.epoch = t["lr"], t["weight_decay"], t["epoch"]
        n_iter = iters_per_epoch if iters_per_epoch is not None else (len(trainloader) if hasattr(trainloader, "__len__") else 1)
        self.total_iters = max(1, n_iter * self.epoch)
        # torch.optim.Adam's update rule and state layout (train.py:147); hip_adam: all 693 tensors in one launch (optim.py) instead of the
        # foreach implementation's ~10 multi-tensor launches (3.9 -> 0.4 ms per step)
        opt_cls = Adam if hip_adam else torch.optim.Adam
        self.optimizer = opt_cls(self.hydranet.parameters(), self.lr, weight_decay=self.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, self.total_iters, eta_min=1e-8)     # iteration based

        self._one = torch.ones((), device=self.device)            # root gradient, allocated once
        s, d, l = cfgs["segment"], cfgs["detection"], cfgs["lane"]
        self.segment_weight = s["segment_weight"]
        self.loss_cls_weight, self.loss_reg_weight, self.detection_weight = d["loss_cls_weight"], d["loss_reg_weight"], d["detection_weight"]
        self.loss_cls_pos_weight, self.loss_cls_neg_weight = l["loss_cls_pos_weight"], l["loss_cls_neg_weight"]
        self.loss_loc_weight, self.lane_weight = l["loss_loc_weight"], l["lane_weight"]
        if self.train_seg:
            self.metric_evaluator_iou = IntersectionOverUnion(n_classes=len(s["class_list"]), device=self.device)

    # ------------------------------------------------------------------------------------------------------------------------------
    def _phase_module(self, phase):
        return {"joint": self.hydranet, "lane": self.hydranet.laneheader, "det": self.hydranet.detectheader, "seg": self.hydranet.segheader}[phase]

    def _make_reducer(self, phase):
        if phase == "joint":
            named = list(self.hydranet.named_parameters())
        else:
            prefix = {"lane": "laneheader.", "det": "detectheader.", "seg": "segheader."}[phase]
            named = [(n, p) for n, p in self.hydranet.named_parameters() if n.startswith(prefix)]
        return GradReducer(named, world_size=self.world, skip=unused_parameters(self.hydranet), payload_dtype=self._grad_payload,
                           force_collectives=self._force_distribute)

    def set_phase(self, phase: str):
        """One branch of main()'s schedule (train.py:462-505): the optimizer's first param group holds the parameters of the whole model
        ("joint") or of ONE head; Adam's per-parameter state is kept across phases, exactly as swapping `param_groups[0]['params']` does.
        In a head-only phase only that head's parameters can change, so the HIP path does not run the backward of the backbone, the neck
        and the other two heads at all (HydraNet.grad_scope: those parts run forward under no_grad -- BatchNorm running statistics still
        update, all six losses are still reported); the reference computes those gradients and throws them away."""
        assert phase in PHASES
        if self._phase_module(phase) is None:
            raise ValueError("phase %r needs the %s head (cfgs['train'])" % (phase, phase))
        self.optimizer.param_groups[0]["params"] = list(self._phase_module(phase).parameters())
        if phase != self.phase:
            self.hydranet.zero_grad(set_to_none=True)               # gradients of parameters leaving the group must not linger (or feed Adam later)
            self._cap = None                                         # a captured step belongs to one phase
            self._eager_iters = 0
            if self.use_distribute:
                self.reducer.remove()
                if phase not in self._reducers:
                    self._reducers[phase] = self._make_reducer(phase)
                else:
                    self._reducers[phase].arm()
                self.reducer = self._reducers[phase]
        self.phase = phase
        self.hydranet.grad_scope = None if phase == "joint" else phase

    def cal_total_loss(self, loss_dict: Dict[str, torch.Tensor]):
        """train.py:192-203: the weighted sum of the task losses (same weights, same association order); on the device it is one launch
        (HydraNet.total_loss -> ops.WeightedLossSum)"""
        return self.hydranet.total_loss(loss_dict)

    def to_gpu(self, batch_data: dict) -> dict:
        """train.py:228-239"""
        batch_data["image"] = batch_data["image"].to(self.device).float()
        if self.train_lane:
            batch_data["gt_loc"] = batch_data["gt_loc"].to(self.device).float()
            batch_data["gt_cls"] = batch_data["gt_cls"].to(self.device).float()
        if self.train_seg:
            batch_data["gt_seg"] = batch_data["gt_seg"].to(self.device).float()
        if self.train_detect:
            batch_data["gt_det"] = batch_data["gt_det"].to(self.device).float()
        return batch_data

    @staticmethod
    def _batch_sig(batch_data: dict):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in batch_data.items() if isinstance(v, torch.Tensor) and v.is_cuda)

    def _captured_fwd_bwd(self, batch_data: dict) -> Dict[str, torch.Tensor]:
        """forward + loss + backward (+ the gradient exchange) as one hipGraph replay (built on first use for this batch shape)"""
        keys = [k for k, v in batch_data.items() if isinstance(v, torch.Tensor) and v.is_cuda]
        sig = self._batch_sig(batch_data)
        if self._cap is None or self._cap[0] != sig:
            static = {k: batch_data[k].clone() for k in keys}
            net = self.hydranet
            guard, net.check_finite = net.check_finite, False        # the divergence guard reads the loss on the host: after the replay

            def fwd_bwd():
                outputs = net(static["image"])
                loss_dict = net.cal_loss(outputs, static)
                loss_dict["total_loss"] = self.cal_total_loss(loss_dict)
                loss_dict["total_loss"].backward(self._one)
                return loss_dict
            # parameter gradients become tensors of the graph's pool (stored by the captured backward, rewritten by every replay)
            zero = lambda: self.optimizer.zero_grad(set_to_none=True)
            try:
                red = self.reducer
                in_graph = red is not None and red.active and red.on_gpu and dist.get_backend(red.group) == "nccl"
                if in_graph:
                    # (no extra warm-up: the two eager iterations ran on self._stream with the hooks armed)
                    graph, loss_dict = capture_exchange_step(red, fwd_bwd, zero, self._stream, warmup=0)
                else:
                    if red is not None:
                        red.remove()                                 # no hooks inside this capture: the exchange follows every replay
                    zero()
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    # thread-local capture mode: helper threads (the autograd engine's allocator calls, an RCCL watchdog) must not invalidate it
                    with torch.cuda.graph(graph, capture_error_mode="thread_local", **({"stream": self._stream} if self._stream is not None else {})):
                        loss_dict = fwd_bwd()
                    if red is not None:
                        red.bind_static_grads()                      # reduce_now() gathers what each replay leaves in these tensors
            finally:
                net.check_finite = guard
            self._cap = (sig, graph, static, loss_dict, in_graph)
        _, graph, static, loss_dict, in_graph = self._cap
        for k in static:
            static[k].copy_(batch_data[k])
        graph.replay()
        if self.reducer is not None and not in_graph:
            self.reducer.reduce_now()
        if self.hydranet.check_finite:
            for name, v in loss_dict.items():
                self.hydranet._guard(v, "cal %s diverge!" % name, allow_zero=name.startswith("loss_det"))
        return {k: v.detach().clone() for k, v in loss_dict.items()}      # the static tensors are rewritten by the next replay

    def train_step(self, batch_data: dict) -> Dict[str, torch.Tensor]:
        """one iteration of train.py:243-267: forward, multitask loss, backward (gradient exchange overlapped), Adam step, LR step"""
        if self._stream is None:
            return self._train_step(batch_data)
        # data-parallel + capture_step: every iteration (eager or replayed) runs on the trainer's own stream, the one the capture uses
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = self._train_step(batch_data)
        cur.wait_stream(self._stream)
        return out

    def _train_step(self, batch_data: dict) -> Dict[str, torch.Tensor]:
        batch_data = self.to_gpu(batch_data)
        if self.capture_step and self._eager_iters >= 2 and self._cap is not None and self._cap[0] != self._batch_sig(batch_data):
            # A different batch shape than the captured one (the loaders' short last batch: drop_last=False, model/train.py:71,81).  The step is
            # NOT captured again on the spot (with the exchange in the graph that would be a second capture of RCCL collectives with no
            # warm-up at the new shape): this iteration and the next run on the eager path (hooks armed), the one after that
            # captures at the shape it sees.
            self._eager_iters = 0
        if self.capture_step and self._eager_iters >= 2:
            loss_dict = self._captured_fwd_bwd(batch_data)
            self.optimizer.step()
            self.scheduler.step()
            return loss_dict
        self._eager_iters += 1
        if self._cap is not None:              # back on the eager path after captured steps: gradients must not stay in the graph's pool
            self.optimizer.zero_grad(set_to_none=True)
            self._cap = None
        if self.reducer is not None:
            self.reducer.arm()                 # (a captured step removed the hooks; no-op while they are registered)
        outputs = self.hydranet(batch_data["image"])
        loss_dict = self.hydranet.cal_loss(outputs, batch_data)
        loss_total = self.cal_total_loss(loss_dict)
        loss_dict.update({"total_loss": loss_total})
        # gradients accumulate straight into the exchange buckets after the first step: zero them in place instead of dropping them
        self.optimizer.zero_grad(set_to_none=self.reducer is None)
        loss_total.backward(self._one)
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        # detached: the losses are for logging; a caller that keeps them must not keep this iteration's autograd nodes alive (stale
        # AccumulateGrad nodes bound to another stream break a later capture)
        return {k: v.detach() for k, v in loss_dict.items()}

    def train_one_epoch(self, epoch: int):
        self.hydranet.train()
        for iter_idx, batch_data in enumerate(self.trainloader):
            loss_dict = self.train_step(batch_data)
            if iter_idx % self.print_interval == 0:
                if self.device.type == "cuda":
                    K_xstage_assert_ok(self.device)          # (a replayed step cannot check its persistent launches itself: ops/xstage.py)
                if self.rank == 0:
                    self.print_loss_info(loss_dict, epoch, iter_idx)

    def print_loss_info(self, loss_dict, epoch, batch_idx, mode="train"):
        lr = self.optimizer.param_groups[0]["lr"]
        print("%s Epoch [%i|%i] Iter [%i] Lr %.5f  " % (mode.upper(), epoch, self.epoch, batch_idx, lr) +
              "  ".join("%s %.3f" % (k, float(v.detach())) for k, v in loss_dict.items()))

    @torch.no_grad()
    def valid(self, epoch: int = 0, eval_dir: Optional[str] = None, lane_coder=None, det_conf_thres: float = 0.3, det_iou_thres: float = 0.3):
        """train.py:271-438 without the third-party evaluators: eval-mode forward + the six losses per batch, streaming mIoU on the device
        (train.py:293-306), detection results through the device post-process in COCO-json form (train.py:308-364; written to
        `eval_dir`/val_bbox_results.json like train.py:416-421 -- the file COCOeval reads), lane decode + NMS on the device and the
        prediction json of LaneHeader.scale_to_org (train.py:366-395) when a `lane_coder` (LaneCodec) is given, and -- when the batches carry
        the ground-truth lanes of train.py:393 as `gt_lane_json` (one {"Lines": [...], "Labels": [...]} dict per image) -- the lane F1 of
        train.py:188,397,433 (LaneMetric, f1_measure, IoU 0.5, width 30, score threshold 0.5).  COCOeval needs pycocotools (out of scope).
        Deliberate deviation: train.py:397 calls `self.lane_metric(output=lane_result)` inside the batch loop with the CUMULATIVE list, so the
        reference counts image k of a validation run (number of batches - batch index of k) times, and never resets the evaluator between
        epochs; here every image is scored exactly once per valid() call (for a single-batch validation the two agree).
        Returns the per-class IoU tensor; everything else is left in self.last_valid."""
        from .coco_json import detections_to_coco, write_results
        from .lane_metric import LaneMetric
        lane_metric = LaneMetric(method="f1_measure", iou_thresh=0.5, lane_width=30, thresh_list=[0.5])           # train.py:188
        lane_pairs = []
        net = self.hydranet
        net.eval()
        if self.train_seg:
            self.metric_evaluator_iou = IntersectionOverUnion(n_classes=self.metric_evaluator_iou.n_classes, device=self.device)
        detect_result, lane_result, losses = [], [], []
        net_w, net_h = net.net_input_width, net.net_input_height
        for iter_idx, batch_data in enumerate(self.validloader):
            batch_data = self.to_gpu(batch_data)
            inputs = batch_data["image"]
            n = inputs.shape[0]
            outputs = net(inputs)
            have_gt = all(k in batch_data for k, on in (("gt_seg", self.train_seg), ("gt_det", self.train_detect), ("gt_cls", self.train_lane),
                                                         ("gt_loc", self.train_lane)) if on)
            if have_gt:
                loss_dict = net.cal_loss(outputs, batch_data)
                loss_dict["total_loss"] = self.cal_total_loss(loss_dict)
                losses.append({k: float(v) for k, v in loss_dict.items()})
                if self.rank == 0 and iter_idx % self.print_interval == 0:
                    self.print_loss_info(loss_dict, epoch, iter_idx, mode="valid")
            shapes = batch_data.get("src_image_shape") or [{"width": net_w, "height": net_h}] * n
            if self.train_seg and "gt_seg" in batch_data:
                from . import ops as K
                self.metric_evaluator_iou.update(K.argmax_channels(outputs["seg"]), batch_data["gt_seg"])
            if self.train_detect:
                d = outputs["detection"]
                preds = net.detectheader.decode(inputs, d["regression"], d["classification"], d["anchors"], conf_thres=det_conf_thres,
                                                iou_thres=det_iou_thres)
                metas = [[net_w, net_h, sh["width"], sh["height"], 0, 0] for sh in shapes]
                preds = net.detectheader.invert_affine(metas, preds)                                       # train.py:334-336
                detect_result += detections_to_coco(preds, iter_idx * self.cfgs["train"].get("batch_size_valid", n) + 1)
            if self.train_lane and lane_coder is not None:
                l = self.cfgs["lane"]
                lanes = net.laneheader.decode_batch(outputs["lane"]["predict_cls"], outputs["lane"]["predict_loc"], lane_coder,
                                                    l.get("conf_thres", 0.5), l.get("nms_thres", 100), False)
                gts = batch_data.get("gt_lane_json")
                for i, (ln, sh) in enumerate(zip(lanes, shapes)):
                    pj = net.laneheader.scale_to_org(ln, net_w, net_h, sh["width"], sh["height"])
                    lane_result.append(dict(pr_result={**pj, **dict(Shape=sh)}))
                    if gts is not None:
                        lane_pairs.append(dict(pr_result=lane_result[-1]["pr_result"], gt_result={**gts[i], **dict(Shape=sh)}))
        net.train()
        scores = self.metric_evaluator_iou.compute() if self.train_seg else None
        path = write_results(detect_result, eval_dir) if (eval_dir and self.train_detect and self.rank == 0) else None
        lane_f1 = None
        if lane_pairs:
            lane_metric(output=lane_pairs)                                                                         # train.py:397
            lane_f1 = lane_metric.summary()                                                                        # train.py:433
            if self.rank == 0:
                print("=========================== metric lane %i ===========================" % epoch)
                print(lane_f1)
        self.last_valid = dict(losses=losses, detect_result=detect_result, detect_json=path, lane_result=lane_result, iou=scores, lane_f1=lane_f1)
        return scores

    def save(self, path: str):
        """checkpoint in the reference's format: a DDP-wrapped module's state_dict carries "module." prefixes (train.py:437)"""
        if self.rank != 0:
            return
        sd = self.hydranet.state_dict()
        if self.use_distribute:
            sd = {"module." + k: v for k, v in sd.items()}
        torch.save(sd, path)
