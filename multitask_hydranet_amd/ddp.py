"""Data-parallel gradient exchange for HydraNet: one process per GPU, bucketed all-reduce over RCCL (torch.distributed backend "nccl" on
ROCm) issued from autograd hooks on a side HIP stream so that it overlaps the rest of backward; gradients are averaged over ranks.

Replaces the reference's single-rank torch DistributedDataParallel(find_unused_parameters=True) (model/train.py:130-137):
  * buckets follow reverse execution order (heads -> neck -> backbone stage 4 .. stem), ~25 MiB each like DDP's default, and live in flat
    buffers (fp32, or bf16 for half the xGMI payload); after the exchange each parameter's .grad is a view into its bucket;
  * parameters that never receive a gradient (neck.bifpn.0.p5_to_p6.* in the 5-stage cfg) are excluded by NAME identically on every rank,
    which replaces DDP's per-step used-parameter bitmap all-reduce;
  * BatchNorm running statistics stay per-replica during training (the reference has no SyncBN); broadcast_state() reproduces DDP's rank-0
    broadcast at construction.
Three ways to run a step:
  1. eager: hooks fire during loss.backward(); finish() joins.                              (overlapped, one launch per kernel)
  2. captured: the SAME hooks fire while the step is being captured into a hipGraph (torch.cuda.graph).  By default each completed
     bucket's gather + all-reduce are captured IN LINE on the capture stream (a synchronous torch.distributed collective is enqueued on
     the current stream), so the graph stays one linear chain: on this runtime a hipGraph with ANY fork/join replays every node ~1.3 us
     slower (tools/graph_branch_probe.py: 1200 dependent tiny kernels 1.9 ms linear, 3.4 ms with one fork; the training step: +1.1 ms =
     5 %), more than the exposed all-reduce time it would hide.  The capture must run on the stream the warm-up steps ran on
     (torch.cuda.graph(g, stream=warmup_stream)): the hooks keep the warm-up's AccumulateGrad nodes alive and autograd runs them -- and
     therefore the exchange -- on the stream they were created on; on another stream the exchange becomes a fork of the graph anyway
     (measured: two 4-element kernels launched from the hooks on the wrong stream cost 0.85 ms per step).  World size 1, big cfg, batch
     16: 20.70 ms without the exchange, 21.01 ms with it (+1.5 %: RCCL's one-rank pre-multiplied-sum kernel over the 171 MB payload,
     0.25 ms, and a 15 us gather of the 15 % of the elements that the backward kernels do not write into the buckets themselves).
     graph_overlap=True (HN_DDP_GRAPH_OVERLAP=1) keeps the forked form: each completed bucket forks the side stream off the capture
     stream with an event, join_capture() joins it back, a replay runs backward and the all-reduces concurrently.
     adopt_bucket_grads() then points .grad at the averaged buckets.
  3. reduce_now(): non-overlapped exchange after a replay of a graph that holds no collectives (fallback if RCCL capture is unavailable).
xGMI note: 8 MI355X are fully meshed with point-to-point links, so a ring all-reduce is per-link bound; a few large buckets keep every
link busy while the backbone's backward (the longest part) is still running.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from ._lib import policy


def _agree_on_avg(group, device) -> bool:
    """ncclAvg availability decided ONCE, identically on every rank: every rank tries a 1-element AVG all-reduce, then the success flags
    are summed (a rank-local try/except around a real exchange could leave ranks on different collectives)."""
    ok = 1.0
    try:
        t = torch.ones(1, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        torch.cuda.synchronize(device)
        ok = 1.0 if abs(float(t) - 1.0) < 1e-6 else 0.0
    except Exception:       # noqa: BLE001  (an RCCL build without ncclAvg)
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
    return float(flag) == float(dist.get_world_size(group))


class GradReducer:
    def __init__(self, named_params: Sequence, world_size: Optional[int] = None, bucket_bytes: int = 25 << 20,
                 skip: Iterable[str] = (), group=None, use_side_stream: Optional[bool] = None, payload_dtype: torch.dtype = torch.float32,
                 force_collectives: bool = False, graph_overlap: Optional[bool] = None):
        """force_collectives: issue the collectives even at world size 1 (exercises RCCL init, ncclAvg and the side-stream / capture path on
        a single GPU).  graph_overlap: inside a hipGraph capture, fork the exchange onto the side stream (True) or keep it in line on the
        capture stream (False, default: a linear graph replays faster than a forked one by more than the overlap hides)."""
        if graph_overlap is None:
            graph_overlap = policy("HN_DDP_GRAPH_OVERLAP", "0") == "1"
        self.graph_overlap = bool(graph_overlap)
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.payload_dtype = payload_dtype
        skip = set(skip)
        params = [(n, p) for n, p in named_params if p.requires_grad and n not in skip]
        params.reverse()                                   # registration order ~ forward order -> reverse ~ backward order
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        esize = torch.empty((), dtype=payload_dtype).element_size()
        for n, p in params:
            cur.append((n, p))
            cur_bytes += p.numel() * esize
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        dev = params[0][1].device if params else torch.device("cpu")
        self.on_gpu = dev.type == "cuda"
        if use_side_stream is None:
            use_side_stream = self.on_gpu
        self.stream = torch.cuda.Stream(device=dev) if (self.on_gpu and use_side_stream) else None
        # RCCL averages inside the collective (ncclAvg): no separate scaling pass over the gradients.  Decided once, on every rank alike.
        self._avg = bool(self.active and self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "nccl" and
                         _agree_on_avg(group, dev))
        self.captured = False
        self.capture_stream = None              # set_capture_stream(): where the in-line captured exchange is enqueued
        self._pending_uploads = []              # gather tables filled during a capture, uploaded by adopt_bucket_grads()
        self._hooks = []
        self.arm()

    # ------------------------------------------------------------------------------------------------------------------------------
    def _close(self, plist):
        # every parameter's slot starts on a 16-byte boundary of the flat buffer (fusion weights of 2-3 elements, 9-tap depthwise filters ...
        # would otherwise leave everything behind them at odd offsets: scalar instead of float4 accesses in the Adam / gather / landing
        # kernels that work on the bucket views).  The padding elements stay zero and ride along in the all-reduce.
        esize = torch.empty((), dtype=self.payload_dtype).element_size()
        al = max(1, 16 // esize)
        offs, off = [], 0
        for _, p in plist:
            offs.append(off)
            off += (p.numel() + al - 1) // al * al
        total = off
        dev = plist[0][1].device
        flat = torch.zeros(total, device=dev, dtype=self.payload_dtype)
        views = [flat[o:o + p.numel()].view_as(p) for o, (_, p) in zip(offs, plist)]
        # fp32 payload: .grad becomes a view of the bucket; reduced payload: a separate fp32 landing buffer receives the averaged values
        land = None
        if self.payload_dtype != torch.float32:
            lflat = torch.zeros(total, device=dev, dtype=torch.float32)
            land = [lflat[o:o + p.numel()].view_as(p) for o, (_, p) in zip(offs, plist)]
        self.buckets.append(dict(params=plist, flat=flat, views=views, land=land, pending=len(plist), work=None, event=None, src=None, keep=None,
                                 offs=offs))

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            if b["pending"] <= 0 or b["work"] is not None or b["event"] is not None:
                raise RuntimeError("a gradient hook fired for a bucket whose exchange is already in flight: call finish() after every "
                                   "backward (gradient accumulation over several backward passes is not supported by this reducer)")
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _copy_many(self, b, key, dsts, srcs, kind):
        """dsts[i] <- srcs[i] for all i in ONE launch on the device (hn_copy_many; the job table is cached while the pointers stay the same --
        always, for the static tensors of a captured step); torch._foreach_copy_ elsewhere (CPU tensors of the gloo tests, mixed layouts)"""
        ok = self.on_gpu and all(d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.numel() == s.numel()
                                 for d, s in zip(dsts, srcs))
        if not ok:
            torch._foreach_copy_(dsts, srcs)
            return
        sig = tuple((s.data_ptr(), d.data_ptr()) for d, s in zip(dsts, srcs))
        nblk = sum((d.numel() + 1023) // 1024 for d in dsts)
        plan = b.get(key)
        if plan is None or plan["cap"][0] < len(dsts) or plan["cap"][1] < nblk:
            if torch.cuda.is_current_stream_capturing():
                torch._foreach_copy_(dsts, srcs)            # (pinned allocations are not allowed inside a capture: the eager warm-up makes them)
                return
            dev = dsts[0].device
            plan = dict(cap=(len(dsts), nblk), sig=None,
                        hrows=torch.empty((len(dsts), 4), dtype=torch.int64).pin_memory(), howner=torch.empty((nblk,), dtype=torch.int32).pin_memory(),
                        drows=torch.empty((len(dsts), 4), dtype=torch.int64, device=dev), downer=torch.empty((nblk,), dtype=torch.int32, device=dev))
            b[key] = plan
        if plan["sig"] != sig:
            rows, owner, blk = [], [], 0
            for i, (d, s) in enumerate(zip(dsts, srcs)):
                nb = (d.numel() + 1023) // 1024
                rows.append([s.data_ptr(), d.data_ptr(), d.numel(), blk])
                owner += [i] * nb
                blk += nb
            # Inside a capture (the graph-private gradient tensors are only known then) the table is filled on the host and uploaded ONCE, after
            # the capture and before the first replay (adopt_bucket_grads): the captured gather only reads the device table.  (Uploads as
            # memcpy nodes of the graph cost ~45 us each per replay: the copy engine's hand-offs stall the otherwise linear kernel chain.)
            if torch.cuda.is_current_stream_capturing():
                plan["hrows"][:len(rows)] = torch.tensor(rows, dtype=torch.int64)
                plan["howner"][:blk] = torch.tensor(owner, dtype=torch.int32)
                self._pending_uploads.append(plan)
            else:                                           # eager: blocking uploads (the pinned tables may still feed an earlier copy)
                plan["drows"][:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
                plan["downer"][:blk].copy_(torch.tensor(owner, dtype=torch.int32))
            plan["sig"], plan["blk"] = sig, blk
        from ._lib import lib
        lib().call("hn_copy_many", plan["drows"].data_ptr(), plan["downer"].data_ptr(), plan["blk"], kind)

    def _gather(self, b):
        """bring the bucket's gradients into its flat payload buffer (one multi-tensor launch; converts when the payload is bf16)"""
        src = b["src"] if b["src"] is not None else [p.grad for _, p in b["params"]]
        need = [(v, s) for v, s in zip(b["views"], src) if s.data_ptr() != v.data_ptr()]
        b["direct_elems"] = sum(v.numel() for v in b["views"]) - sum(v.numel() for v, _ in need)     # produced in place by the HIP backward (ops.grad_out)
        if need:
            dsts, srcs = [v for v, _ in need], [s for _, s in need]
            same = all(d.dtype == s.dtype for d, s in need)
            if same and dsts[0].dtype == torch.float32:
                self._copy_many(b, "plan_gather", dsts, srcs, 0)
            elif all(d.dtype == torch.bfloat16 and s.dtype == torch.float32 for d, s in need):
                self._copy_many(b, "plan_gather", dsts, srcs, 1)
            else:
                torch._foreach_copy_(dsts, srcs)

    def _scatter(self, b):
        """after the exchange: .grad = averaged values (a view of the bucket, or of the fp32 landing buffer for a reduced payload)"""
        if b["land"] is not None:
            self._copy_many(b, "plan_land", b["land"], b["views"], 2)
        dst = b["land"] if b["land"] is not None else b["views"]
        for (n, p), v in zip(b["params"], dst):
            p.grad = v

    def _allreduce_mean(self, flat):
        if self._avg:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                flat.mul_(1.0 / self.world)

    def _launch(self, b):
        if not self.active:
            return
        capturing = self.on_gpu and torch.cuda.is_current_stream_capturing()
        if capturing:
            b["src"] = [p.grad for _, p in b["params"]]        # graph-private tensors every replay rewrites
            self.captured = True
        elif self.stream is not None:
            # The gather below runs on the SIDE stream and reads the gradients autograd allocated on the main stream; _point() then rebinds
            # .grad to the bucket views, which would drop the last reference to them while the gather may still be queued behind the previous
            # bucket's all-reduce -- the caching allocator could hand their blocks to a later backward kernel.  Keep them until finish().
            b["keep"] = [p.grad for _, p in b["params"]]
        if capturing and not self.graph_overlap:
            # in line on the capture stream: gather, all-reduce (a synchronous collective runs on the current stream), landing copy.
            # The hook runs on the stream its AccumulateGrad node was created on.  It must BE the capture stream: AccumulateGrad may do work
            # of its own there (a deep copy of a gradient it cannot steal, `+=` for a second contribution), so an exchange enqueued on
            # another stream would read p.grad unordered behind that work -- and would be a fork of the graph besides.  The
            # caller keeps the two equal by capturing on the stream the warm-up steps ran on (bench.TrainRun._capture).
            cs = self.capture_stream
            if cs is not None and cs != torch.cuda.current_stream():
                raise RuntimeError("GradReducer: a gradient hook fired on %r while the step is being captured on %r; run the eager warm-up "
                                   "steps on the capture stream (torch.cuda.graph(g, stream=warmup_stream))" % (torch.cuda.current_stream(), cs))
            self._gather(b)
            self._allreduce_mean(b["flat"])
            if b["land"] is not None:
                self._copy_many(b, "plan_land", b["land"], b["views"], 2)
        elif self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)                          # fork: the side stream joins the capture through this dependency
            with torch.cuda.stream(self.stream):
                self._gather(b)
                self._allreduce_mean(b["flat"])
                if b["land"] is not None:
                    self._copy_many(b, "plan_land", b["land"], b["views"], 2)
                done = torch.cuda.Event()
                done.record(self.stream)
            b["event"] = done
            if not capturing:
                self._point(b)
        else:
            self._gather(b)
            self._point(b)
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _point(self, b):
        dst = b["land"] if b["land"] is not None else b["views"]
        for (n, p), v in zip(b["params"], dst):
            p.grad = v                                          # from now on autograd accumulates straight into the bucket

    # ------------------------------------------------------------------------------------------------------------------------------
    def finish(self):
        """call after loss.backward(): waits for every bucket and re-arms the hooks for the next step."""
        for b in self.buckets:
            if self.active:
                if b["pending"] != 0:
                    raise RuntimeError("a bucket never completed: parameters without gradient must be listed in `skip`: " +
                                       ", ".join(n for n, p in b["params"] if p.grad is None))
                if b["event"] is not None:
                    torch.cuda.current_stream().wait_event(b["event"])
                    b["event"] = None
                b["keep"] = None                                # the main stream is now ordered behind the gather: safe to recycle
                if b["work"] is not None:
                    b["work"].wait()
                    if self.world > 1:
                        b["flat"].mul_(1.0 / self.world)
                    if b["land"] is not None:
                        self._copy_many(b, "plan_land", b["land"], b["views"], 2)
                    b["work"] = None
            b["pending"] = len(b["params"])

    def set_capture_stream(self, stream):
        """the stream a following torch.cuda.graph(...) capture runs on: the in-line exchange of a captured step is enqueued there"""
        self.capture_stream = stream

    def join_capture(self):
        """inside torch.cuda.graph(...), after loss.backward(): join the side stream back into the capture stream (every bucket's all-reduce
        becomes a branch of the graph that ends here) and re-arm."""
        self.finish()

    def adopt_bucket_grads(self):
        """after the capture and BEFORE the first replay: uploads the gather tables the capture filled, and points .grad of every exchanged
        parameter at the averaged values a replay leaves in the buckets"""
        for plan in self._pending_uploads:
            plan["drows"].copy_(plan["hrows"])
            plan["downer"].copy_(plan["howner"])
        self._pending_uploads = []
        for b in self.buckets:
            if b["src"] is not None:
                self._point(b)

    def bind_static_grads(self):
        """after-replay mode: a captured step always writes the SAME gradient tensors (graph-private memory).  Remember them, so that every
        reduce_now() first gathers their fresh contents into the buckets."""
        for b in self.buckets:
            if any(p.grad is None for _, p in b["params"]):
                raise RuntimeError("bind_static_grads() needs the gradients of a captured step: " +
                                   ", ".join(n for n, p in b["params"] if p.grad is None))
            b["src"] = [p.grad for _, p in b["params"]]

    def reduce_now(self):
        """non-overlapped variant (after a replay of a graph without collectives): exchange every bucket, then finish()."""
        for b in self.buckets:
            b["pending"] = 0
            self._launch(b)
            if self.stream is None or not self.active:
                pass
        self.finish()
        for b in self.buckets:
            if b["src"] is not None and self.active:
                self._point(b)

    def arm(self):
        """(re-)register the autograd hooks (construction does; remove() + arm() parks a reducer while another parameter set trains: the
        fine-tuning phases of HydraTrainer.set_phase)"""
        if self._hooks:
            return
        for bi, b in enumerate(self.buckets):
            b["pending"], b["work"], b["event"], b["keep"] = len(b["params"]), None, None, None
            b["src"] = None                 # (static gradient tensors of an earlier captured step: the hooks gather from .grad again)
            for (n, p), off in zip(b["params"], b["offs"]):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
                if self.active and self.payload_dtype == torch.float32:
                    # the HIP backward writes this parameter's gradient straight into its bucket slot where it can (ops.grad_out):
                    # the gather before the all-reduce then only moves what autograd produced elsewhere
                    p._hn_grad_slot = (b["flat"], off)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for b in self.buckets:
            for n, p in b["params"]:
                if hasattr(p, "_hn_grad_slot"):
                    del p._hn_grad_slot

    def direct_fraction(self) -> float:
        """share of the payload elements that the last step's backward wrote straight into the buckets (no gather copy)"""
        tot = sum(v.numel() for b in self.buckets for v in b["views"])
        return sum(b.get("direct_elems", 0) for b in self.buckets) / max(tot, 1)

    def describe(self, after_replay: bool = False) -> str:
        how = ("captured inside the hipGraph on a side stream, overlapped with backward" if self.graph_overlap else
               "captured in line inside the hipGraph (linear graph: no fork/join)") if self.captured else \
              ("after each hipGraph replay (reduce_now, not overlapped)" if after_replay else
               ("on a side stream from autograd hooks, overlapped with backward" if self.stream is not None else "after backward"))
        return "%d buckets (%s payload, %s, %.0f%% of the elements written in place by the backward kernels) %s" % (
            len(self.buckets), str(self.payload_dtype).replace("torch.", ""), "ncclAvg" if self._avg else "sum + scale",
            100.0 * self.direct_fraction(), how)


def settle_collectives(device=None):
    """Before a capture that follows eager collectives: wait until they are complete on the device, then give ProcessGroupNCCL's watchdog
    thread time to reap their work objects (it polls on a fixed ~100 ms cadence and offers no handle to wait on).  A work object it still
    polls with hipEventQuery while the stream its end event was recorded on is being captured aborts the process (seen once in ~40 runs
    before this wait existed).  Collectives issued DURING a capture are never handed to the watchdog, so this is the only window."""
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        # > 3 watchdog rounds by default; HN_SETTLE_SECONDS lengthens it on a loaded host (the work objects themselves were waited for by
        # GradReducer.finish(); this wait only covers the watchdog's own bookkeeping, which offers no handle)
        time.sleep(max(0.35, float(os.environ.get("HN_SETTLE_SECONDS", "0.35"))))


def capture_exchange_step(reducer: "GradReducer", fwd_bwd: Callable[[], torch.Tensor], zero_grad: Callable[[], None], stream,
                          warmup: int = 2):
    """One training iteration's forward + loss + backward WITH the gradient exchange as one hipGraph (the data-parallel form of
    model/train.py:243-267 that bench.py times and HydraTrainer(capture_step=True) runs at world size > 1; RCCL only -- other backends'
    collectives cannot be captured, see capture_plain_step):
      * `warmup` eager iterations and the capture run on ONE stream: the reducer's hooks keep the AccumulateGrad nodes of earlier iterations
        alive, and autograd runs such a node -- and its post-accumulate hook, i.e. the bucket's gather + all-reduce -- on the stream it was
        created on; on any other stream the exchange becomes a fork of the graph (a hipGraph with a fork replays every node slower);
      * gradients start from None inside the capture: the backward kernels then write into the bucket slots (ops.grad_out) or into
        graph-private tensors the captured gather reads;
      * after the capture the gather tables are uploaded and .grad points at the averaged buckets (adopt_bucket_grads); the hooks are
        removed (a replay needs none).
    -> (graph, fwd_bwd's return value: tensors of the graph's pool that every replay rewrites -- detach them before keeping them)."""
    reducer.arm()
    reducer.set_capture_stream(stream)
    cur = torch.cuda.current_stream()
    if cur != stream:
        stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            zero_grad()
            fwd_bwd()
            reducer.finish()
    if cur != stream:
        cur.wait_stream(stream)
    settle_collectives()
    zero_grad()
    g = torch.cuda.CUDAGraph()
    # thread-local capture mode: helper threads (the autograd engine's allocator calls, the RCCL watchdog's event queries) must not
    # invalidate the capture
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        out = fwd_bwd()
        reducer.join_capture()
    reducer.adopt_bucket_grads()
    reducer.remove()
    return g, out


def broadcast_state(module: torch.nn.Module, src: int = 0, group=None):
    """rank-0 broadcast of every parameter and buffer (what DDP does at construction, model/train.py:137)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


UNUSED_5STAGE = ("neck.bifpn.0.p5_to_p6.0.conv.weight", "neck.bifpn.0.p5_to_p6.0.conv.bias", "neck.bifpn.0.p5_to_p6.1.weight",
                 "neck.bifpn.0.p5_to_p6.1.bias")


def unused_parameters(net) -> tuple:
    """names of the parameters a HydraNet never gives a gradient (excluded from the exchange on every rank alike): the first BiFPN cell's
    p5_to_p6 reducer when the backbone has 5 stages (net/bifpn.py:162-165: the last stage IS P6); none in the 4-stage configurations"""
    return UNUSED_5STAGE if len(net.depths) == 5 else ()
