"""Data-parallel gradient exchange for HydraNet: one process per GPU, bucketed sum-all-reduce over RCCL (torch.distributed backend
"nccl" on ROCm) issued from autograd hooks on a side HIP stream so it overlaps the rest of backward; gradients are averaged over ranks.

Replaces the reference's single-rank torch DistributedDataParallel(find_unused_parameters=True) (model/train.py:130-137):
  * buckets follow reverse execution order (heads -> neck -> backbone stage 4 .. stem), ~25 MiB each like DDP's default, and live
    in flat fp32 buffers; after the exchange each parameter's .grad is a view into its bucket (no copy back),
  * parameters that never receive a gradient (neck.bifpn.0.p5_to_p6.* in the 5-stage cfg) are excluded by NAME identically on every
    rank, which replaces DDP's per-step used-parameter bitmap all-reduce,
  * BatchNorm running statistics stay per-replica during training (the reference has no SyncBN); broadcast_buffers() reproduces DDP's
    rank-0 broadcast when a checkpoint is written.
xGMI note: 8 MI355X are fully meshed with point-to-point links, so the exchange is per-link bound; a few large buckets keep every link
busy while the backbone's backward (the longest part) is still running.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, named_params: Sequence, world_size: Optional[int] = None, bucket_bytes: int = 25 << 20,
                 skip: Iterable[str] = (), group=None, use_side_stream: Optional[bool] = None):
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        skip = set(skip)
        params = [(n, p) for n, p in named_params if p.requires_grad and n not in skip]
        params.reverse()                                   # registration order ~ forward order -> reverse ~ backward order
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for n, p in params:
            cur.append((n, p))
            cur_bytes += p.numel() * 4
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
        # RCCL averages in the collective itself (ncclAvg): no separate scaling pass over the 171 MB of gradients.  Falls back to
        # SUM + scale on the first failure and for backends without AVG (gloo).
        self._avg = self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for n, p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, plist):
        total = sum(p.numel() for _, p in plist)
        dev = plist[0][1].device
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        views, off = [], 0
        for _, p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append(dict(params=plist, flat=flat, views=views, pending=len(plist), work=None, event=None))

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.world == 1:
            return
        for (n, p), v in zip(b["params"], b["views"]):
            if p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v                                    # from now on autograd accumulates straight into the bucket
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self._allreduce_mean(b["flat"])
                done = torch.cuda.Event()
                done.record(self.stream)
            b["event"] = done
        else:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _allreduce_mean(self, flat):
        if self._avg:
            try:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
                return
            except Exception:                   # noqa: BLE001  (an RCCL build without ncclAvg)
                self._avg = False
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)

    def finish(self):
        """call after loss.backward(): waits for every bucket and re-arms the hooks for the next step."""
        for b in self.buckets:
            if self.world > 1:
                if b["pending"] != 0:
                    raise RuntimeError("a bucket never completed: parameters without gradient must be listed in `skip`: " +
                                       ", ".join(n for n, p in b["params"] if p.grad is None))
                if b["event"] is not None:
                    torch.cuda.current_stream().wait_event(b["event"])
                    b["event"] = None
                if b["work"] is not None:
                    b["work"].wait()
                    b["flat"].mul_(1.0 / self.world)
                    b["work"] = None
            b["pending"] = len(b["params"])

    def bind_static_grads(self):
        """hipGraph mode: a captured step always writes the SAME gradient tensors (graph-private memory).  Remember them, so that every
        reduce_now() first gathers their fresh contents into the buckets (after the first exchange .grad points at the bucket views, which
        a replay no longer touches)."""
        for b in self.buckets:
            if any(p.grad is None for _, p in b["params"]):
                raise RuntimeError("bind_static_grads() needs the gradients of a captured step: " +
                                   ", ".join(n for n, p in b["params"] if p.grad is None))
            b["src"] = [p.grad for _, p in b["params"]]

    def reduce_now(self):
        """non-overlapped variant (after a hipGraph replay of forward+backward): exchange every bucket, then finish()."""
        for b in self.buckets:
            src = b.get("src")
            if src is not None:
                torch._foreach_copy_(b["views"], src)
                for (n, p), v in zip(b["params"], b["views"]):
                    p.grad = v
            b["pending"] = 0
            self._launch(b)
        self.finish()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_state(module: torch.nn.Module, src: int = 0, group=None):
    """rank-0 broadcast of every parameter and buffer (what DDP does at construction, model/train.py:137)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


UNUSED_5STAGE = ("neck.bifpn.0.p5_to_p6.0.conv.weight", "neck.bifpn.0.p5_to_p6.0.conv.bias", "neck.bifpn.0.p5_to_p6.1.weight",
                 "neck.bifpn.0.p5_to_p6.1.bias")
