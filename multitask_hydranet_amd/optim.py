"""torch.optim.Adam's update as ONE HIP launch over all parameters (hn_adam_step).

The reference trains with `torch.optim.Adam(params, lr, weight_decay=wd)` (model/train.py:147).  On this model that is 693 parameter
tensors: the foreach implementation issues ~10 multi-tensor launches and takes 3.9 ms per step on MI355X, 17 % on top of the 22.5 ms
forward + loss + backward.  One pass over (p, g, m, v) moves 1.2 GB: ~0.35 ms.  Same update rule, same state layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter -- state_dict() / load_state_dict() are interchangeable with torch.optim.Adam's), same operation order (so it tracks torch's result to
the last bit or two of fp32); LR schedulers work on `param_groups[i]["lr"]` as usual.  fp32 CUDA parameters only; not amsgrad / maximize.
"""
from __future__ import annotations

import weakref

import torch

from ._lib import lib
from .ops.core import bump_mutation_epoch, mutation_cells


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or lr < 0.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}                # (group index, step value) -> (pointer signature, jobs, block_job, blocks)
        self._fast = {}                 # group index -> (gradient tensors of the last step, shared step scalar, plan)

    def state_dict(self):
        """torch.optim.Adam's layout.  Internally all parameters of a cohort share ONE host `step` scalar (one increment per step instead of
        693); torch.optim.Adam bumps every parameter's `step` separately, so a shared tensor would advance by #params per step once loaded
        there: the exported state gives every parameter its own copy."""
        sd = super().state_dict()
        sd["state"] = {k: ({**v, "step": v["step"].clone()} if isinstance(v.get("step"), torch.Tensor) else dict(v)) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """the cached job tables hold raw exp_avg / exp_avg_sq pointers of the state they were built from: drop them with it"""
        super().load_state_dict(state_dict)
        self._plans.clear()
        self._fast.clear()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plans, self._fast = {}, {}

    def _plan(self, key, ps):
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        pl = self._plans.get(key)
        if pl is not None and pl[0] == sig:
            return pl
        rows, owner, blk = [], [], 0
        for i, p in enumerate(ps):
            st = self.state[p]
            nb = (p.numel() + 1023) // 1024
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk])
            owner += [i] * nb
            blk += nb
        dev = ps[0].device
        pl = (sig, torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(owner, dtype=torch.int32).to(dev), blk)
        self._plans[key] = pl
        return pl

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # hn_adam_step writes the parameters through raw pointers: eval-mode caches keyed on `_version` are stale -- those of the modules that own
        # these parameters (the owner cells are collected once per parameter list, not per step)
        sig = tuple(len(g["params"]) for g in self.param_groups)
        if getattr(self, "_mut_sig", None) != sig:
            self._mut_sig, self._mut_cells = sig, mutation_cells([p for g in self.param_groups for p in g["params"]])
        bump_mutation_epoch(self._mut_cells)
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            fast = self._fast.get(gi)
            if fast is not None and len(fast[0]) == len(group["params"]) and all((p.grad is None) if g is None else (p.grad is g()) for p, g in zip(group["params"], fast[0])):
                # the same gradient tensors as last time (a captured step rewrites them in place): no per-parameter work on the host
                _, step_t, (_, jobs, owner, blocks) = fast
                step_t += 1
                lib().call("hn_adam_step", jobs.data_ptr(), owner.data_ptr(), blocks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                           float(group["weight_decay"]), int(step_t))
                continue
            by_step = {}
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32 and
                        p.grad.is_contiguous() and not p.grad.is_sparse):
                    raise RuntimeError("multitask_hydranet_amd.optim.Adam: fp32 contiguous CUDA parameters / gradients only")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)                      # host scalar, as torch.optim.Adam keeps it
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                by_step.setdefault(int(st["step"]), []).append(p)
            self._fast.pop(gi, None)
            for t, ps in by_step.items():
                shared = torch.full((), float(t + 1), dtype=torch.float32)                 # one host scalar for the whole cohort
                for p in ps:
                    self.state[p]["step"] = shared
                plan = self._plan((gi, len(by_step) > 1 and t), ps)
                lib().call("hn_adam_step", plan[1].data_ptr(), plan[2].data_ptr(), plan[3], float(group["lr"]), float(b1), float(b2),
                           float(group["eps"]), float(group["weight_decay"]), t + 1)
                if len(by_step) == 1:
                    # weak references: never keep a dropped gradient alive (its address could not be reused by the next backward)
                    self._fast[gi] = ([None if p.grad is None else weakref.ref(p.grad) for p in group["params"]], shared, plan)
        return loss
