"""ops.core -- tensor helpers, the packed-weight cache (PackPlan), raw kernel wrappers (no autograd: k_gemm_nt, k_gemm_tn, the fused
BatchNorm passes ...) and the deferred parameter-gradient machinery (GradQueue, DeferredGrads).  See multitask_hydranet_amd/ops/__init__.py."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from .._lib import lib, policy

ACT_NONE, ACT_RELU, ACT_SWISH, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3, 4
GCONV_MFMA = True          # stride-1 grouped 3x3 convs as block-diagonal 64-channel MFMA tiles (False: VALU stencil kernels)
BF16 = torch.bfloat16
F32 = torch.float32



# --------------------------------------------------------------------------------------------------------------
# tensor helpers
# --------------------------------------------------------------------------------------------------------------
def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def ld(t: torch.Tensor) -> int:
    """row stride (elements) of an NHWC activation (or channel-slice view of one).  Size-1 dims carry no stride information."""
    assert t.dim() == 4 and (t.stride(3) == 1 or t.shape[3] == 1), (t.shape, t.stride())
    n, h, w, c = t.shape
    if w > 1:
        s = t.stride(2)
    elif h > 1:
        s = t.stride(1)
    elif n > 1:
        s = t.stride(0)
    else:
        s = c
    assert (w == 1 or h == 1 or t.stride(1) == w * s) and (n == 1 or h * w == 1 or t.stride(0) == h * w * s), (t.shape, t.stride())
    assert s % 8 == 0, (t.shape, t.stride())
    return s


def rows(t: torch.Tensor) -> int:
    return t.shape[0] * t.shape[1] * t.shape[2]


def zeros(shape, device, dtype=F32):
    """zero tensor written by a fill KERNEL (torch.zeros / zero_() become hipMemsetAsync -> memset nodes inside a captured hipGraph, whose
    ordering against the consumer kernel proved unreliable on this stack)"""
    return torch.full(tuple(shape), 0.0, device=device, dtype=dtype)


def new_act(n, h, w, c, device, dtype=BF16):
    return torch.empty((n, h, w, c), device=device, dtype=dtype)


def kp32(c: int) -> int:
    return (c + 31) // 32 * 32


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def dense(t: torch.Tensor) -> torch.Tensor:
    """make a gradient tensor usable by the kernels (NHWC bf16, unit channel stride, uniform row stride)."""
    if t.dtype != BF16:
        t = t.to(BF16)
    try:
        ld(t)
    except AssertionError:
        t = t.contiguous()
    return t


# --------------------------------------------------------------------------------------------------------------
# packed-weight cache: a weight is cast/packed once per optimizer step (keyed by storage + version counter)
# --------------------------------------------------------------------------------------------------------------
_PACK_CACHE = {}
# Raw-pointer writes (hn_adam_step on the parameters, the BatchNorm running statistics written by the training-mode kernels) do not bump
# torch's per-tensor version counters, so "nothing changed since" cannot be decided from `_version` alone: everything in this
# library that mutates a parameter or a running statistic behind autograd's back bumps an epoch, and every freshness test includes it.
# The epoch is scoped to the OWNER of the tensors: a module tags its parameters and buffers with its own cell
# (tag_mutation_owner), mutators that know which tensors they write bump those cells only, and a prepared (folded) module is not
# invalidated by another module's optimizer step or training forward.  Tensors nobody tagged fall back to the process-wide cell.
_MUTATION_EPOCH = [0]


def new_mutation_cell():
    return [0]


def tag_mutation_owner(tensors, cell):
    for t in tensors:
        t._hn_mut_cell = cell


def mutation_cells(tensors):
    """the distinct owner cells of `tensors` (the process-wide cell stands in for untagged ones): what a mutator of exactly these tensors bumps"""
    cells = {}
    for t in tensors:
        c = getattr(t, "_hn_mut_cell", None)
        c = _MUTATION_EPOCH if c is None else c
        cells[id(c)] = c
    return list(cells.values())


def bump_mutation_epoch(cells=None):
    """cells: mutation_cells(the tensors written); None: owner unknown, everything in the process is stale"""
    for c in ([_MUTATION_EPOCH] if cells is None else cells):
        c[0] += 1


def mutation_epoch(owner=None):
    """freshness key for the tensors of `owner` (a tagged tensor, a cell, or None = untagged)"""
    c = owner if isinstance(owner, list) else getattr(owner, "_hn_mut_cell", None)
    return (_MUTATION_EPOCH[0], -1 if c is None else c[0])


def clear_pack_cache():
    _PACK_CACHE.clear()


_PACK_LOG = None            # when a list: conv weights packed one by one are recorded here (HydraNet builds its PackPlan from it)


def _cached(key, w: torch.Tensor, make, meta=None):
    # the entry keeps a strong reference to the weight tensor, so its id() cannot be recycled while the entry lives
    k = (key, id(w))
    hit = _PACK_CACHE.get(k)
    if hit is not None and hit[0] is w and hit[1] == w._version:
        return hit[2]
    v = make()
    _PACK_CACHE[k] = (w, w._version, v)
    if _PACK_LOG is not None:
        _PACK_LOG.append((key, w, meta))
    return v


def start_pack_log():
    global _PACK_LOG
    _PACK_LOG = []


def stop_pack_log():
    global _PACK_LOG
    log, _PACK_LOG = _PACK_LOG, None
    return log


class PackPlan:
    """Every per-step weight pack of a model in TWO launches (the big cfg needed ~190 one by one): dense conv weights go through the tiled
    transposing kernel (hn_pack_weights_batched), everything else -- depthwise taps, grouped-conv stencil / block-diagonal operands,
    channel-slice and phase-form packs -- through one elementwise launch (hn_pack_small_batched).  Built from the pack log of a forward:
    entries (key, weight, meta) as recorded by _cached().  Owns persistent packed buffers; run() refreshes them and primes the pack cache
    so that the pack_*() helpers hit."""

    def __init__(self, log):
        seen, entries = set(), []
        for key, w, meta in log:
            if (key, id(w)) not in seen:
                seen.add((key, id(w)))
                entries.append((key, w, meta))
        self.entries = entries
        dev = entries[0][1].device
        self.values = []                                   # cache value per entry (tuple of tensors [+ bias bookkeeping for phase packs])
        dense_rows, dense_owner, dblk = [], [], 0
        small_rows, small_owner = [], []
        sblk = [0]

        def small(row, elements):
            nb = (elements + 255) // 256
            row[6] = sblk[0]
            small_rows.append(row + [0] * (16 - len(row)))
            small_owner.extend([len(small_rows) - 1] * nb)
            sblk[0] += nb

        for key, w, meta in entries:
            kind = key[0] if isinstance(key, tuple) else key
            if kind == "conv":
                cout, cin = w.shape[0], w.shape[1]
                taps = w.shape[2] * w.shape[3]
                assert taps in (1, 9)
                wp = torch.empty((cout, taps * kp32(cin)), device=dev, dtype=BF16)
                wt = torch.empty((cin, taps * kp32(cout)), device=dev, dtype=BF16)
                dense_rows.append([w.data_ptr(), wp.data_ptr(), wt.data_ptr(), cout, cin, taps, dblk, kp32(cin) // 32])
                nb = (kp32(cout) // 32) * (kp32(cin) // 32)               # one workgroup per 32 x 32 (cout, cin) tile
                dense_owner += [len(dense_rows) - 1] * nb
                dblk += nb
                self.values.append((wp, wt))
            elif kind == "dw":
                c = w.shape[0]
                wk, wf = torch.empty((9 * c,), device=dev, dtype=BF16), torch.empty((9 * c,), device=dev, dtype=BF16)
                small([w.data_ptr(), wk.data_ptr(), wf.data_ptr(), 0, 0, 1, 0, c], 9 * c)
                self.values.append((wk, wf))
            elif kind == "g":
                c, flip = w.shape[0], key[1]
                wk, wd = torch.empty((72 * c,), device=dev, dtype=BF16), torch.empty((72 * c,), device=dev, dtype=BF16)
                small([w.data_ptr(), wk.data_ptr(), wd.data_ptr(), 0, 0, 2, 0, c // 8, flip], 72 * c)
                self.values.append((wk, wd))
            elif kind == "gdiag":
                c = w.shape[0]
                # zero-filled ONCE: the batched kernel rewrites only the 8 x 8 diagonal blocks (1/8 of the operand) every step
                wk, wd = torch.zeros((c, 576), device=dev, dtype=BF16), torch.zeros((c, 576), device=dev, dtype=BF16)
                small([w.data_ptr(), wk.data_ptr(), wd.data_ptr(), 0, 0, 3, 0, c], 72 * c)
                self.values.append((wk, wd))
            elif kind in ("slice", "phase"):
                cout, cin_total, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                phase = 1 if kind == "phase" else 0
                ci0, cin = (0, key[1]) if phase else (key[1], key[2])
                coute = 4 * cout if phase else cout
                wp = torch.empty((coute, taps * kp32(cin)), device=dev, dtype=BF16)
                wt = torch.empty((cin, taps * kp32(coute)), device=dev, dtype=BF16) if (not phase or key[2]) else None
                b_eff = torch.empty((coute,), device=dev, dtype=F32) if phase else None
                bias = meta if phase else None
                small([w.data_ptr(), wp.data_ptr(), wt.data_ptr() if wt is not None else 0, b_eff.data_ptr() if phase else 0,
                       bias.data_ptr() if phase else 0, 4, 0, cout, cin_total, ci0, cin, taps, phase],
                      wp.numel() // 8 + (wt.numel() // 8 if wt is not None else 0) + (coute if phase else 0))   # eight K entries per thread
                self.values.append((wp, wt, b_eff, bias) if phase else (wp, wt))
            else:
                raise KeyError(key)
        self.dense_blocks, self.small_blocks = dblk, sblk[0]
        mk = lambda rows_, dt: torch.tensor(rows_, dtype=dt).to(dev) if rows_ else None
        self.dense_table, self.dense_owner = mk(dense_rows, torch.int64), mk(dense_owner, torch.int32)
        self.small_table, self.small_owner = mk(small_rows, torch.int64), mk(small_owner, torch.int32)
        self.n_dense = len(dense_rows)
        self.device = dev
        self._ran = None                                   # parameter versions at the last run()
        self.ptrs = [(w.data_ptr(), meta.data_ptr() if isinstance(meta, torch.Tensor) else 0) for _, w, meta in entries]

    @property
    def packs(self):
        """(wp, wt) of the dense conv weights, in log order (tests)"""
        return [v for (key, _, _), v in zip(self.entries, self.values) if key == "conv"]

    def valid(self):
        """the device job tables hold raw weight pointers: a parameter whose storage was swapped (`p.data = ...`, vector_to_parameters, a
        device move) invalidates the plan (HydraNet.forward rebuilds it)"""
        return all(w.data_ptr() == p_[0] and w.device == self.device and (not isinstance(meta, torch.Tensor) or meta.data_ptr() == p_[1])
                   for (_, w, meta), p_ in zip(self.entries, self.ptrs))

    def _versions(self):
        return (mutation_epoch(self.entries[0][1] if self.entries else None),
                [(w._version, meta._version if isinstance(meta, torch.Tensor) else 0) for _, w, meta in self.entries])

    def fresh(self):
        """the packed buffers still hold what run() made of the CURRENT parameter values (no in-place update since -- neither through
        autograd-visible ops (version counters) nor through this library's raw-pointer kernels (mutation epoch)) and the pack cache still
        points at them: an eval-mode forward may skip run() (serving: the weights are constants; inside a captured deploy forward the two
        pack launches then are not part of the graph at all)"""
        return self._ran == self._versions() and all(_PACK_CACHE.get((key, id(w)), (None,))[0] is w for key, w, _ in self.entries)

    def run(self):
        self._ran = self._versions()
        if self.dense_table is not None:
            lib().call("hn_pack_weights_batched", ptr(self.dense_table), self.n_dense, self.dense_blocks, ptr(self.dense_owner))
        if self.small_table is not None:
            lib().call("hn_pack_small_batched", ptr(self.small_table), ptr(self.small_owner), self.small_blocks)
        for (key, w, meta), v in zip(self.entries, self.values):
            if isinstance(key, tuple) and key[0] == "phase":
                v = (v[0], v[1], v[2], meta, meta._version)
            _PACK_CACHE[(key, id(w))] = (w, w._version, v)


def pack_conv_weight(w: torch.Tensor):
    """fp32 [Cout, Cin, kh, kw] -> (wp [Cout, taps*KP(Cin)], wt [Cin, taps*KP(Cout)]) bf16."""
    def make():
        cout, cin = w.shape[0], w.shape[1]
        taps = w.shape[2] * w.shape[3]
        wp = torch.empty((cout, taps * kp32(cin)), device=w.device, dtype=BF16)
        wt = torch.empty((cin, taps * kp32(cout)), device=w.device, dtype=BF16)
        lib().call("hn_pack_weight", ptr(w), ptr(wp), ptr(wt), cout, cin, taps)
        return wp, wt
    return _cached("conv", w, make)


def pack_conv_weight_slice(w: torch.Tensor, ci0: int, cin: int):
    """pack_conv_weight of the input-channel slice w[:, ci0:ci0+cin] without materialising the slice"""
    def make():
        cout, taps = w.shape[0], w.shape[2] * w.shape[3]
        wp = torch.empty((cout, taps * kp32(cin)), device=w.device, dtype=BF16)
        wt = torch.empty((cin, taps * kp32(cout)), device=w.device, dtype=BF16)
        lib().call("hn_pack_weight_ex", ptr(w), ptr(wp), ptr(wt), cout, w.shape[1], ci0, cin, taps, 0, None, None)
        return wp, wt
    return _cached(("slice", ci0, cin), w, make)


def pack_phase_weight(w: torch.Tensor, c0: int, bias: torch.Tensor, want_wt: bool = True):
    """phase-form effective weights of the first c0 input channels of a 3x3 conv over a nearest-x2 up-sampled map (SegConvUp / SegOutUp):
    (wp_eff [4k, 9*KP(c0)], wt_eff [c0, 9*KP(4k)] | None, b_eff [4k]) in one launch.  want_wt = False: the layer's data gradient does not
    run in phase form (decoder.1: the transposed operand alone was 9.4 M scattered-read elements of the per-step pack)."""
    key = ("phase", c0, 1 if want_wt else 0)

    def make():
        k = w.shape[0]
        wp = torch.empty((4 * k, 9 * kp32(c0)), device=w.device, dtype=BF16)
        wt = torch.empty((c0, 9 * kp32(4 * k)), device=w.device, dtype=BF16) if want_wt else None
        b_eff = torch.empty((4 * k,), device=w.device, dtype=F32)
        lib().call("hn_pack_weight_ex", ptr(w), ptr(wp), ptr(wt), k, w.shape[1], 0, c0, 9, 1, ptr(bias), ptr(b_eff))
        return wp, wt, b_eff, bias, bias._version
    v = _cached(key, w, make, meta=bias)
    if v[3] is not bias or v[4] != bias._version:               # the bias changed without the weight: repack
        _PACK_CACHE.pop((key, id(w)), None)
        v = _cached(key, w, make, meta=bias)
    return v[0], v[1], v[2]


def pack_gconv_weight(w: torch.Tensor, flip: int):
    def make():
        c = w.shape[0]
        wk = torch.empty((9 * 8 * c,), device=w.device, dtype=BF16)
        wd = torch.empty((9 * 8 * c,), device=w.device, dtype=BF16)
        lib().call("hn_gconv_pack", ptr(w), ptr(wk), ptr(wd), c, flip)
        return wk, wd
    return _cached(("g", flip), w, make)


def pack_gconv_diag(w: torch.Tensor):
    """grouped weights [C, 8, 3, 3] -> block-diagonal MFMA operands (wk forward, wd stride-1 dgrad), bf16 [C, 9*64]"""
    def make():
        c = w.shape[0]
        wk = torch.empty((c, 9 * 64), device=w.device, dtype=BF16)
        wd = torch.empty((c, 9 * 64), device=w.device, dtype=BF16)
        lib().call("hn_gconv_pack_diag", ptr(w), ptr(wk), ptr(wd), c)
        return wk, wd
    return _cached("gdiag", w, make)


def pack_dw_weight(w: torch.Tensor):
    def make():
        c = w.shape[0]
        wk = torch.empty((9 * c,), device=w.device, dtype=BF16)
        wf = torch.empty((9 * c,), device=w.device, dtype=BF16)
        lib().call("hn_dw_pack", ptr(w), ptr(wk), ptr(wf), c)
        return wk, wf
    return _cached("dw", w, make)


# --------------------------------------------------------------------------------------------------------------
# raw kernel wrappers (no autograd)
# --------------------------------------------------------------------------------------------------------------
def k_gemm_nt(x0, x1, mode, grid, wp, nout, kp, taps, bias=None, act=ACT_NONE, out=None, out_f32=False, up=0, stats=False,
              c0=None, c1=None, rpi=0, img_stride=0, ldc=None, xform=None, addend=None, add_pre=False, add_s2=False, estat=None):
    """grid = (N, H, W) of the OUTPUT pixel grid.  Returns (out, psum, psq).
    estat = (emode, ez, ecoef): the statistics rows carry the SE gate-gradient partials (emode 1; psq is None) or the BatchNorm-backward
    partial sums (emode 2) of (output, ez) instead of the output's BatchNorm statistics (hn_conv_gemm_nt_stat).
    xform = (scale, shift, gate | None, rows_per_image, act): operand transform of hn_conv_gemm_nt_ex; addend: bf16 tensor added in the
    epilogue (same rows / channels as the output; add_s2: the addend lives on the stride-2 sub-grid and is added at even (y, x))."""
    n, h, w = grid
    m = n * h * w
    dev = x0.device
    c0 = x0.shape[3] if c0 is None else c0
    c1 = (x1.shape[3] if x1 is not None else 0) if c1 is None else c1
    if out is None:
        out = torch.empty((n, h, w, nout), device=dev, dtype=F32 if out_f32 else BF16)
    if ldc is None:
        ldc = out.stride(2) if out.dim() == 4 else nout
    psum = psq = None
    if stats or estat is not None:
        pr = lib().query("hn_direct_stat_rows", n, h, w) if mode == 5 else lib().query("hn_nt_stat_rows", m, nout)
        psum = torch.empty((pr, nout), device=dev, dtype=F32)
        psq = torch.empty((pr, nout), device=dev, dtype=F32) if (estat is None or estat[0] != 1) else None
    if estat is not None:
        assert xform is None and not add_pre
        emode, ez, ecoef = estat[:3]
        if emode == 3:                                          # (3, ez, ecoef, ey): the previous XBlock's BN3-backward reduce pass
            ey = estat[3]
            assert mode == 0 and taps == 1 and addend is not None and not add_s2 and x1 is None and not out_f32
            lib().call("hn_conv_gemm_nt_stat3", ptr(x0), ld(x0), m, c0, ptr(wp), nout, kp, ptr(out), ldc, ptr(psum), ptr(psq), ptr(addend),
                       ld(addend), ptr(ez), ld(ez), ptr(ey), ld(ey), ptr(ecoef))
            return out, psum, psq
        lib().call("hn_conv_gemm_nt_stat", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
                   ptr(wp), nout, kp, taps, ptr(bias), act, ptr(out), 0, ldc, rpi, img_stride, ptr(psum), ptr(psq), ptr(addend),
                   ld(addend) if addend is not None else 0, 1 if add_s2 else 0, emode, ptr(ez), ld(ez), ptr(ecoef))
    elif xform is None and addend is None:
        lib().call("hn_conv_gemm_nt", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
                   ptr(wp), nout, kp, taps, ptr(bias), act, ptr(out), 1 if out_f32 else 0, ldc, rpi, img_stride, ptr(psum), ptr(psq))
    else:
        xs, xh, xg, xhw, xact = xform if xform is not None else (None, None, None, 0, 0)
        lib().call("hn_conv_gemm_nt_ex", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
                   ptr(wp), nout, kp, taps, ptr(bias), act, ptr(out), 1 if out_f32 else 0, ldc, rpi, img_stride, ptr(psum), ptr(psq),
                   ptr(xs), ptr(xh), ptr(xg), xhw, xact, ptr(addend),
                   (-ld(addend) if add_pre else ld(addend)) if addend is not None else 0, 1 if add_s2 else 0)
    return out, psum, psq


def wgrad_bias_ok(mode, kp):
    """weight-gradient launches that can return the conv's bias gradient (column sums of dz) from the same pass: all but the grouped mode"""
    return mode != 5


class WgradBatch:
    """collects the slab reduces of up to four weight gradients (k_gemm_tn(..., defer=batch)) and runs them in one launch (flush()); the
    gradients are complete only after flush()"""

    def __init__(self):
        self.jobs = (ctypes.c_long * 32)()
        self.n = 0
        self.keep = []                  # workspaces stay alive until the reduce was enqueued

    def slot(self, ws):
        assert self.n < 4
        self.keep.append(ws)
        self.n += 1
        return ctypes.addressof(self.jobs) + 64 * (self.n - 1)

    def flush(self):
        if self.n:
            lib().call("hn_wgrad_reduce_jobs", ctypes.addressof(self.jobs), self.n)
        self.n = 0
        self.keep = []


DEFER_WGRAD = policy("HN_DEFER_WGRAD", "1") != "0"   # 1x1 weight gradients of a backbone stage in one grouped launch at the stage boundary


class GradQueue:
    """Parameter gradients that are NOT on the backward pass's critical path, deferred to a segment boundary (DeferredGrads.backward) and
    finished by a handful of grouped launches instead of one or two small launches each:
      * add_gemm: 1x1-conv weight gradients -> one grouped GEMM launch (hn_wgrad_group; + one slab reduce if pixel splits were needed);
      * add_rows / add_fuse / add_se: partial-row folds of depthwise / stride-2 grouped conv weight gradients, BiFPN fusion-weight
        Jacobians, SE MLP outer products -> one launch (hn_grad_tail).
    Launched one by one behind every data gradient these were ~350 of the ~1600 launches of a step, each at or near the ~5 us floor of a
    dependent launch.  The backward nodes only QUEUE (parameter, operands) here and return None for the parameter; the queue keeps the
    operands alive; DeferredGrads hands the gradients to autograd under the same parameter objects."""
    MAX_GEMM = 32
    MAX_TAIL = 64

    def __init__(self):
        self.weights = ()           # the parameters DeferredGrads hands gradients back for, in its argument order
        self.jobs = []              # 1x1 weight gradients: (weight, x0, dz, mode, (n, h, w), cin, nout)
        self.gconv = []             # grouped 3x3 weight gradients: (weight, x, dz, (n, h, w), c)
        self.tail = []              # (weights tuple, kind, a, b, n0, n1, n2, out shapes)

    # -- queueing ------------------------------------------------------------------------------------------------------------------
    def add(self, weight, x0, dz, mode, grid, cin, nout):
        self.jobs.append((weight, x0, dz, mode, grid, cin, nout))

    add_gemm = add

    def add_gconv(self, weight, x, dz, grid, c):
        """weight.grad [c, 8, 3, 3] of a stride-1 grouped 3x3 conv (group width 8): input x, output gradient dz, both [n, h, w, c] bf16"""
        self.gconv.append((weight, x, dz, grid, c))

    def add_rows(self, weight, part, rows, cols, shape):
        """weight.grad (shape `shape`, rows * 0 + cols elements) = column sums of part [rows, cols]"""
        self.tail.append(((weight,), 0, part, None, rows, cols, 0, (shape,)))

    def add_fuse(self, praw, pw, blocks, eps):
        self.tail.append(((praw,), 1, pw, praw, blocks, praw.numel(), ctypes.c_uint.from_buffer(ctypes.c_float(eps)).value, (tuple(praw.shape),)))

    def add_outer(self, weight, bias, p, q):
        """weight.grad [PI, QJ, 1, 1] = p^T q, bias.grad [PI] = column sums of p;  p fp32 [N, PI], q fp32 [N, QJ] (SE MLP)"""
        self.tail.append(((weight, bias), 2, p, q, p.shape[1], q.shape[1], p.shape[0], (tuple(weight.shape), tuple(bias.shape))))

    # -- flushing ------------------------------------------------------------------------------------------------------------------
    def flush(self):
        """-> one fp32 gradient (or None) per entry of self.weights"""
        jobs, self.jobs = self.jobs, []
        tail, self.tail = self.tail, []
        gconv, self.gconv = self.gconv, []
        out = {}
        taken = set()               # parameters whose gradient went straight into their data-parallel bucket slot (grad_out)

        def put(wgt, g):
            out[id(wgt)] = g if id(wgt) not in out else out[id(wgt)] + g      # (a weight applied several times: per-level det towers)
        for c0 in range(0, len(jobs), self.MAX_GEMM):
            chunk = jobs[c0:c0 + self.MAX_GEMM]
            tab = (ctypes.c_long * (12 * len(chunk)))()
            dws = []
            for i, (wgt, x0, dz, mode, (n, h, w), cin, nout) in enumerate(chunk):
                dw = grad_out(wgt, (nout, cin, 1, 1), dz.device, taken)
                dws.append(dw)
                ldz = dz.stride(2) if dz.dim() == 4 else dz.stride(0)
                tab[12 * i:12 * i + 12] = [x0.data_ptr(), dz.data_ptr(), dw.data_ptr(), mode, n, h, w, cin, ld(x0), ldz, nout, n * h * w]
            wsb = lib().query("hn_wgrad_group_ws_bytes", ctypes.addressof(tab), len(chunk))
            if wsb < 0:
                raise RuntimeError("hn_wgrad_group: bad job table")
            ws = torch.empty((wsb // 4,), device=chunk[0][2].device, dtype=F32)
            lib().call("hn_wgrad_group", ctypes.addressof(tab), len(chunk), ptr(ws))
            for (wgt, *_), dw in zip(chunk, dws):
                put(wgt, dw)
        for c0 in range(0, len(gconv), self.MAX_GEMM):
            chunk = gconv[c0:c0 + self.MAX_GEMM]
            tab = (ctypes.c_long * (9 * len(chunk)))()
            dws = []
            for i, (wgt, x, dz, (n, h, w), c) in enumerate(chunk):
                dw = grad_out(wgt, (c, 8, 3, 3), dz.device, taken)
                dws.append(dw)
                tab[9 * i:9 * i + 9] = [x.data_ptr(), dz.data_ptr(), dw.data_ptr(), n, h, w, c, ld(x), ld(dz)]
            wsb = lib().query("hn_gconv_wgrad_group_ws_bytes", ctypes.addressof(tab), len(chunk))
            if wsb < 0:
                raise RuntimeError("hn_gconv_wgrad_group: bad job table")
            ws = torch.empty((wsb // 4,), device=chunk[0][2].device, dtype=F32)
            lib().call("hn_gconv_wgrad_group", ctypes.addressof(tab), len(chunk), ptr(ws))
            for (wgt, *_), dw in zip(chunk, dws):
                put(wgt, dw)
        for c0 in range(0, len(tail), self.MAX_TAIL):
            chunk = tail[c0:c0 + self.MAX_TAIL]
            tab = (ctypes.c_long * (8 * len(chunk)))()
            done = []
            for i, (wts, kind, a, b, n0, n1, n2, shapes) in enumerate(chunk):
                outs = [grad_out(wgt, shp, a.device, taken) for wgt, shp in zip(wts, shapes)]
                tab[8 * i:8 * i + 8] = [kind, a.data_ptr(), b.data_ptr() if b is not None else 0, outs[0].data_ptr(),
                                        outs[1].data_ptr() if len(outs) > 1 else 0, n0, n1, n2]
                done += list(zip(wts, outs))
            lib().call("hn_grad_tail", ctypes.addressof(tab), len(chunk))
            for wgt, g in done:         # only now: put() ADDS when a weight was queued more than once (per-level det towers), and the sum
                put(wgt, g)             # must read what the launch above has written
        return [out.get(id(w)) for w in self.weights]


def grad_out(wgt, shape, dev, taken):
    """fp32 output tensor for a parameter's gradient.  When a data-parallel reducer registered the parameter's slot in its flat fp32 bucket
    (ddp.GradReducer.arm: wgt._hn_grad_slot = (flat, offset)) and this is the parameter's first gradient of the step, a FRESH view of that
    slot: autograd's AccumulateGrad adopts the tensor as .grad, so the bucket already holds the gradient when the exchange starts and the
    gather copy skips it.  A new tensor otherwise."""
    slot = getattr(wgt, "_hn_grad_slot", None)
    if slot is not None and wgt.grad is None and id(wgt) not in taken:
        flat, off = slot
        n = 1
        for d in shape:
            n *= d
        if flat.device == dev and flat.dtype == F32:
            taken.add(id(wgt))
            return flat[off:off + n].view(shape)
    return torch.empty(shape, device=dev, dtype=F32)


WgradGroup = GradQueue
_CUR_QUEUE = None            # the GradQueue of the segment whose forward is being built (HydraNet sets it around neck + det / lane heads)


def set_queue(q):
    global _CUR_QUEUE
    _CUR_QUEUE = q


def cur_queue():
    return _CUR_QUEUE if DEFER_WGRAD else None


SIDE_FLUSH = policy("HN_SIDE_FLUSH", "0") == "1"     # run the deferred-gradient flushes on a side HIP stream (a hipGraph branch)
_SIDE = {}                                                   # device -> (stream, [tensors kept alive until the join], join-queued flag)


def _side_state(dev):
    st = _SIDE.get(dev)
    if st is None:
        st = _SIDE[dev] = [torch.cuda.Stream(device=dev), [], False]
    return st


def _side_join(dev):
    """end of the backward pass (engine callback): the main stream waits for the side stream's flushes; the operands may be recycled"""
    st = _SIDE[dev]
    torch.cuda.current_stream(dev).wait_stream(st[0])
    st[1].clear()
    st[2] = False


class DeferredGrads(torch.autograd.Function):
    """Identity on the tensor that ENTERS a segment (a backbone stage; the backbone's last output for the neck + heads).  Its backward runs
    when the gradient leaves the segment -- after every node of the segment has run its backward and queued its deferred parameter
    gradients in `group` (autograd runs ready nodes in reverse creation order) -- flushes the queue and returns the gradients for `weights`
    (the same parameter tensors the segment's nodes received; those nodes return None for them)."""

    @staticmethod
    def forward(ctx, x, group, *weights):
        ctx.group = group
        group.weights = weights
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if SIDE_FLUSH and g.is_cuda:
            # Deferred gradients are off the critical path by construction: their launches go to a side stream (inside a captured step: a
            # branch of the hipGraph) next to the latency-bound data-gradient chain, which leaves most of the chip idle; one join at the
            # end of the backward pass.
            dev = g.device
            st = _side_state(dev)
            cur = torch.cuda.current_stream(dev)
            st[0].wait_stream(cur)
            st[1].append((list(ctx.group.jobs), list(ctx.group.tail)))          # operands stay alive until the join
            with torch.cuda.stream(st[0]):
                grads = ctx.group.flush()
            if not st[2]:
                st[2] = True
                torch.autograd.Variable._execution_engine.queue_callback(lambda: _side_join(dev))
            return (g, None, *grads)
        return (g, None, *ctx.group.flush())


def k_gemm_tn(x0, x1, mode, grid, dz, nout, kp, taps, cin, up=0, kh=1, want_bias=False, defer=None):
    """weight gradient, returns fp32 [nout, cin, kh, kh] (want_bias: and the bias gradient [nout] out of the same launches).
    defer: a WgradBatch -- the slab reduce is left to its flush()."""
    n, h, w = grid
    m = n * h * w
    dev = x0.device
    splits, rps, wsb = ctypes.c_int(), ctypes.c_long(), ctypes.c_long()
    lib().query("hn_wgrad_plan", mode, n, h, w, m, nout, kp, taps, ctypes.addressof(splits), ctypes.addressof(rps), ctypes.addressof(wsb))
    ws = torch.empty((wsb.value // 4,), device=dev, dtype=F32)
    dw = torch.empty((nout, cin, kh, kh), device=dev, dtype=F32)
    c0 = x0.shape[3]
    c1 = x1.shape[3] if x1 is not None else 0
    ldz = dz.stride(2) if dz.dim() == 4 else dz.stride(0)
    if defer is not None:
        assert not want_bias
        lib().call("hn_conv_gemm_tn_deferred", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
                   ptr(dz), ldz, nout, kp, taps, ptr(ws), ptr(dw), defer.slot(ws))
        return dw
    if want_bias:
        db = torch.empty((nout,), device=dev, dtype=F32)
        lib().call("hn_conv_gemm_tn_bias", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
                   ptr(dz), ldz, nout, kp, taps, ptr(ws), ptr(dw), ptr(db))
        return dw, db
    lib().call("hn_conv_gemm_tn", ptr(x0), ptr(x1), mode, n, h, w, c0, c1, ld(x0), ld(x1) if x1 is not None else 0, up, m,
               ptr(dz), ldz, nout, kp, taps, ptr(ws), ptr(dw))
    return dw


def k_col_stats(x, align=0):
    m, c = rows(x), x.shape[3]
    r = lib().query("hn_colred_rows", m, align)
    pr = (m + r - 1) // r
    ps = torch.empty((pr, c), device=x.device, dtype=F32)
    pq = torch.empty((pr, c), device=x.device, dtype=F32)
    lib().call("hn_col_stats", ptr(x), ld(x), m, c, r, ptr(ps), ptr(pq))
    return ps, pq, r


def k_rows_reduce(part, groups, s, c, alpha=1.0):
    """out[g][c] = alpha * sum_j part[g*s + j][c].  A single tall group (the partial rows of a wgrad / column reduction) is folded in two
    2-D launches: 16 row lanes walking thousands of rows serially took 35 us, two short folds take < 10 us."""
    if groups == 1 and s > 256:
        tmp = torch.empty((32, c), device=part.device, dtype=F32)
        lib().call("hn_rows_reduce2", ptr(part), None, ptr(tmp), None, s, 32, c)
        part, s = tmp, 32
    out = torch.empty((groups, c), device=part.device, dtype=F32)
    lib().call("hn_rows_reduce", ptr(part), ptr(out), groups, s, c, float(alpha))
    return out


def fold_rows(p1, p2, limit=128, groups=32):
    """partial-row arrays with more than `limit` rows are folded to `groups` rows by one 2-D launch (the finalize kernels walk the rows
    with 8 lanes per channel)"""
    rows_, c = p1.shape
    if rows_ <= limit:
        return p1, p2
    o = torch.empty((2, groups, c), device=p1.device, dtype=F32)
    lib().call("hn_rows_reduce2", ptr(p1), ptr(p2), ptr(o[0]), ptr(o[1]), rows_, groups, c)
    return o[0], o[1]


def k_bn_finalize(psum, psq, count, gamma, beta, eps, momentum, rm, rv):
    c = gamma.shape[0]
    psum, psq = fold_rows(psum, psq)
    coef = torch.empty((4, c), device=gamma.device, dtype=F32)     # scale, shift, mean, rstd
    lib().call("hn_bn_finalize", ptr(psum), ptr(psq), psum.shape[0], c, count, ptr(gamma), ptr(beta), float(eps), float(momentum),
               ptr(rm), ptr(rv), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]))
    return coef


def k_bn_eval_coeff(gamma, beta, rm, rv, eps):
    c = gamma.shape[0]
    coef = torch.empty((4, c), device=gamma.device, dtype=F32)
    lib().call("hn_bn_eval_coeff", ptr(gamma), ptr(beta), ptr(rm), ptr(rv), float(eps), c, ptr(coef[0]), ptr(coef[1]))
    return coef


def k_bn_act(z, coef, act, res=None, out=None):
    n, h, w, c = z.shape
    if out is None:
        out = new_act(n, h, w, c, z.device)
    lib().call("hn_bn_act", ptr(z), ld(z), ptr(coef[0]) if coef is not None else None, ptr(coef[1]) if coef is not None else None,
               ptr(res), ld(res) if res is not None else 0, None, None, act, ptr(out), ld(out), rows(z), c)
    return out


def k_eltwise(op, a, b=None, act=ACT_NONE, alpha=1.0, out=None):
    n, h, w, c = a.shape
    if out is None:
        out = new_act(n, h, w, c, a.device)
    lib().call("hn_eltwise", op, ptr(a), ld(a), ptr(b), ld(b) if b is not None else 0, ptr(out), ld(out), rows(a), c, act, float(alpha))
    return out


FUSED_BN = policy("HN_FUSED_BN", "1") != "0"   # BatchNorm finalize in the prologue of the consuming elementwise kernel (hn_fused.hip); False: round-1 kernels
MAX_PROLOGUE_ROWS = int(policy("HN_MAX_PROLOGUE_ROWS", "128"))    # partial rows a consumer prologue reduces itself (+1.5 us at 128 rows); more are folded to 32 rows first (one
                           # ~5 us launch, only for the large early-stage tensors whose passes take 15-40 us anyway)


def k_col_stats_fused(x, align=0):
    """per-row-block channel sums / sums of squares of a bf16 tensor: (psum, psq) [P <= 512][C]"""
    m, c = rows(x), x.shape[3]
    rb = lib().query("hn_fused_row_block", m, c, align, 0, 1)
    pr = (m + rb - 1) // rb
    ps = torch.empty((pr, c), device=x.device, dtype=F32)
    pq = torch.empty((pr, c), device=x.device, dtype=F32)
    lib().call("hn_col_stats_fused", ptr(x), ld(x), m, c, rb, ptr(ps), ptr(pq))
    return ps, pq


def k_bn_apply_fused(z, psum, psq, count, gamma, beta, eps, momentum, rm, rv, act, res=None, out=None, want_out=True, pool_align=0,
                     training=True, coef=None, gate=None, hw=0):
    """out = act(BN(z) [+ res]) [* gate] with the BatchNorm finalize in the kernel prologue.  coef given: use it as is (no statistics).
    Returns (out, coef [4, C], pool_partials | None, RB)."""
    n, h, w, c = z.shape
    m = rows(z)
    dev = z.device
    if coef is not None:
        P = 0
    elif training:
        if psum.shape[0] > MAX_PROLOGUE_ROWS:
            psum, psq = fold_rows(psum, psq, limit=MAX_PROLOGUE_ROWS)
        P = psum.shape[0]
    else:
        P = -1
    rb = lib().query("hn_fused_row_block", m, c, pool_align or hw, max(P, 0), 0)
    if coef is None:
        coef = torch.empty((4, c), device=dev, dtype=F32)
    if want_out and out is None:
        out = new_act(n, h, w, c, dev)
    pool = torch.empty(((m + rb - 1) // rb, c), device=dev, dtype=F32) if pool_align else None
    lib().call("hn_bn_apply_fused", ptr(z), ld(z), m, c, ptr(psum) if P > 0 else None, ptr(psq) if P > 0 else None, P, count,
               ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(rm), ptr(rv), ptr(coef), ptr(res), ld(res) if res is not None else 0,
               act, ptr(out) if want_out else None, ld(out) if want_out else 0, ptr(pool), ptr(gate), hw, rb)
    return out, coef, pool, rb


def bn_backward_fused(dout, z, y, coef, act, count, want_g=False, gate=None, dpool=None, hw=0, zero_c=None, parts=None):
    """BatchNorm(+activation) backward in two launches (reduce, apply with the finalize in its prologue): (dz, dgamma, dbeta, g|None).
    The two passes use their own row blocks: the reduce pass's block count is the number of partial rows the apply prologue folds.
    parts = (pg, pgx): the partial sums already exist (the producer of dout made them in its epilogue, k_gemm_nt(estat=(2, ...))): one launch."""
    n, h, w, c = z.shape
    m = rows(z)
    dev = z.device
    if parts is not None:
        pg, pgx = parts
        pr = pg.shape[0]
    else:
        rb_r = lib().query("hn_fused_row_block", m, c, hw, 0, 1)
        pr = (m + rb_r - 1) // rb_r
        pg = torch.empty((pr, c), device=dev, dtype=F32)
        pgx = torch.empty((pr, c), device=dev, dtype=F32)
        lib().call("hn_bn_bwd_reduce_fused", ptr(dout), ld(dout), ptr(z), ld(z), ptr(y), ld(y) if y is not None else 0, ptr(coef), act,
                   ptr(gate), ptr(dpool), hw, m, c, rb_r, ptr(pg), ptr(pgx))
    if pr > MAX_PROLOGUE_ROWS:
        pg, pgx = fold_rows(pg, pgx, limit=MAX_PROLOGUE_ROWS)
        pr = pg.shape[0]
    rb_a = lib().query("hn_fused_row_block", m, c, hw, pr, 0)
    dgamma = torch.empty((c,), device=dev, dtype=F32)
    dbeta = torch.empty((c,), device=dev, dtype=F32)
    dz = new_act(n, h, w, c, dev)
    g = new_act(n, h, w, c, dev) if want_g else None
    lib().call("hn_bn_bwd_apply_fused", ptr(dout), ld(dout), ptr(z), ld(z), ptr(y), ld(y) if y is not None else 0, ptr(coef), act,
               ptr(gate), ptr(dpool), hw, ptr(pg), ptr(pgx), pr, count, ptr(dgamma), ptr(dbeta), ptr(dz), ld(dz), ptr(g),
               ld(g) if g is not None else 0, m, c, rb_a, ptr(zero_c))
    return dz, dgamma, dbeta, g


def bn_backward(dout, z, y, coef, act, count, want_g=False):
    """shared BatchNorm(+activation) backward: returns (dz, dgamma, dbeta, g|None)."""
    n, h, w, c = z.shape
    m = rows(z)
    r = lib().query("hn_colred_rows", m, 0)
    pr = (m + r - 1) // r
    dev = z.device
    pg = torch.empty((pr, c), device=dev, dtype=F32)
    pgx = torch.empty((pr, c), device=dev, dtype=F32)
    lib().call("hn_bn_bwd_reduce", ptr(dout), ld(dout), ptr(z), ld(z), ptr(y), ld(y) if y is not None else 0, ptr(coef[0]), ptr(coef[1]),
               ptr(coef[2]), ptr(coef[3]), act, m, c, r, ptr(pg), ptr(pgx))
    red = torch.empty((2, c), device=dev, dtype=F32)               # mean(g), mean(g*xhat)
    dgamma = torch.empty((c,), device=dev, dtype=F32)
    dbeta = torch.empty((c,), device=dev, dtype=F32)
    pg, pgx = fold_rows(pg, pgx)
    pr = pg.shape[0]
    lib().call("hn_bn_bwd_finalize", ptr(pg), ptr(pgx), pr, c, count, ptr(dgamma), ptr(dbeta), ptr(red[0]), ptr(red[1]))
    dz = new_act(n, h, w, c, dev)
    g = new_act(n, h, w, c, dev) if want_g else None
    lib().call("hn_bn_bwd_apply", ptr(dout), ld(dout), ptr(z), ld(z), ptr(y), ld(y) if y is not None else 0, ptr(coef[0]), ptr(coef[1]),
               ptr(coef[2]), ptr(coef[3]), ptr(red[0]), ptr(red[1]), act, ptr(dz), ld(dz), ptr(g), ld(g) if g is not None else 0, m, c)
    return dz, dgamma, dbeta, g


# geometry of level-packed tensors (det-head towers, ops.heads; the depthwise kernels of ops.neck take the packed form too)
LEVEL_ALIGN = 128           # rows: GEMM pixel tiles / BatchNorm row blocks never straddle two pyramid levels


def levels_packable(feats):
    return len(feats) <= 5


def _pad_rows(r):
    return (r + LEVEL_ALIGN - 1) // LEVEL_ALIGN * LEVEL_ALIGN


def _geom_arrays(geom):
    """(nlev, H[], W[], padded rows per level R[], real rows per level CNT[]) as ctypes arrays"""
    n, hs, ws = geom
    nl = len(hs)
    H = (ctypes.c_int * nl)(*hs)
    W = (ctypes.c_int * nl)(*ws)
    R = (ctypes.c_long * nl)(*[_pad_rows(n * h * w) for h, w in zip(hs, ws)])
    CNT = (ctypes.c_long * nl)(*[n * h * w for h, w in zip(hs, ws)])
    return nl, H, W, R, CNT


__all__ = [n for n in dir() if not n.startswith("__")]      # everything, incl. single-underscore helpers: the package is one namespace
