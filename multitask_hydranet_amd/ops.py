"""Host-side operator layer: thin wrappers that allocate outputs with torch and enqueue the HIP kernels of
libhydranet_hip.so, plus the torch.autograd.Function objects that give them a backward.

This is the same plug-in point the reference uses for its one hand-written fwd/bwd op (SwishImplementation,
model/net/common.py:11-22): torch.autograd.Function.forward/backward.  Tensors here are NHWC bf16 ("channels last"):
shape [N, H, W, C], unit channel stride, possibly a channel-slice view of a wider buffer (row stride = stride(2)).
There is no eager / CPU fallback in this module: every op goes through lib().call and raises if the library is missing.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from ._lib import lib

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
                      wp.numel() + (wt.numel() if wt is not None else 0) + (coute if phase else 0))
                self.values.append((wp, wt, b_eff, bias) if phase else (wp, wt))
            else:
                raise KeyError(key)
        self.dense_blocks, self.small_blocks = dblk, sblk[0]
        mk = lambda rows_, dt: torch.tensor(rows_, dtype=dt).to(dev) if rows_ else None
        self.dense_table, self.dense_owner = mk(dense_rows, torch.int64), mk(dense_owner, torch.int32)
        self.small_table, self.small_owner = mk(small_rows, torch.int64), mk(small_owner, torch.int32)
        self.n_dense = len(dense_rows)
        self.device = dev
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

    def run(self):
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
        emode, ez, ecoef = estat
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


DEFER_WGRAD = os.environ.get("HN_DEFER_WGRAD", "1") != "0"   # 1x1 weight gradients of a backbone stage in one grouped launch at the stage boundary


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


SIDE_FLUSH = os.environ.get("HN_SIDE_FLUSH", "0") == "1"     # run the deferred-gradient flushes on a side HIP stream (a hipGraph branch)
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


FUSED_BN = os.environ.get("HN_FUSED_BN", "1") != "0"   # BatchNorm finalize in the prologue of the consuming elementwise kernel (hn_fused.hip); False: round-1 kernels
MAX_PROLOGUE_ROWS = int(os.environ.get("HN_MAX_PROLOGUE_ROWS", "128"))    # partial rows a consumer prologue reduces itself (+1.5 us at 128 rows); more are folded to 32 rows first (one
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


# --------------------------------------------------------------------------------------------------------------
# conv (1x1 | 1x1 stride 2 | grouped 3x3 | stem) + BatchNorm + activation (+ residual)
# --------------------------------------------------------------------------------------------------------------
class ConvBnAct(torch.autograd.Function):
    """out = act(BN(conv(x) [+ conv_bias]) [+ res]).  kind: "1x1", "g3x3", "stem"."""

    @staticmethod
    def forward(ctx, x, weight, conv_bias, gamma, beta, rm, rv, nbt, res, kind, stride, act, eps, momentum, training, slot=None):
        dev = x.device
        cout = weight.shape[0]
        ctx.slot = slot if (kind == "1x1" and stride == 1) else None     # GradSlot of x (see Share): dgrad accumulates in its GEMM epilogue
        if kind == "stem":
            n, _, hi, wi = x.shape
            ho, wo = hi // 2, wi // 2
            z = new_act(n, ho, wo, 32, dev)
            packs = new_act(n, ho, wo, 32, dev) if training else None       # bf16 im2col rows for the MFMA weight gradient
            lib().call("hn_stem_fwd", ptr(x), ptr(weight), ptr(z), ptr(packs), n, hi, wi)
            psum = psq = None
        elif kind == "g3x3":
            n, hi, wi, c = x.shape
            ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
            if stride == 1 and GCONV_MFMA:          # block-diagonal 64-channel tiles on MFMA (hn_conv_gemm_nt mode 5)
                packs = pack_gconv_diag(weight)
                z, _, _ = k_gemm_nt(x, None, 5, (n, ho, wo), packs[0], c, 64, 9)
            else:
                packs = pack_gconv_weight(weight, 1 if stride == 1 else 0)
                z = new_act(n, ho, wo, c, dev)
                # (stride 2 contracts with packed bf16 dots over the input channels: the pack with i contiguous = packs[1], hydranet_hip.h)
                lib().call("hn_gconv_fwd", ptr(x), ld(x), ptr(packs[0] if stride == 1 else packs[1]), ptr(z), ld(z), n, hi, wi, c, stride)
            psum = psq = None
        else:
            n, hi, wi, cin = x.shape
            ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
            packs = pack_conv_weight(weight)
            z, psum, psq = k_gemm_nt(x, None, 0 if stride == 1 else 1, (n, ho, wo), packs[0], cout, kp32(cin), 1, bias=conv_bias,
                                     stats=training)
        count = n * ho * wo
        if FUSED_BN:
            if training and psum is None:
                psum, psq = k_col_stats_fused(z)
            out, coef, _, _ = k_bn_apply_fused(z, psum, psq, count, gamma, beta, eps, momentum, rm, rv, act, res=res, training=training)
            if training and nbt is not None:
                nbt.add_(1)
        else:
            if training:
                if psum is None:
                    psum, psq, _ = k_col_stats(z)
                coef = k_bn_finalize(psum, psq, count, gamma, beta, eps, momentum, rm, rv)
                if nbt is not None:
                    nbt.add_(1)
            else:
                coef = k_bn_eval_coeff(gamma, beta, rm, rv, eps)
            out = k_bn_act(z, coef, act, res=res)
        ctx.kind, ctx.stride, ctx.act, ctx.count = kind, stride, act, count
        ctx.queue = cur_queue() if (kind == "1x1" and training and weight.requires_grad) else None
        ctx.wref = weight
        ctx.has_bias = conv_bias is not None
        ctx.has_res = res is not None
        ctx.training = training
        ctx.packs = packs
        # ReLU mask: recomputed from z in backward (scale*z+shift > 0); the saved output is only needed when a residual went into the ReLU
        y_save = out if (act == ACT_RELU and (res is not None or not FUSED_BN)) else None
        ctx.save_for_backward(x, weight, z, coef, y_save, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, z, coef, y, gamma = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        dout = dense(dout)
        kind, stride = ctx.kind, ctx.stride
        # a conv bias that feeds BatchNorm has zero gradient: the zeros come out of the BN backward launch (no fill kernel)
        dbias = torch.empty((z.shape[3],), device=z.device, dtype=F32) if ctx.has_bias else None
        if FUSED_BN:
            dz, dgamma, dbeta, g = bn_backward_fused(dout, z, y, coef, ctx.act, ctx.count, want_g=ctx.has_res and ctx.act != ACT_NONE,
                                                     zero_c=dbias)
        else:
            dz, dgamma, dbeta, g = bn_backward(dout, z, y, coef, ctx.act, ctx.count, want_g=ctx.has_res and ctx.act != ACT_NONE)
            if dbias is not None:
                dbias = zeros((z.shape[3],), z.device)
        dres = None
        if ctx.has_res:
            dres = g if g is not None else dout
        dev = z.device
        n, ho, wo, cout = z.shape
        dx = None
        if kind == "stem":
            dw = k_gemm_tn(ctx.packs, None, 0, (n, ho, wo), dz, 32, 32, 1, 32)[:, :27].reshape(32, 3, 3, 3)
        elif kind == "g3x3":
            _, hi, wi, c = x.shape
            wk, wd = ctx.packs
            mfma = stride == 1 and GCONV_MFMA
            if ctx.needs_input_grad[0]:
                if mfma:
                    dx, _, _ = k_gemm_nt(dz, None, 5, (n, hi, wi), wd, c, 64, 9)
                else:
                    dx = new_act(n, hi, wi, c, dev)
                    if stride == 1:
                        lib().call("hn_gconv_fwd", ptr(dz), ld(dz), ptr(wd), ptr(dx), ld(dx), n, hi, wi, c, 1)
                    else:
                        lib().call("hn_gconv_dgrad_s2", ptr(dz), ld(dz), ptr(wk), ptr(dx), ld(dx), n, hi, wi, c)
            if mfma:
                dw = k_gemm_tn(x, None, 5, (n, ho, wo), dz, c, 64, 9, 8, kh=3)
            else:
                chunks = lib().query("hn_wgrad_chunks", n * ho * wo, (c // 8) * 9)
                part = torch.empty((chunks, c * 72), device=dev, dtype=F32)
                lib().call("hn_gconv_wgrad", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), n, hi, wi, c, stride)
                dw = k_rows_reduce(part, 1, chunks, c * 72).view(c, 8, 3, 3)
        else:
            _, hi, wi, cin = x.shape
            wp, wt = ctx.packs
            if ctx.needs_input_grad[0] and ctx.slot is not None:
                # x has other consumers: the first one to run stores its data gradient, the others add theirs in the GEMM epilogue (in place)
                sl = ctx.slot
                if sl.buf is None:
                    sl.buf, _, _ = k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1)
                else:
                    k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1, addend=sl.buf, out=sl.buf)
            elif ctx.needs_input_grad[0]:
                dxs, _, _ = k_gemm_nt(dz, None, 0, (n, ho, wo), wt, cin, kp32(cout), 1)
                if stride == 1:
                    dx = dxs
                else:
                    dx = zeros((n, hi, wi, cin), dev, BF16)
                    lib().call("hn_add_strided2", ptr(dx), ld(dx), ptr(dxs), ld(dxs), n, ho, wo, cin)
            if ctx.queue is not None:
                dw = ctx.queue.add_gemm(ctx.wref, x, dz, 0 if stride == 1 else 1, (n, ho, wo), cin, cout)
            else:
                dw = k_gemm_tn(x, None, 0 if stride == 1 else 1, (n, ho, wo), dz, cout, kp32(cin), 1, cin)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, dres, None, None, None, None, None, None, None


def conv_bn_act(x, weight, conv_bias, bn, res=None, kind="1x1", stride=1, act=ACT_NONE, eps=1e-5, momentum=0.1, training=True, slot=None):
    gamma, beta, rm, rv, nbt = bn
    return ConvBnAct.apply(x, weight, conv_bias, gamma, beta, rm, rv, nbt, res, kind, stride, act, eps, momentum, training, slot)


# --------------------------------------------------------------------------------------------------------------
# Stride-1 identity XBlock as ONE autograd node (net/anynet.py:65-76; 25 of the 30 blocks of the big backbone):
#   z1 = conv1x1(x); a = relu(bn1(z1)); z2 = gconv3x3(a); b = relu(bn2(z2)); gate = SE(avgpool(b)); z3 = conv1x1(b * gate);
#   out = relu(bn3(z3) + x)
# Forward is 9 launches: the two 1x1 GEMMs and the grouped conv emit their BatchNorm partial statistics; BN1 apply and the final
# BN3 + residual + ReLU are materialising passes; BN2's output is never stored on its own -- one pass over z2 finalizes its statistics
# and produces the SE squeeze, a second one writes relu(bn2(z2)) * gate, the operand of conv_block_3.  Backward is 21 launches
# (BN backward = reduce + apply with the finalize in the prologue; the SE gate gradient and the gated wgrad operand come out of one pass
# over (dbg, z2); the SE data-path backward is folded into the BN2 reduce/apply pair; the residual gradient is added in conv_block_1's
# dgrad epilogue).  The unfused composition of ConvBnAct / SEGate nodes is ~16 + ~27 launches per block.
# --------------------------------------------------------------------------------------------------------------
FUSED_XBLOCK = os.environ.get("HN_FUSED_XBLOCK", "1") != "0"
EPILOGUE_STATS = os.environ.get("HN_EPILOGUE_STATS", "1") != "0"   # backward reduce passes folded into their producers' epilogues
XBLOCK_XF_GEMM = os.environ.get("HN_XBLOCK_XF", "0") == "1"


class XBlockFn(torch.autograd.Function):
    """stride 1 without shortcut: the identity blocks; stride 2 (or a channel change) with the projection shortcut conv + BN
    (ws, gs, bs, rms, rvs): the first block of every stage.  There the grouped conv runs on the stride-2 stencil kernels, and the data
    gradient of the shortcut (a stride-2 row gather) joins conv_block_1's data gradient in that GEMM's epilogue (add_s2): x has ONE
    consumer node, no zero-filled full-resolution tensor, no separate additions."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, sw1, sb1, sw2, sb2, w3, g3, b3, rm3, rv3, eps, momentum, training,
                stride=1, ws=None, gs=None, bs=None, rms=None, rvs=None, group=None):
        """group (WgradGroup of the stage, or None): the 1x1 weight gradients are queued there instead of being launched here"""
        n, h, w, cin = x.shape
        c = w1.shape[0]
        ho, wo = h // stride, w // stride
        m_in, m, hw = n * h * w, n * ho * wo, ho * wo
        cs = sw1.shape[0]
        dev = x.device
        grid = (n, ho, wo)
        wp1, wt1 = pack_conv_weight(w1)
        z1, ps, pq = k_gemm_nt(x, None, 0, (n, h, w), wp1, c, kp32(cin), 1, stats=training)
        a, coef1, _, _ = k_bn_apply_fused(z1, ps, pq, m_in, g1, b1, eps, momentum, rm1, rv1, ACT_RELU, training=training)
        if stride == 1:
            wk2, wd2 = pack_gconv_diag(w2)
            z2, ps, pq = k_gemm_nt(a, None, 5, grid, wk2, c, 64, 9, stats=training)
        else:
            # stride 2: packed-bf16-dot kernels, contraction index contiguous (hydranet_hip.h): the forward takes the (o, i)-swapped pack,
            # the data gradient the plain one
            wd2, wk2 = pack_gconv_weight(w2, 0)
            z2 = new_act(n, ho, wo, c, dev)
            lib().call("hn_gconv_fwd", ptr(a), ld(a), ptr(wk2), ptr(z2), ld(z2), n, h, w, c, stride)
            ps, pq = k_col_stats_fused(z2) if training else (None, None)
        _, coef2, pool, rb = k_bn_apply_fused(z2, ps, pq, m, g2, b2, eps, momentum, rm2, rv2, ACT_RELU, want_out=False, pool_align=hw,
                                              training=training)
        pooled = torch.empty((n, c), device=dev, dtype=F32)
        hid = torch.empty((n, cs), device=dev, dtype=F32)
        gate = torch.empty((n, c), device=dev, dtype=F32)
        lib().call("hn_se_mlp_fwd_parts", ptr(pool), hw // rb, 1.0 / hw, ptr(sw1), ptr(sb1), ptr(sw2), ptr(sb2), ptr(pooled), ptr(hid),
                   ptr(gate), n, c, cs)
        wp3, wt3 = pack_conv_weight(w3)
        if XBLOCK_XF_GEMM:      # BN2 + ReLU + gate in conv_block_3's operand loader (register-staged: measured 7-10 us slower per launch
            bg = None           # than the LDS-DMA loader, more than the extra pass below costs)
            z3, ps, pq = k_gemm_nt(z2, None, 0, grid, wp3, c, kp32(c), 1, stats=training, xform=(coef2[0], coef2[1], gate, hw, ACT_RELU))
        else:                   # second pass over z2: bg = relu(bn2(z2)) * gate, kept for conv_block_3's weight gradient
            bg, _, _, _ = k_bn_apply_fused(z2, None, None, m, g2, b2, eps, momentum, None, None, ACT_RELU, coef=coef2, gate=gate, hw=hw)
            z3, ps, pq = k_gemm_nt(bg, None, 0, grid, wp3, c, kp32(c), 1, stats=training)
        zs = coefs = wts = None
        res = x
        if ws is not None:      # projection shortcut: 1x1 conv (stride-2 row gather) + BatchNorm, no activation
            wps, wts = pack_conv_weight(ws)
            zs, pss, pqs = k_gemm_nt(x, None, 0 if stride == 1 else 1, grid, wps, c, kp32(cin), 1, stats=training)
            res, coefs, _, _ = k_bn_apply_fused(zs, pss, pqs, m, gs, bs, eps, momentum, rms, rvs, ACT_NONE, training=training)
        out, coef3, _, _ = k_bn_apply_fused(z3, ps, pq, m, g3, b3, eps, momentum, rm3, rv3, ACT_RELU, res=res, training=training)
        ctx.training, ctx.stride = training, stride
        ctx.packs = (wt1, wd2, wt3, wts)
        ctx.group = group
        ctx.wrefs = (w1, w3, ws, w2, sw1, sb1, sw2, sb2)   # identities under which the stage's DeferredGrads node returns the gradients
        ctx.save_for_backward(x, z1, a, z2, z3, out, coef1, coef2, coef3, pooled, hid, gate, sw1, sw2, bg, zs, coefs)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z1, a, z2, z3, out, coef1, coef2, coef3, pooled, hid, gate, sw1, sw2, bg, zs, coefs = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        wt1, wd2, wt3, wts = ctx.packs
        stride = ctx.stride
        dout = dense(dout)
        n, h, w, cin = x.shape
        _, ho, wo, c = z2.shape
        m_in, m, hw = n * h * w, n * ho * wo, ho * wo
        cs = sw1.shape[0]
        dev = x.device
        grid = (n, ho, wo)
        # out = relu(bn3(z3) + shortcut): g = dout * [out > 0] is also the gradient of the shortcut branch
        dz3, dg3, db3, g = bn_backward_fused(dout, z3, out, coef3, ACT_RELU, m, want_g=True)
        make_bg = bg is None
        # SE gate-gradient partials sum_rows dbg * relu(bn2(z2)) from the GEMM's own epilogue (one partial row per pixel tile) where a
        # tile lies inside one image and an image has few tiles (the deep stages); otherwise by a pass over (dbg, z2) below
        bp = m // lib().query("hn_nt_stat_rows", m, c)
        ep_dot = EPILOGUE_STATS and not make_bg and hw % bp == 0 and hw // bp <= 16
        dbg, pdot, _ = k_gemm_nt(dz3, None, 0, grid, wt3, c, kp32(c), 1, estat=(1, z2, coef2) if ep_dot else None)
        group = ctx.group                                     # WgradGroup: the 1x1 weight gradients wait for the stage boundary
        w1_, w3_, ws_, w2_, sw1_, sb1_, sw2_, sb2_ = ctx.wrefs
        batch = WgradBatch()                                  # the slab reduces of dw3 / dw2 / dw1 / dws: one launch at the end
        if not make_bg:                                       # dz3's second reader right behind the first: still in the XCDs' L2s
            dw3 = group.add(w3_, bg, dz3, 0, grid, c, c) if group is not None else k_gemm_tn(bg, None, 0, grid, dz3, c, kp32(c), 1, c, defer=batch)
        # one pass over (dbg, z2): gate-gradient partials and the gated operand bg = relu(bn2(z2)) * gate of conv_block_3's wgrad
        rb = bp
        if not ep_dot:
            rb = lib().query("hn_fused_row_block", m, c, hw, 0, 1)
            if make_bg:
                bg = new_act(n, ho, wo, c, dev)
            pdot = torch.empty(((m + rb - 1) // rb, c), device=dev, dtype=F32)
            lib().call("hn_se_bwd_reduce_fused", ptr(dbg), ld(dbg), ptr(z2), ld(z2), ptr(coef2), ptr(gate), hw, ptr(bg) if make_bg else None,
                       ld(bg), ptr(pdot), m, c, rb)
        if make_bg:
            dw3 = group.add(w3_, bg, dz3, 0, grid, c, c) if group is not None else k_gemm_tn(bg, None, 0, grid, dz3, c, kp32(c), 1, c, defer=batch)
        dpre2 = torch.empty((n, c), device=dev, dtype=F32)
        dpool = torch.empty((n, c), device=dev, dtype=F32)
        dpre1 = torch.empty((n, cs), device=dev, dtype=F32)
        if group is not None:                                  # the two outer products wait for the stage boundary (hn_grad_tail)
            dsw1 = dsb1 = dsw2 = dsb2 = None
            group.add_outer(sw2_, sb2_, dpre2, hid)
            group.add_outer(sw1_, sb1_, dpre1, pooled)
        else:
            dsw1, dsb1 = torch.empty_like(sw1), torch.empty((cs,), device=dev, dtype=F32)
            dsw2, dsb2 = torch.empty_like(sw2), torch.empty((c,), device=dev, dtype=F32)
        lib().call("hn_se_mlp_bwd_parts", ptr(pdot), hw // rb, ptr(gate), ptr(hid), ptr(pooled), ptr(sw1), ptr(sw2), ptr(dpre2), ptr(dpre1),
                   ptr(dpool), ptr(dsw1), ptr(dsb1), ptr(dsw2), ptr(dsb2), n, c, cs)
        # BN2 backward with the SE data path folded in: g2 = (dbg * gate + dpool / HW) * [bn2(z2) > 0]
        dz2, dg2, db2, _ = bn_backward_fused(dbg, z2, None, coef2, ACT_RELU, m, gate=gate, dpool=dpool, hw=hw)
        parts1 = None
        if stride == 1:
            # BatchNorm-1 backward partial sums (sum g, sum g * xhat over (da, z1)) from the data-gradient conv's epilogue: one row per
            # 16 x 16 patch, folded by the apply pass's prologue
            ep_bn = EPILOGUE_STATS and lib().query("hn_direct_stat_rows", n, ho, wo) <= MAX_PROLOGUE_ROWS
            da, pg1, pgx1 = k_gemm_nt(dz2, None, 5, grid, wd2, c, 64, 9, estat=(2, z1, coef1) if ep_bn else None)
            if ep_bn:
                parts1 = (pg1, pgx1)
            if group is not None:
                dw2 = group.add_gconv(w2_, a, dz2, grid, c)
            else:
                dw2 = k_gemm_tn(a, None, 5, grid, dz2, c, 64, 9, 8, kh=3, defer=batch)
        else:
            da = new_act(n, h, w, c, dev)
            lib().call("hn_gconv_dgrad_s2", ptr(dz2), ld(dz2), ptr(wd2), ptr(da), ld(da), n, h, w, c)
            chunks = lib().query("hn_wgrad_chunks", m, (c // 8) * 9)
            part = torch.empty((chunks, c * 72), device=dev, dtype=F32)
            lib().call("hn_gconv_wgrad", ptr(a), ld(a), ptr(dz2), ld(dz2), ptr(part), n, h, w, c, stride)
            if group is not None:
                dw2 = group.add_rows(w2_, part, chunks, c * 72, (c, 8, 3, 3))
            else:
                dw2 = k_rows_reduce(part, 1, chunks, c * 72).view(c, 8, 3, 3)
        dz1, dg1, db1, _ = bn_backward_fused(da, z1, None, coef1, ACT_RELU, m_in, parts=parts1)
        dws = dgs = dbs = None
        addend, add_s2 = g, False                               # identity block: + gradient of the identity branch
        if zs is not None:
            dzs, dgs, dbs, _ = bn_backward_fused(g, zs, None, coefs, ACT_NONE, m)
            addend, _, _ = k_gemm_nt(dzs, None, 0, grid, wts, cin, kp32(c), 1)           # shortcut data gradient on the output grid
            add_s2 = stride == 2
            if group is not None:
                group.add(ws_, x, dzs, 0 if stride == 1 else 1, grid, cin, c)
            else:
                dws = k_gemm_tn(x, None, 0 if stride == 1 else 1, grid, dzs, c, kp32(cin), 1, cin, defer=batch)
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _, _ = k_gemm_nt(dz1, None, 0, (n, h, w), wt1, cin, kp32(c), 1, addend=addend, add_s2=add_s2)
        if group is not None:
            dw1 = group.add(w1_, x, dz1, 0, (n, h, w), cin, c)
        else:
            dw1 = k_gemm_tn(x, None, 0, (n, h, w), dz1, c, kp32(cin), 1, cin, defer=batch)
        batch.flush()
        return (dx, dw1, dg1, db1, None, None, dw2, dg2, db2, None, None, dsw1, dsb1, dsw2, dsb2, dw3, dg3, db3, None, None,
                None, None, None, None, dws, dgs, dbs, None, None, None)


# --------------------------------------------------------------------------------------------------------------
# Inference with folded BatchNorm (BASELINE config 5; reference: demo.py:191-202 runs the eval-mode module).  In eval mode
# BN(conv(x)) = conv(x, W * scale) + shift with scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (+ conv bias
# * scale): HydraNet.prepare_inference() folds scale into the packed bf16 weights once and keeps shift as an fp32 bias, so conv + BN +
# activation (+ the XBlock's identity branch) is ONE GEMM launch with a bias / addend / activation epilogue.
# --------------------------------------------------------------------------------------------------------------
def fold_conv_bn(w, conv_bias, gamma, beta, rm, rv, eps, kind):
    """-> (packed bf16 operand, fp32 bias) for the inference path; kind "1x1" or "g3x3" """
    with torch.no_grad():
        scale = gamma.float() / torch.sqrt(rv.float() + eps)
        shift = beta.float() - rm.float() * scale
        if conv_bias is not None:
            shift = shift + conv_bias.float() * scale
        wf = (w.float() * scale.view(-1, 1, 1, 1)).contiguous()
        if kind == "g3x3":
            c = wf.shape[0]
            wk = torch.empty((c, 9 * 64), device=wf.device, dtype=BF16)
            wd = torch.empty((c, 9 * 64), device=wf.device, dtype=BF16)
            lib().call("hn_gconv_pack_diag", ptr(wf), ptr(wk), ptr(wd), c)
            return wk, shift.contiguous()
        cout, cin = wf.shape[0], wf.shape[1]
        wp = torch.empty((cout, kp32(cin)), device=wf.device, dtype=BF16)
        lib().call("hn_pack_weight", ptr(wf), ptr(wp), None, cout, cin, 1)
        return wp, shift.contiguous()


def conv_infer(x, packed, bias, cout, kind, stride, act, res=None):
    """act(conv(x) + bias [+ res]) with folded-BatchNorm operands: one launch"""
    n, hi, wi, cin = x.shape
    ho, wo = (hi, wi) if stride == 1 else (hi // 2, wi // 2)
    if kind == "g3x3":
        assert stride == 1 and res is None
        out, _, _ = k_gemm_nt(x, None, 5, (n, ho, wo), packed, cout, 64, 9, bias=bias, act=act)
    else:
        out, _, _ = k_gemm_nt(x, None, 0 if stride == 1 else 1, (n, ho, wo), packed, cout, kp32(cin), 1, bias=bias, act=act, addend=res,
                              add_pre=res is not None)
    return out


def se_gate_infer(b, w1, b1, w2, b2):
    """SE squeeze / excite for the inference path: per-image channel sums (one pass), the MLP fed by the partial rows, then b * gate"""
    n, h, w, c = b.shape
    hw, m = h * w, n * h * w
    cs = w1.shape[0]
    dev = b.device
    rb = lib().query("hn_fused_row_block", m, c, hw, 0, 1)
    if hw % rb:
        rb = hw
    pr = m // rb
    ps = torch.empty((pr, c), device=dev, dtype=F32)
    pq = torch.empty((pr, c), device=dev, dtype=F32)
    lib().call("hn_col_stats_fused", ptr(b), ld(b), m, c, rb, ptr(ps), ptr(pq))
    pooled = torch.empty((n, c), device=dev, dtype=F32)
    hid = torch.empty((n, cs), device=dev, dtype=F32)
    gate = torch.empty((n, c), device=dev, dtype=F32)
    lib().call("hn_se_mlp_fwd_parts", ptr(ps), hw // rb, 1.0 / hw, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(pooled), ptr(hid), ptr(gate), n, c, cs)
    out = new_act(n, h, w, c, dev)
    rb2 = lib().query("hn_fused_row_block", m, c, hw, 0, 0)
    lib().call("hn_bn_apply_fused", ptr(b), ld(b), m, c, None, None, 0, m, None, None, 0.0, 0.0, None, None, None, None, 0, ACT_NONE, ptr(out),
               ld(out), None, ptr(gate), hw, rb2)
    return out


def xblock_fusable(x, w1, stride, has_se, has_shortcut):
    """the fused node covers XBlocks with SE whose channel counts are multiples of 8 and whose output grid is a multiple of 128 pixels:
    the stride-1 identity blocks and the stride-2 first block of a stage (projection shortcut)"""
    cout, cin = w1.shape[0], w1.shape[1]
    ho, wo = x.shape[1] // stride, x.shape[2] // stride
    shape_ok = (stride == 1 and not has_shortcut and cout == cin) or (stride == 2 and has_shortcut and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0)
    return (FUSED_XBLOCK and FUSED_BN and GCONV_MFMA and x.is_cuda and has_se and shape_ok and cout % 8 == 0 and cin % 8 == 0
            and (ho * wo) % 128 == 0 and 3 * kp32(cout) * 4 <= 32768 and x.shape[0] * x.shape[1] * x.shape[2] < (1 << 31))


# --------------------------------------------------------------------------------------------------------------
# Squeeze-and-Excitation: out = b * sigmoid(W2 relu(W1 avgpool(b) + b1) + b2)        (net/anynet.py:40-48,68-69)
# pooling, the [N, C] x [C, C/4] excitation MLP (hn_se_mlp_fwd / hn_se_mlp_bwd), gating and their backward are all HIP kernels.
# --------------------------------------------------------------------------------------------------------------
class SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, w1, b1, w2, b2):
        n, h, w, c = b.shape
        hw = h * w
        cs = w1.shape[0]
        dev = b.device
        ps, _, r = k_col_stats(b, align=hw)
        pooled = k_rows_reduce(ps, n, hw // r, c, 1.0 / hw)                      # [N, C]
        hid = torch.empty((n, cs), device=dev, dtype=F32)
        gate = torch.empty((n, c), device=dev, dtype=F32)
        lib().call("hn_se_mlp_fwd", ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hid), ptr(gate), n, c, cs)
        out = new_act(n, h, w, c, dev)
        lib().call("hn_scale_rows", ptr(b), ld(b), ptr(gate), hw, ptr(out), ld(out), n * hw, c)
        ctx.save_for_backward(b, pooled, hid, gate, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, pooled, hid, gate, w1, w2 = ctx.saved_tensors
        dout = dense(dout)
        n, h, w, c = b.shape
        hw = h * w
        m = n * hw
        cs = w1.shape[0]
        dev = b.device
        r = lib().query("hn_colred_rows", m, hw)
        pr = (m + r - 1) // r
        pd = torch.empty((pr, c), device=dev, dtype=F32)
        pz = torch.empty((pr, c), device=dev, dtype=F32)
        lib().call("hn_col_dot", ptr(dout), ld(dout), ptr(b), ld(b), m, c, r, ptr(pd), ptr(pz))
        dgate = k_rows_reduce(pd, n, hw // r, c, 1.0)                            # sum_hw dout * b
        dpre2 = torch.empty((n, c), device=dev, dtype=F32)
        dpool = torch.empty((n, c), device=dev, dtype=F32)
        dpre1 = torch.empty((n, cs), device=dev, dtype=F32)
        dw1, db1 = torch.empty_like(w1), torch.empty((cs,), device=dev, dtype=F32)
        dw2, db2 = torch.empty_like(w2), torch.empty((c,), device=dev, dtype=F32)
        lib().call("hn_se_mlp_bwd", ptr(dgate), ptr(gate), ptr(hid), ptr(pooled), ptr(w1), ptr(w2), ptr(dpre2), ptr(dpre1), ptr(dpool),
                   ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), n, c, cs)
        db = new_act(n, h, w, c, dev)
        lib().call("hn_se_bwd_apply", ptr(dout), ld(dout), ptr(gate), ptr(dpool), hw, ptr(db), ld(db), m, c)
        return db, dw1, db1, dw2, db2


# --------------------------------------------------------------------------------------------------------------
# depthwise 3x3 (zero pad 1)
# --------------------------------------------------------------------------------------------------------------
def k_dwconv(x, wk, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = new_act(n, h, w, c, x.device)
    lib().call("hn_dwconv_fwd", ptr(x), ld(x), ptr(wk), ptr(out), ld(out), n, h, w, c)
    return out


def k_dwconv_wgrad(x, dz):
    n, h, w, c = x.shape
    chunks = lib().query("hn_dwconv_wgrad_blocks", n * h * ((w + 3) // 4), c)
    part = torch.empty((chunks, c * 9), device=x.device, dtype=F32)
    lib().call("hn_dwconv_wgrad", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), n, h, w, c)
    return k_rows_reduce(part, 1, chunks, c * 9).view(c, 1, 3, 3)


def k_dwconv_bwd(dz, x, wf, geom=None, want_dx=True, into=None, queue=None, weight=None):
    """depthwise 3x3 backward in one launch (+ the partial-row reduce): (dx | None, dweight [C,1,3,3]).  geom: level-packed tensors;
    into: an existing tensor dx is ADDED to (GradSlot accumulation); queue (GradQueue): the partial-row fold of the weight gradient is
    queued for `weight` instead of launched (returns None for it)."""
    c = x.shape[3]
    if geom is None:
        n, h, w, _ = x.shape
        H, W, nl, align = (ctypes.c_int * 1)(h), (ctypes.c_int * 1)(w), 1, 1
        strips = n * h * ((w + 3) // 4)
    else:
        nl, H, W, _, _ = _geom_arrays(geom)
        n, align = geom[0], LEVEL_ALIGN
        strips = sum(n * hh * ((ww + 3) // 4) for hh, ww in zip(geom[1], geom[2]))
    blocks = lib().query("hn_dwconv_bwd_blocks", strips, c)
    part = torch.empty((blocks, c * 9), device=x.device, dtype=F32)
    dx = None
    if want_dx:
        dx = torch.empty_like(dz) if into is None else into
    lib().call("hn_dwconv_bwd_levels", ptr(dz), ld(dz), ptr(x), ld(x), ptr(wf), ptr(dx), ld(dx) if dx is not None else 0, ptr(part), n, c, nl,
               ctypes.addressof(H), ctypes.addressof(W), align, 0 if into is None else 1)
    if queue is not None:
        queue.add_rows(weight, part, blocks, c * 9, (c, 1, 3, 3))
        return dx, None
    return dx, k_rows_reduce(part, 1, blocks, c * 9).view(c, 1, 3, 3)


class DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        wk, wf = pack_dw_weight(weight)
        ctx.queue, ctx.wref = (cur_queue() if weight.requires_grad else None), weight
        ctx.save_for_backward(x, wf)
        return k_dwconv(x, wk)

    @staticmethod
    def backward(ctx, dout):
        x, wf = ctx.saved_tensors
        dout = dense(dout)
        return k_dwconv_bwd(dout, x, wf, want_dx=ctx.needs_input_grad[0], queue=ctx.queue, weight=ctx.wref)


# --------------------------------------------------------------------------------------------------------------
# max pools, up-sampling
# --------------------------------------------------------------------------------------------------------------
def k_maxpool(x, mode, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = new_act(n, h // 2, w // 2, c, x.device)
    lib().call("hn_maxpool_fwd", ptr(x), ld(x), ptr(out), ld(out), n, h, w, c, mode)
    return out


def k_maxpool_bwd(x, dout, mode, wscale=None, into=None, accumulate=False):
    """into: destination of x's shape (default: a new tensor); accumulate: add to what `into` already holds (GradSlot)"""
    n, h, w, c = x.shape
    dx = new_act(n, h, w, c, x.device) if into is None else into
    arg = torch.empty((n * (h // 2) * (w // 2) * c,), device=x.device, dtype=torch.uint8)
    lib().call("hn_maxpool_bwd2", ptr(x), ld(x), ptr(dout), ld(dout), ptr(dx), ld(dx), ptr(wscale), ptr(arg), n, h, w, c, mode,
               1 if accumulate else 0)
    return dx


class MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        ctx.save_for_backward(x)
        return k_maxpool(x, mode)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return k_maxpool_bwd(x, dense(dout), ctx.mode), None


# --------------------------------------------------------------------------------------------------------------
# BiFPN fusion node
# --------------------------------------------------------------------------------------------------------------
def _fuse_args(ins, modes):
    arr_p = (ctypes.c_void_p * 3)(*[ptr(t) if t is not None else None for t in ins])
    arr_l = (ctypes.c_int * 3)(*[ld(t) if t is not None else 0 for t in ins])
    arr_m = (ctypes.c_int * 3)(*modes)
    return arr_p, arr_l, arr_m


class GradSlot:
    """Where the consumers of one multi-consumer tensor meet in backward: the first consumer to run allocates `buf` and stores its
    contribution, the later ones add theirs in place inside their own kernels.  Share.backward hands `buf` to the producer."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class Share(torch.autograd.Function):
    """Identity with n aliases, one per consumer of x (a BiFPN map feeds 2-3 fusion nodes, net/bifpn.py:186-231).  Consumers that were
    given the slot accumulate their input gradient into slot.buf inside their own backward kernels and return None for this input;
    consumers that do not know about slots return a gradient as usual and it is added here with one elementwise launch.  Without this
    node the autograd engine sums the k gradients of a k-consumer tensor with k-1 separate ATen add kernels (54 per step)."""

    @staticmethod
    def forward(ctx, x, slot, n):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        buf, ctx.slot.buf = ctx.slot.buf, None
        gs = [dense(g) for g in grads if g is not None]
        if buf is None and gs:
            buf = gs.pop(0)
            if gs and buf.data_ptr() in [g.data_ptr() for g in gs]:     # (never: every consumer returns its own tensor)
                buf = buf.clone()
        while gs:                                                   # up to three addends per launch, fp32 sum, one rounding
            part, gs = gs[:3], gs[3:]
            part += [None] * (3 - len(part))
            lib().call("hn_add_n", ptr(buf), ld(buf), ptr(part[0]), ld(part[0]), ptr(part[1]), ld(part[1]) if part[1] is not None else 0,
                       ptr(part[2]), ld(part[2]) if part[2] is not None else 0, rows(buf), buf.shape[3])
        return buf, None, None


def share(x, n):
    """-> (aliases, slot) for a tensor with n consumers; n == 1: the tensor itself and no slot"""
    if n <= 1 or not x.requires_grad:
        return (x,) * max(n, 1), None
    slot = GradSlot()
    return Share.apply(x, slot, n), slot


# HN_FUSE_SUM2X2=1: the 2x2 gradient sums of up-sampled fusion inputs inside hn_fuse_bwd (quad walk) instead of 12 hn_sum2x2 launches per
# step.  Measured SLOWER (780.9 vs 783.9 img/s, same box): a thread then walks four pixels in sequence, and these launches are bound by
# their dependent-load chains, not by their count -- off.
FUSE_SUM2X2 = os.environ.get("HN_FUSE_SUM2X2", "0") == "1"


class Fuse(torch.autograd.Function):
    """out = swish(w0*T0(a) + w1*T1(b) [+ w2*T2(c)]), w = relu(p)/(sum relu(p) + 1e-4) from the raw fusion parameter p (2 or 3 values).
    slots (optional): one GradSlot | None per input -- the gradient of a slotted input is accumulated into slot.buf (see Share)."""

    @staticmethod
    def forward(ctx, praw, m0, m1, m2, a, b, c, slots=None):
        ins = [a, b, c]
        modes = [m0, m1, m2]
        n, h, wd, ch = a.shape                        # input 0 is always at the output resolution (mode 1)
        assert m0 == 1
        dev = a.device
        w = torch.empty((3,), device=dev, dtype=F32)
        out = new_act(n, h, wd, ch, dev)
        ap, al, am = _fuse_args(ins, modes)
        lib().call("hn_fuse_fwd_raw", ctypes.addressof(ap), ctypes.addressof(al), ctypes.addressof(am), ptr(praw), praw.numel(), 1e-4, ptr(w),
                   ptr(out), ld(out), n, h, wd, ch)
        ctx.modes = modes
        ctx.slots = slots if slots is not None else (None, None, None)
        ctx.queue, ctx.pref = (cur_queue() if praw.requires_grad else None), praw
        ctx.save_for_backward(praw, w, *[t for t in ins if t is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        praw, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
        rest = list(ctx.saved_tensors[2:])
        modes, slots = ctx.modes, ctx.slots
        ins = [rest.pop(0) if m else None for m in modes]
        dout = dense(dout)
        n, h, wd, ch = dout.shape
        dev = dout.device
        g = new_act(n, h, wd, ch, dev)
        # destination of every input gradient: a fresh tensor, or the slot's buffer (first consumer: allocate + store, later: accumulate)
        dst, accum = [None] * 3, [0] * 3
        for i, m in enumerate(modes):
            if not m:
                continue
            s = slots[i]
            if s is not None and s.buf is not None:
                dst[i], accum[i] = s.buf, 1
            else:
                dst[i] = new_act(*ins[i].shape, dev)
                if s is not None:
                    s.buf = dst[i]
        ap, al, am = _fuse_args(ins, modes)
        # mode 1 (same grid) and mode 2 (nearest x2 of a half-resolution map: the kernel sums its 2x2 quads itself) gradients come out of
        # the fusion kernel; mode 3 (max-pooled input) is routed by the max-pool backward below
        inside = (1, 2) if FUSE_SUM2X2 else (1,)
        dp_ = (ctypes.c_void_p * 3)(*[ptr(dst[i]) if modes[i] in inside else None for i in range(3)])
        dl = (ctypes.c_int * 3)(*[ld(dst[i]) if modes[i] in inside else 0 for i in range(3)])
        da = (ctypes.c_int * 3)(*accum)
        blocks = lib().query("hn_fuse_bwd_blocks", n, h, wd, ch)
        pw = torch.empty((blocks, 3), device=dev, dtype=F32)
        lib().call("hn_fuse_bwd", ctypes.addressof(ap), ctypes.addressof(al), ctypes.addressof(am), ptr(w), ptr(dout), ld(dout), ptr(g), ld(g),
                   ctypes.addressof(dp_), ctypes.addressof(dl), ctypes.addressof(da), ptr(pw), n, h, wd, ch)
        if ctx.queue is not None:
            dpraw = ctx.queue.add_fuse(ctx.pref, pw, blocks, 1e-4)
        else:
            dpraw = torch.empty_like(praw)
            lib().call("hn_fuse_dweights", ptr(pw), blocks, ptr(praw), praw.numel(), 1e-4, ptr(dpraw))
        for i, m in enumerate(modes):
            if m == 2 and not FUSE_SUM2X2:                         # nearest x2 of a half-res input: 2x2 sum of g
                lib().call("hn_sum2x2", ptr(g), ld(g), ptr(dst[i]), ld(dst[i]), ptr(w[i]), n, h // 2, wd // 2, ch, accum[i])
            elif m == 3:                                           # zero-pad-same max pool of a double-res input
                k_maxpool_bwd(ins[i], g, 0, wscale=w[i], into=dst[i], accumulate=bool(accum[i]))
        dins = [None if (slots[i] is not None or not modes[i]) else dst[i] for i in range(3)]
        return dpraw, None, None, None, dins[0], dins[1], dins[2], None


# --------------------------------------------------------------------------------------------------------------
# segmentation decoder block: y = act(conv3x3(reflect_pad(cat[up2(x0)|x0, x1])) + bias)
# --------------------------------------------------------------------------------------------------------------
SEG_FOLD_DIRECT = os.environ.get("HN_SEG_FOLD_DIRECT", "1") != "0"


SEG_FOLD_MIN_ELEMS = 1 << 24   # measured (step trace, N = 16): the folding epilogue wins 130 / 45 / 19 / 11 us on the 134M / 67M / 33M / 17M-element
                               # gradients and loses 2...26 us on the <= 8M-element ones (a few hundred workgroups cannot hide the extra loads)


def dgrad_fold_ok(nout, h, w, n=None):
    """shapes hn_conv3x3_dgrad_fold covers (the staged bf16 epilogue of the 64 / 128-cout tiles); with n: and where it pays"""
    return SEG_FOLD_DIRECT and nout % 8 == 0 and nout > 32 and h >= 4 and w >= 4 and (n is None or n * h * w * nout >= SEG_FOLD_MIN_ELEMS)


def k_dgrad_fold(dz, wt, n, h, w, nout, kp, phase_k, clamp, yprev):
    """dx [N,h,w,nout] = folded data gradient (* ELU'(yprev)): conv with a folding epilogue + border fix-up, no padded-grid tensor"""
    dev = dz.device
    dx = new_act(n, h, w, nout, dev)
    ring = torch.empty((n, lib().query("hn_fold_ring_rows", h, w), nout), device=dev, dtype=BF16)
    lib().call("hn_conv3x3_dgrad_fold", ptr(dz), ld(dz), dz.shape[3], n, h, w, ptr(wt), nout, kp, phase_k, clamp, ptr(dx), ld(dx),
               ptr(yprev), ld(yprev) if yprev is not None else 0, ptr(ring))
    return dx


class SegConv(torch.autograd.Function):
    """ConvBlock / Conv3x3 of the seg decoder.  Along the decoder chain every x0 is the ELU output of the previous block and has no other
    consumer, so ELU' of the previous block is applied where this block folds its data gradient (x0_is_elu: hn_seg_fold multiplies by
    ELU'(x0)) and the previous block is told that the gradient it receives is already its dz (dy_is_dz) -- one elementwise pass less per
    block."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, up, act, out_f32, x0_is_elu=False, dy_is_dz=False):
        n, h0, w0, c0 = x0.shape
        h, w = (h0 * 2, w0 * 2) if up else (h0, w0)
        cout, cin = weight.shape[0], weight.shape[1]
        wp, wt = pack_conv_weight(weight)
        y, _, _ = k_gemm_nt(x0, x1, 2, (n, h, w), wp, cout, kp32(cin), 9, bias=bias, act=act, out_f32=out_f32, up=up)
        ctx.up, ctx.act, ctx.out_f32 = up, act, out_f32
        ctx.x0_is_elu, ctx.dy_is_dz = x0_is_elu, dy_is_dz
        ctx.wt = wt
        ctx.has_x1 = x1 is not None
        ctx.save_for_backward(x0, x1, weight, y if (act == ACT_ELU and not dy_is_dz) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight, y = ctx.saved_tensors
        up = ctx.up
        n, h0, w0, c0 = x0.shape
        h, w = (h0 * 2, w0 * 2) if up else (h0, w0)
        cout, cin = weight.shape[0], weight.shape[1]
        c1 = cin - c0
        dev = x0.device
        m = n * h * w
        if ctx.out_f32:                                              # logits gradient from the loss: fp32 [N,H,W,cout] -> padded bf16
            dy = dy.contiguous()
            dz = new_act(n, h, w, pad8(cout), dev)
            lib().call("hn_cast_f32_to_bf16_pad", ptr(dy), cout, ptr(dz), pad8(cout), m, cout)
        else:
            dy = dense(dy)
            dz = k_eltwise(1, dy, y, act=ACT_ELU) if (ctx.act == ACT_ELU and not ctx.dy_is_dz) else dy
        # bias gradient (per-channel sum of dz): out of the weight-gradient launches
        if wgrad_bias_ok(2, kp32(cin)):
            dw, dbias = k_gemm_tn(x0, x1, 2, (n, h, w), dz, cout, kp32(cin), 9, cin, up=up, kh=3, want_bias=True)
        else:
            ps, _, r = k_col_stats(dz)
            dbias = k_rows_reduce(ps, 1, ps.shape[0], dz.shape[3]).view(-1)
            if dbias.numel() != cout:
                dbias = dbias[:cout] + 0.0                           # owning copy by a kernel (a clone would be a memcpy node in the graph)
            dw = k_gemm_tn(x0, x1, 2, (n, h, w), dz, cout, kp32(cin), 9, cin, up=up, kh=3)
        # data gradient on the padded (H+2)x(W+2) grid, then fold the reflection / up-sampling / concat back
        dx0 = dx1 = None
        if not up and not ctx.has_x1 and dgrad_fold_ok(c0, h, w, n):
            if ctx.needs_input_grad[0]:
                dx0 = k_dgrad_fold(dz, ctx.wt, n, h, w, c0, kp32(cout), 0, 0, x0 if ctx.x0_is_elu else None)
            return dx0, dx1, dw, dbias, None, None, None, None, None
        dvp, _, _ = k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), ctx.wt, cin, kp32(cout), 9, c0=dz.shape[3], c1=0)
        if ctx.needs_input_grad[0]:
            dx0 = new_act(n, h0, w0, c0, dev)
            yp = x0 if ctx.x0_is_elu else None
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c0, up)
        if ctx.has_x1 and ctx.needs_input_grad[1]:
            dx1 = new_act(n, h, w, c1, dev)
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), c0, ptr(dx1), ld(dx1), None, 0, n, h, w, c1, 0)
        return dx0, dx1, dw, dbias, None, None, None, None, None


# --------------------------------------------------------------------------------------------------------------
# head outputs: (optional depthwise 3x3) -> 1x1 conv + bias -> fp32, written straight into the caller's layout
# --------------------------------------------------------------------------------------------------------------
class HeadOut(torch.autograd.Function):
    """feats: list of NHWC maps (pyramid levels) sharing one (dw, pw, bias) -> fp32 [N, sum_l H_l*W_l*rep, k] (rep*k = Cout).
    Covers SeparableConvBlock(norm=False) headers of the det towers (head_detect/detection.py:24,36-44,61,73-83) and, with
    dw_weight=None, the final 1x1(+bias) of a lane branch (head_lane/lanedetect.py:49,56,63,86-92)."""

    @staticmethod
    def forward(ctx, dw_weight, pw_weight, bias, k, act, *feats):
        n = feats[0].shape[0]
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        wp, wt = pack_conv_weight(pw_weight)
        rows_total = sum(f.shape[1] * f.shape[2] for f in feats)
        rep = cout // k
        out = torch.empty((n, rows_total * rep, k), device=feats[0].device, dtype=F32)
        ldc, img_stride, ch_off = cout, rows_total * cout, 0
        mids = []
        off = 0
        for f in feats:
            _, h, w, _ = f.shape
            mid = k_dwconv(f, pack_dw_weight(dw_weight)[0]) if dw_weight is not None else f
            mids.append(mid if dw_weight is not None else None)
            dst = out.view(-1)[off * ldc + ch_off:]
            k_gemm_nt(mid, None, 0, (n, h, w), wp, cout, kp32(cin), 1, bias=bias, act=act, out=dst, out_f32=True, ldc=ldc,
                      rpi=h * w, img_stride=img_stride)
            off += h * w
        ctx.meta = (k, act, ch_off, ldc, img_stride, dw_weight is not None, len(feats))
        ctx.wt = wt
        ctx.save_for_backward(dw_weight, pw_weight, out if act == ACT_SIGMOID else None, *feats, *[t for t in mids if t is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        k, act, ch_off, ldc, img_stride, has_dw, nf = ctx.meta
        saved = ctx.saved_tensors
        dw_weight, pw_weight, yout = saved[0], saved[1], saved[2]
        feats = saved[3:3 + nf]
        mids = saved[3 + nf:] if has_dw else feats
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        dout = dout.contiguous()
        dev = dout.device
        ldz = pad8(cout)
        dpw = dbias = ddw = None                             # first level: the gradients themselves; further levels accumulate
        dfeats = []
        off = 0
        acc = lambda tot, part: part if tot is None else tot.add_(part)
        for f, mid in zip(feats, mids):
            n, h, w, _ = f.shape
            m = n * h * w
            dz = new_act(n, h, w, ldz, dev)
            base = off * ldc + ch_off
            lib().call("hn_head_grad", ptr(dout.view(-1)[base:]), ptr(yout.view(-1)[base:]) if yout is not None else None, h * w, img_stride,
                       ldc, cout, ptr(dz), ldz, m, 1 if act == ACT_SIGMOID else 0)
            dpw_l, db_l = k_gemm_tn(mid, None, 0, (n, h, w), dz, cout, kp32(cin), 1, cin, want_bias=True)
            dbias = acc(dbias, db_l)
            dpw = acc(dpw, dpw_l)
            dmid, _, _ = k_gemm_nt(dz, None, 0, (n, h, w), ctx.wt, cin, kp32(cout), 1, c0=ldz, c1=0)
            if has_dw:
                df, dwl = k_dwconv_bwd(dmid, f, pack_dw_weight(dw_weight)[1])
                ddw = acc(ddw, dwl)
                dfeats.append(df)
            else:
                dfeats.append(dmid)
            off += h * w
        return (ddw, dpw, dbias, None, None, *dfeats)


class HeadOutCat(torch.autograd.Function):
    """cat([1x1(ta; wa, ba), 1x1(tb; wb, bb)], -1) as fp32 [N, H*W, ca + cb]: both branch outputs are written side by side by their GEMM
    epilogues (channel offset + row stride), so neither the concat (head_lane/lanedetect.py:93: cat([down, up], 1)) nor the slicing of
    its gradient exist as separate kernels."""

    @staticmethod
    def forward(ctx, wa, ba, wb, bb, ta, tb):
        n, h, w, _ = ta.shape
        ca, cb = wa.shape[0], wb.shape[0]
        ldc = ca + cb
        out = torch.empty((n, h * w, ldc), device=ta.device, dtype=F32)
        wts = []
        for wgt, bias, t, off in ((wa, ba, ta, 0), (wb, bb, tb, ca)):
            wp, wt = pack_conv_weight(wgt)
            wts.append(wt)
            k_gemm_nt(t, None, 0, (n, h, w), wp, wgt.shape[0], kp32(wgt.shape[1]), 1, bias=bias, out=out.view(-1)[off:], out_f32=True,
                      ldc=ldc, rpi=h * w, img_stride=h * w * ldc)
        ctx.wts = wts
        ctx.save_for_backward(ta, tb, wa, wb)
        return out

    @staticmethod
    def backward(ctx, dout):
        ta, tb, wa, wb = ctx.saved_tensors
        n, h, w, _ = ta.shape
        ca, cb = wa.shape[0], wb.shape[0]
        ldc = ca + cb
        dout = dout.contiguous()
        dev = dout.device
        res = []
        for wgt, wt, t, off in ((wa, ctx.wts[0], ta, 0), (wb, ctx.wts[1], tb, ca)):
            cout, cin = wgt.shape[0], wgt.shape[1]
            ldz = pad8(cout)
            dz = new_act(n, h, w, ldz, dev)
            lib().call("hn_head_grad", ptr(dout.view(-1)[off:]), None, h * w, h * w * ldc, ldc, cout, ptr(dz), ldz, n * h * w, 0)
            dwgt, dbias = k_gemm_tn(t, None, 0, (n, h, w), dz, cout, kp32(cin), 1, cin, want_bias=True)
            dt, _, _ = k_gemm_nt(dz, None, 0, (n, h, w), wt, cin, kp32(cout), 1, c0=ldz, c1=0)
            res.append((dwgt, dbias, dt))
        return res[0][0], res[0][1], res[1][0], res[1][1], res[0][2], res[1][2]


# --------------------------------------------------------------------------------------------------------------
# lane-head input fusion: cat[mp(mp(P3)), mp(P4), P5, up2(P6)] with nn.MaxPool2d(3,2,1)   (head_lane/lanedetect.py:76-80)
# --------------------------------------------------------------------------------------------------------------
class LaneConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p3, p4, p5, p6):
        n, h, w, c = p5.shape
        out = new_act(n, h, w, 4 * c, p5.device)
        t3 = k_maxpool(p3, 1)
        k_maxpool(t3, 1, out=out[..., 0:c])
        k_maxpool(p4, 1, out=out[..., c:2 * c])
        k_eltwise(2, p5, alpha=1.0, out=out[..., 2 * c:3 * c])
        lib().call("hn_up2_fwd", ptr(p6), ld(p6), ptr(out[..., 3 * c:]), ld(out), n, h // 2, w // 2, c)
        ctx.save_for_backward(p3, p4, t3)
        return out

    @staticmethod
    def backward(ctx, dout):
        p3, p4, t3 = ctx.saved_tensors
        dout = dense(dout)
        n, h, w, c4 = dout.shape
        c = c4 // 4
        dt3 = k_maxpool_bwd(t3, dout[..., 0:c], 1)
        d3 = k_maxpool_bwd(p3, dt3, 1)
        d4 = k_maxpool_bwd(p4, dout[..., c:2 * c], 1)
        d5 = k_eltwise(2, dout[..., 2 * c:3 * c], alpha=1.0)
        d6 = new_act(n, h // 2, w // 2, c, dout.device)
        lib().call("hn_sum2x2", ptr(dout[..., 3 * c:]), ld(dout), ptr(d6), ld(d6), None, n, h // 2, w // 2, c, 0)
        return d3, d4, d5, d6


# --------------------------------------------------------------------------------------------------------------
# segmentation loss (weighted CE + ignore_index + top-k hardest pixels) and deploy-mode argmax
# --------------------------------------------------------------------------------------------------------------
class SegLoss(torch.autograd.Function):
    """logits: fp32 NHWC [N, H, W, C] (dense rows); target: [N, H, W] int64 or float32 class ids."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, use_top_k, top_k_ratio, ignore_index, slot=None):
        """slot (GradSlot of the producing SegOutUp node): the gradient is handed over as that node's space-to-depth bf16 operand
        (hn_seg_loss_bwd_s2d) instead of an fp32 dlogits tensor"""
        n, h, w, c = logits.shape
        hw = h * w
        ctx.slot = slot if (slot is not None and h % 2 == 0 and w % 2 == 0) else None
        ctx.hw_dims = (h, w)
        k = int(top_k_ratio * hw) if use_top_k else hw
        dev = logits.device
        ws = torch.empty((lib().query("hn_seg_loss_ws_bytes", n, hw),), device=dev, dtype=torch.uint8)
        out = torch.empty((1,), device=dev, dtype=F32)
        tf = 1 if target.dtype == torch.float32 else 0
        assert target.dtype in (torch.float32, torch.int64) and target.is_contiguous()
        lib().call("hn_seg_loss_fwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(class_weights), ignore_index, n, hw,
                   1 if use_top_k else 0, k, ptr(ws), ptr(out))
        ctx.meta = (n, hw, c, tf, 1 if use_top_k else 0, k, ignore_index)
        ctx.save_for_backward(logits, target, class_weights, ws)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, cw, ws = ctx.saved_tensors
        n, hw, c, tf, topk, k, ign = ctx.meta
        g = gout.contiguous().to(F32).view(1)
        if ctx.slot is not None and ctx.slot.buf is None:
            h, w = ctx.hw_dims
            dz = new_act(n, h // 2, w // 2, pad8(4 * c), logits.device)
            lib().call("hn_seg_loss_bwd_s2d", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), ign, n, h, w, topk, k, ptr(ws), ptr(g),
                       ptr(dz), ld(dz))
            ctx.slot.buf = dz
            return None, None, None, None, None, None, None
        dl = torch.empty_like(logits)
        lib().call("hn_seg_loss_bwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), ign, n, hw, topk, k, ptr(ws), ptr(g),
                   ptr(dl), dl.stride(2))
        return dl, None, None, None, None, None, None


def seg_loss_hip(seg_nchw, target, class_weights, use_top_k, top_k_ratio, ignore_index=255, slot=None):
    """seg_nchw: the module's "seg" output (fp32, NCHW-shaped view of NHWC memory).  slot: GradSlot of the SegOutUp node that produced
    exactly this tensor (HydraNet passes it when its own cal_loss consumes its own "seg" output)."""
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits, slot = logits.contiguous(), None
    return SegLoss.apply(logits, target.contiguous(), class_weights, use_top_k, top_k_ratio, ignore_index, slot)


class SegFocalLoss(torch.autograd.Function):
    """focal variant of the seg loss (head_seg/segmentation_loss.py:31-46): logits fp32 NHWC [N, H, W, C] (dense rows), target [N, H, W]
    int64 or float32 class ids; mean over all pixels"""

    @staticmethod
    def forward(ctx, logits, target, class_weights, gamma, alpha):
        n, h, w, c = logits.shape
        hw = h * w
        dev = logits.device
        tf = 1 if target.dtype == torch.float32 else 0
        assert target.dtype in (torch.float32, torch.int64) and target.is_contiguous()
        ws = torch.empty((lib().query("hn_seg_loss_blocks", n, hw),), device=dev, dtype=F32)
        out = torch.empty((1,), device=dev, dtype=F32)
        lib().call("hn_seg_focal_fwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(class_weights), float(gamma), float(alpha), n, hw,
                   ptr(ws), ptr(out))
        ctx.meta = (n, hw, c, tf, float(gamma), float(alpha))
        ctx.save_for_backward(logits, target, class_weights)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, cw = ctx.saved_tensors
        n, hw, c, tf, gamma, alpha = ctx.meta
        g = gout.contiguous().to(F32).view(1)
        dl = torch.empty_like(logits)
        lib().call("hn_seg_focal_bwd", ptr(logits), logits.stride(2), c, ptr(target), tf, ptr(cw), gamma, alpha, n, hw, ptr(g), ptr(dl),
                   dl.stride(2))
        return dl, None, None, None, None


def seg_focal_loss_hip(seg_nchw, target, class_weights, gamma=2.0, alpha=1.0):
    """CrossEntropyLoss.forward with use_focal=True (gamma 2, alpha 1: the defaults model.py:119-124 leaves untouched)"""
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits = logits.contiguous()
    return SegFocalLoss.apply(logits, target.contiguous(), class_weights, gamma, alpha)


def argmax_channels(seg_nchw):
    logits = seg_nchw.permute(0, 2, 3, 1)
    if not logits.is_contiguous():
        logits = logits.contiguous()
    n, h, w, c = logits.shape
    out = torch.empty((n, h, w), device=logits.device, dtype=torch.int64)
    lib().call("hn_argmax_channels", ptr(logits), logits.stride(2), c, n * h * w, ptr(out))
    return out


# --------------------------------------------------------------------------------------------------------------
# detection loss (focal BCE + smooth-L1 with IoU anchor assignment)
# --------------------------------------------------------------------------------------------------------------
class DetLoss(torch.autograd.Function):
    """returns a 2-vector (classification loss, regression loss), both batch means like FocalLoss.forward."""

    @staticmethod
    def forward(ctx, cls, reg, anchors, ann):
        n, a, k = cls.shape
        mx = ann.shape[1]
        dev = cls.device
        cls, reg, ann = cls.contiguous(), reg.contiguous(), ann.contiguous().float()
        anc = anchors.reshape(-1, 4).contiguous()
        blocks = lib().query("hn_det_loss_blocks", a)
        assign = torch.empty((n, a), device=dev, dtype=torch.int16)
        part = torch.empty((n, blocks, 3), device=dev, dtype=F32)
        npos = torch.empty((n,), device=dev, dtype=F32)
        out = torch.empty((2,), device=dev, dtype=F32)
        lib().call("hn_det_loss_fwd", ptr(cls), ptr(reg), ptr(anc), ptr(ann), n, a, k, mx, ptr(assign), ptr(part), ptr(npos), ptr(out))
        ctx.save_for_backward(cls, reg, anc, ann, assign, npos)
        return out

    @staticmethod
    def backward(ctx, gout):
        cls, reg, anc, ann, assign, npos = ctx.saved_tensors
        n, a, k = cls.shape
        dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)
        g = gout.contiguous().to(F32)
        lib().call("hn_det_loss_bwd", ptr(cls), ptr(reg), ptr(anc), ptr(ann), n, a, k, ann.shape[1], ptr(assign), ptr(npos), ptr(g), ptr(dcls),
                   ptr(dreg))
        return dcls, dreg, None, None


def det_loss_hip(classification, regression, anchors, annotations):
    out = DetLoss.apply(classification, regression, anchors, annotations)
    return out[0:1], out[1:2]


# --------------------------------------------------------------------------------------------------------------
# Level-packed det-head towers.  Regressor / Classifier apply the SAME SeparableConvBlock to the five pyramid levels and differ only
# in the per-level BatchNorm (head_detect/detection.py:20-35,57-72).  Launched level by level that is ~650 launches per step, most of
# them on 4x8 ... 16x32 maps where a launch is pure latency.  Here the levels live stacked in one [sum_l N*H_l*W_l, C] tensor: the
# depthwise conv, the pointwise GEMM (+ statistics), the BatchNorm passes and every backward kernel run ONCE for all levels, with the
# per-level BatchNorm parameters selected per row block inside the kernels.  Levels whose row count is not a multiple of 128 (640x640: P7 =
# 25 rows per image) are padded up to one ("ragged" packing): the depthwise kernel writes zeros to the alignment rows, so the pointwise conv
# output there is exactly bf16(bias) and is subtracted from the BatchNorm statistics; gradients at those rows are zero by construction.
# --------------------------------------------------------------------------------------------------------------
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


def packed_rows(geom):
    n, hs, ws = geom
    return sum(_pad_rows(n * h * w) for h, w in zip(hs, ws))


def has_pad_rows(geom):
    n, hs, ws = geom
    return any((n * h * w) % LEVEL_ALIGN for h, w in zip(hs, ws))


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def level_views(packed, geom):
    """NHWC views of the (real rows of the) levels of a packed [1, 1, rows, C] tensor; every level starts on an aligned row"""
    n, hs, ws = geom
    out, off = [], 0
    for h, w in zip(hs, ws):
        m = n * h * w
        out.append(packed[0, 0, off:off + m].view(n, h, w, packed.shape[3]))
        off += _pad_rows(m)
    return out


def k_dwconv_levels(x, wk, geom, into=None):
    """into: an existing tensor the result is ADDED to (GradSlot accumulation of a data gradient)"""
    nl, H, W, _, _ = _geom_arrays(geom)
    out = torch.empty_like(x) if into is None else into
    lib().call("hn_dwconv_fwd_levels", ptr(x), ld(x), ptr(wk), ptr(out), ld(out), geom[0], x.shape[3], nl, ctypes.addressof(H),
               ctypes.addressof(W), LEVEL_ALIGN, 0 if into is None else 1)
    return out


def k_dwconv_wgrad_levels(x, dz, geom):
    nl, H, W, _, _ = _geom_arrays(geom)
    c = x.shape[3]
    chunks = lib().query("hn_dwconv_wgrad_blocks", sum(geom[0] * hh * ((ww + 3) // 4) for hh, ww in zip(geom[1], geom[2])), c)
    part = torch.empty((chunks, c * 9), device=x.device, dtype=F32)
    lib().call("hn_dwconv_wgrad_levels", ptr(x), ld(x), ptr(dz), ld(dz), ptr(part), geom[0], c, nl, ctypes.addressof(H), ctypes.addressof(W),
               LEVEL_ALIGN)
    return k_rows_reduce(part, 1, chunks, c * 9).view(c, 1, 3, 3)


class PackLevels(torch.autograd.Function):
    """stack pyramid levels [N,H_l,W_l,C] into one [1, 1, sum rows, C] tensor (backward hands out views of the packed gradient)"""

    @staticmethod
    def forward(ctx, *feats):
        n, c = feats[0].shape[0], feats[0].shape[3]
        geom = (n, tuple(f.shape[1] for f in feats), tuple(f.shape[2] for f in feats))
        total = packed_rows(geom)
        out = torch.empty((1, 1, total, c), device=feats[0].device, dtype=BF16)
        for v, f in zip(level_views(out, geom), feats):
            k_eltwise(2, f, alpha=1.0, out=v)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(level_views(dense(g), ctx.geom))


class TowerLayer(torch.autograd.Function):
    """out = act(BN_level(pointwise(depthwise(x)) + bias)) on level-packed rows; bn = nlev x (gamma, beta, running_mean, running_var)."""

    @staticmethod
    def forward(ctx, x, dw_w, pw_w, pw_b, geom, act, eps, momentum, training, slot, *bn):
        nl, H, W, R, CNT = _geom_arrays(geom)
        ctx.slot = slot                                     # GradSlot of x (the packed map feeds both towers), see Share
        total, c = x.shape[2], x.shape[3]
        cout = pw_w.shape[0]
        dev = x.device
        wk, wf = pack_dw_weight(dw_w)
        d = k_dwconv_levels(x, wk, geom)
        wp, wt = pack_conv_weight(pw_w)
        z, psum, psq = k_gemm_nt(d, None, 0, (1, 1, total), wp, cout, kp32(c), 1, bias=pw_b, stats=training)
        gam, bet = [bn[4 * l] for l in range(nl)], [bn[4 * l + 1] for l in range(nl)]
        rms, rvs = [bn[4 * l + 2] for l in range(nl)], [bn[4 * l + 3] for l in range(nl)]
        coef = torch.empty((nl, 4, cout), device=dev, dtype=F32)
        if training:
            div = total // psum.shape[0]
            ga, ba, rma, rva = _ptr_array(gam), _ptr_array(bet), _ptr_array(rms), _ptr_array(rvs)    # keep the host arrays alive
            lib().call("hn_bn_finalize_levels", ptr(psum), ptr(psq), div, cout, nl, ctypes.addressof(R), ctypes.addressof(CNT),
                       ctypes.addressof(ga), ctypes.addressof(ba), ctypes.addressof(rma), ctypes.addressof(rva), float(eps),
                       float(momentum), ptr(pw_b), ptr(coef))
        else:
            for l in range(nl):
                lib().call("hn_bn_eval_coeff", ptr(gam[l]), ptr(bet[l]), ptr(rms[l]), ptr(rvs[l]), float(eps), cout, ptr(coef[l, 0]),
                           ptr(coef[l, 1]))
        out = torch.empty((1, 1, total, cout), device=dev, dtype=BF16)
        lib().call("hn_bn_act_levels", ptr(z), ld(z), ptr(coef), act, ptr(out), ld(out), cout, nl, ctypes.addressof(R))
        ctx.geom, ctx.act, ctx.training = geom, act, training
        ctx.has_bias = pw_b is not None
        ctx.packs = (wf, wt)
        ctx.queue, ctx.wrefs = (cur_queue() if (training and dw_w.requires_grad and pw_w.requires_grad) else None), (dw_w, pw_w)
        ctx.save_for_backward(x, d, z, coef, pw_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, d, z, coef, pw_w = ctx.saved_tensors
        assert ctx.training, "backward through eval-mode BatchNorm is not part of the hot path"
        geom, act = ctx.geom, ctx.act
        nl, H, W, R, CNT = _geom_arrays(geom)
        wf, wt = ctx.packs
        dout = dense(dout)
        total, c = x.shape[2], x.shape[3]
        cout = z.shape[3]
        dev = z.device
        r = lib().query("hn_colred_rows", total, 128)
        pr = total // r
        pg = torch.empty((pr, cout), device=dev, dtype=F32)
        pgx = torch.empty((pr, cout), device=dev, dtype=F32)
        lib().call("hn_bn_bwd_reduce_levels", ptr(dout), ld(dout), ptr(z), ld(z), None, 0, ptr(coef), act, cout, r, nl, ctypes.addressof(R),
                   ptr(pg), ptr(pgx))
        red = torch.empty((nl, 2, cout), device=dev, dtype=F32)
        # one owning tensor per parameter gradient: autograd's AccumulateGrad clones views before storing them in .grad
        dgam = [torch.empty((cout,), device=dev, dtype=F32) for _ in range(nl)]
        dbet = [torch.empty((cout,), device=dev, dtype=F32) for _ in range(nl)]
        dga, dba = _ptr_array(dgam), _ptr_array(dbet)
        dbias = torch.empty((cout,), device=dev, dtype=F32) if ctx.has_bias else None    # a bias feeding BatchNorm has zero gradient
        lib().call("hn_bn_bwd_finalize_levels", ptr(pg), ptr(pgx), r, cout, nl, ctypes.addressof(R), ctypes.addressof(CNT),
                   ctypes.addressof(dga), ctypes.addressof(dba), ptr(red), ptr(dbias))
        dz = torch.empty_like(z)
        lib().call("hn_bn_bwd_apply_levels", ptr(dout), ld(dout), ptr(z), ld(z), None, 0, ptr(coef), ptr(red), act, ptr(dz), ld(dz), cout, nl,
                   ctypes.addressof(R))
        dd, _, _ = k_gemm_nt(dz, None, 0, (1, 1, total), wt, c, kp32(cout), 1)
        q_, (dw_ref, pw_ref) = ctx.queue, ctx.wrefs
        if q_ is not None:
            dpw = q_.add_gemm(pw_ref, d, dz, 0, (1, 1, total), c, cout)
        else:
            dpw = k_gemm_tn(d, None, 0, (1, 1, total), dz, cout, kp32(c), 1, c)
        dx = None
        if ctx.needs_input_grad[0] and ctx.slot is not None:
            sl = ctx.slot
            if sl.buf is None:
                sl.buf, ddw = k_dwconv_bwd(dd, x, wf, geom, queue=q_, weight=dw_ref)
            else:
                _, ddw = k_dwconv_bwd(dd, x, wf, geom, into=sl.buf, queue=q_, weight=dw_ref)
        else:
            dx, ddw = k_dwconv_bwd(dd, x, wf, geom, want_dx=ctx.needs_input_grad[0], queue=q_, weight=dw_ref)
        bn_grads = []
        for l in range(nl):
            bn_grads += [dgam[l], dbet[l], None, None]
        return (dx, ddw, dpw, dbias, None, None, None, None, None, None, *bn_grads)


class HeadOutPacked(torch.autograd.Function):
    """HeadOut on a level-packed input: depthwise, data gradient and all weight gradients run once for all levels; only the pointwise
    GEMM forward (per-level output mapping into the [N, sum_l H_l*W_l*rep, k] concat) and its gradient gather stay per level."""

    @staticmethod
    def forward(ctx, dw_weight, pw_weight, bias, k, act, geom, x):
        n, hs, ws = geom
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        wp, wt = pack_conv_weight(pw_weight)
        wk, wf = pack_dw_weight(dw_weight)
        rows_total = sum(h * w for h, w in zip(hs, ws))
        rep = cout // k
        out = torch.empty((n, rows_total * rep, k), device=x.device, dtype=F32)
        ldc, img_stride = cout, rows_total * cout
        mid = k_dwconv_levels(x, wk, geom)
        off = 0
        for v, h, w in zip(level_views(mid, geom), hs, ws):
            k_gemm_nt(v, None, 0, (n, h, w), wp, cout, kp32(cin), 1, bias=bias, act=act, out=out.view(-1)[off * ldc:], out_f32=True, ldc=ldc,
                      rpi=h * w, img_stride=img_stride)
            off += h * w
        ctx.meta = (k, act, ldc, img_stride, geom)
        ctx.packs = (wf, wt)
        ctx.queue, ctx.wref = (cur_queue() if dw_weight.requires_grad else None), dw_weight
        ctx.save_for_backward(pw_weight, out if act == ACT_SIGMOID else None, x, mid)
        return out

    @staticmethod
    def backward(ctx, dout):
        k, act, ldc, img_stride, geom = ctx.meta
        n, hs, ws = geom
        pw_weight, yout, x, mid = ctx.saved_tensors
        wf, wt = ctx.packs
        cout, cin = pw_weight.shape[0], pw_weight.shape[1]
        dout = dout.contiguous()
        dev = dout.device
        ldz = pad8(cout)
        total = x.shape[2]
        # alignment rows of a ragged packing must read as zeros in the bias / weight gradient sums and in the data gradient
        dz = zeros((1, 1, total, ldz), dev, BF16) if has_pad_rows(geom) else torch.empty((1, 1, total, ldz), device=dev, dtype=BF16)
        off = 0
        for v, h, w in zip(level_views(dz, geom), hs, ws):
            base = off * ldc
            lib().call("hn_head_grad", ptr(dout.view(-1)[base:]), ptr(yout.view(-1)[base:]) if yout is not None else None, h * w, img_stride,
                       ldc, cout, ptr(v), ldz, n * h * w, 1 if act == ACT_SIGMOID else 0)
            off += h * w
        dpw, dbias = k_gemm_tn(mid, None, 0, (1, 1, total), dz, cout, kp32(cin), 1, cin, want_bias=True)
        dmid, _, _ = k_gemm_nt(dz, None, 0, (1, 1, total), wt, cin, kp32(cout), 1, c0=ldz, c1=0)
        dx, ddw = k_dwconv_bwd(dmid, x, wf, geom, queue=ctx.queue, weight=ctx.wref)
        return ddw, dpw, dbias, None, None, None, dx


# --------------------------------------------------------------------------------------------------------------
# Final seg conv in phase form.  Conv3x3(reflect-pad(nearest_up2(x))) (head_seg/segmentation.py:101-104) on the up-sampled grid reads
# every low-resolution pixel four times and, for the 5-class output layer, runs its data gradient on a 514x1026x64 padded grid.  On
# the LOW-resolution grid the same function is a 3x3 conv with replicate padding and 4*k outputs (one k-vector per output phase
# (py,px)): W_eff[(py,px,o)][c][dy][dx] = sum of the taps (ky,kx) whose up-sampled source row/col falls on low-res offset (dy,dx)
#   phase 0: ky=0 -> dy=-1, ky=1,2 -> dy=0;   phase 1: ky=0,1 -> dy=0, ky=2 -> dy=+1     (same for kx/dx)
# and reflection of the up-sampled index is exactly clamping of the low-res index.  4x fewer pixels forward, and the data gradient is
# produced directly at the producer's resolution (no full-resolution padded dgrad, fold, 2x2 sum).
# --------------------------------------------------------------------------------------------------------------
_PHASE_T = {}


def _phase_matrix(device):
    """T[(py,px,dy,dx), (ky,kx)] in {0,1}: W_eff.view(k*c, 36) = W.view(k*c, 9) @ T^T"""
    t = _PHASE_T.get(device)
    if t is None:
        a = torch.zeros(2, 3, 3)                       # a[p][d+1][k]
        a[0, 0, 0] = 1; a[0, 1, 1] = 1; a[0, 1, 2] = 1
        a[1, 1, 0] = 1; a[1, 1, 1] = 1; a[1, 2, 2] = 1
        t = torch.einsum("pdk,qel->pqdekl", a, a).reshape(36, 9).to(device)
        _PHASE_T[device] = t
    return t


class SegOutUp(torch.autograd.Function):
    """logits[N, 2h, 2w, k] (fp32) = Conv3x3(ReflectionPad2d(1)(nearest_up2(x))) + bias, x [N, h, w, c] bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, x_is_elu=False, slot=None):
        """slot: GradSlot through which the loss may deliver the gradient already in this node's space-to-depth bf16 operand form"""
        n, h, w, c = x.shape
        k = weight.shape[0]
        ctx.x_is_elu = x_is_elu
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        wp, wt, b_eff = pack_phase_weight(weight, c, bias)
        out = torch.empty((n, 2 * h, 2 * w, k), device=x.device, dtype=F32)
        # img_stride = -k: the conv epilogue scatters phase (py, px) of low-res pixel (y, x) to output pixel (2y+py, 2x+px) itself
        k_gemm_nt(x, None, 4, (n, h, w), wp, 4 * k, kp32(c), 9, bias=b_eff, out=out, out_f32=True, ldc=4 * k, img_stride=-k)
        ctx.wt = wt
        ctx.k = k
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        k = ctx.k
        dev = x.device
        ldz = pad8(4 * k)
        dz = None
        if ctx.slot is not None and ctx.slot.buf is not None:          # delivered by SegLoss.backward in operand form
            dz, ctx.slot.buf = ctx.slot.buf, None
        if dy is not None:
            d2 = new_act(n, h, w, ldz, dev)
            lib().call("hn_space_to_depth", ptr(dy.contiguous()), ptr(d2), ldz, n, h, w, k)
            dz = d2 if dz is None else k_eltwise(0, dz, d2)
        if dz is None:
            return None, None, None, None, None
        if wgrad_bias_ok(4, kp32(c)):
            dw_eff, db_eff = k_gemm_tn(x, None, 4, (n, h, w), dz, 4 * k, kp32(c), 9, c, kh=3, want_bias=True)       # [4k, c, 3, 3], [4k]
        else:
            ps, _, _ = k_col_stats(dz)
            db_eff = k_rows_reduce(ps, 1, ps.shape[0], ldz)
            dw_eff = k_gemm_tn(x, None, 4, (n, h, w), dz, 4 * k, kp32(c), 9, c, kh=3)
        dw = torch.empty((k, c, 3, 3), device=dev, dtype=F32)
        dbias = torch.empty((k,), device=dev, dtype=F32)
        lib().call("hn_phase_fold", ptr(dw_eff), None, ptr(db_eff), ptr(dw), ptr(dbias), k, c, 0)
        dx = None
        if ctx.needs_input_grad[0]:
            yp = x if ctx.x_is_elu else None
            if dgrad_fold_ok(c, h, w, n):
                dx = k_dgrad_fold(dz, ctx.wt, n, h, w, c, kp32(4 * k), 0, 1, yp)
            else:
                dvp, _, _ = k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), ctx.wt, c, kp32(4 * k), 9, c0=ldz, c1=0)
                dx = new_act(n, h, w, c, dev)
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx), ld(dx), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c, 2)
        return dx, dw, dbias, None, None


def seg_out_argmax(x, weight, bias):
    """deploy forward of the seg output layer fused with the arg-max over the classes (model/model.py:197): int64 mask [N, 2h, 2w]; the fp32
    logits are never written.  No gradient (inference only)."""
    n, h, w, c = x.shape
    k = weight.shape[0]
    wp, _, b_eff = pack_phase_weight(weight, c, bias)
    mask = torch.empty((n, 2 * h, 2 * w), device=x.device, dtype=torch.int64)
    lib().call("hn_conv3x3_out_argmax", ptr(x), n, h, w, c, ld(x), ptr(wp), k, kp32(c), ptr(b_eff), ptr(mask))
    return mask


def seg_out_argmax_ok(x, weight):
    k, c = weight.shape[0], x.shape[3]
    return x.is_cuda and not torch.is_grad_enabled() and 4 * k <= 32 and kp32(c) == 64 and (x.shape[2] * k) % 2 == 0


class SegConvUp(torch.autograd.Function):
    """y [N, 2h, 2w, k] = ELU(Conv3x3(ReflectionPad2d(1)(cat[nearest_up2(x0), x1])) + bias) in PHASE form (decoder blocks 1/3/5/7,
    head_seg/segmentation.py:92-100).  The up-sampled operand is convolved on its own low-resolution grid with the effective weights of
    SegOutUp (4 output phases, 2x2 non-zero taps each: 16 instead of 36 tap products per low-res pixel -- 2.25x fewer MACs in forward,
    data gradient and weight gradient); the skip operand x1 (already full resolution) goes through the ordinary direct 3x3 kernel and
    joins as a pre-activation addend in the phase conv's epilogue.  The data gradient w.r.t. x0 is produced directly at x0's resolution
    (no full-resolution padded grid, no 2x2 fold).  ELU' folding along the decoder chain as in SegConv (x0_is_elu / dy_is_dz)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, x0_is_elu=False, dy_is_dz=False):
        n, h, w, c0 = x0.shape
        k, cin = weight.shape[0], weight.shape[1]
        c1 = cin - c0
        dev = x0.device
        z1 = wt1 = wt_full = None
        # Per-layer choice of form (measured, tools/bench_seg.py): the forward runs full-resolution when the skip operand is so narrow that its
        # own conv would be mostly K padding (decoder.5: 24 channels); the data gradient w.r.t. x0 runs full-resolution when the padded
        # low-resolution grid cannot fill the chip (decoder.1: 18x34 cells -> 384 workgroups); the weight gradient is always in phase form.
        fwd_phase = (c1 == 0 or c1 >= 32) if SEG_FWD_PHASE is None else SEG_FWD_PHASE
        tiles = n * ((h + 2 + 15) // 16) * ((w + 2 + 15) // 16) * ((c0 + 127) // 128)
        ctx.dgrad_phase = (tiles >= SEG_DGRAD_PHASE_MIN_TILES) if SEG_DGRAD_PHASE is None else SEG_DGRAD_PHASE
        wp_eff, wt_eff, b_eff = pack_phase_weight(weight, c0, bias, want_wt=ctx.dgrad_phase)
        if c1 and (fwd_phase or ctx.dgrad_phase):
            wp1, wt1 = pack_conv_weight_slice(weight, c0, c1)
        if not fwd_phase or not ctx.dgrad_phase:
            wp_full, wt_full = pack_conv_weight(weight)
        if fwd_phase:
            if c1:
                z1, _, _ = k_gemm_nt(x1, None, 2, (n, 2 * h, 2 * w), wp1, k, kp32(c1), 9)      # skip operand: plain reflect-pad 3x3, no bias / act
            y = new_act(n, 2 * h, 2 * w, k, dev)
            lib().call("hn_conv3x3_phase", ptr(x0), 4, n, h, w, c0, ld(x0), ptr(wp_eff), 4 * k, kp32(c0), ptr(b_eff), ACT_ELU, ptr(y), ld(y), k,
                       ptr(z1), ld(z1) if z1 is not None else 0)
        else:
            y, _, _ = k_gemm_nt(x0, x1, 2, (n, 2 * h, 2 * w), wp_full, k, kp32(cin), 9, bias=bias, act=ACT_ELU, up=1)
        ctx.x0_is_elu, ctx.dy_is_dz = x0_is_elu, dy_is_dz
        ctx.packs = (wt_eff, wt1, wt_full)
        ctx.save_for_backward(x0, x1, y if not dy_is_dz else None)
        ctx.dims = (k, c0, c1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, y = ctx.saved_tensors
        wt_eff, wt1, wt_full = ctx.packs
        k, c0, c1 = ctx.dims
        n, h, w, _ = x0.shape
        dev = x0.device
        dy = dense(dy)
        dz = dy if ctx.dy_is_dz else k_eltwise(1, dy, y, act=ACT_ELU)
        # space-to-depth gradient: the operand of both low-resolution contractions
        dzs = new_act(n, h, w, 4 * k, dev)
        lib().call("hn_space_to_depth_bf16", ptr(dz), ld(dz), ptr(dzs), n, h, w, k, None)
        # effective-weight gradient (zeros at the taps a phase does not use), mapped back to the 3x3 weights by the phase matrix; the bias
        # gradient (channel sums of dz, per phase) comes out of the same launches
        splits, rps, wsb = ctypes.c_int(), ctypes.c_long(), ctypes.c_long()
        lib().query("hn_wgrad_plan_phase", n, h, w, 4 * k, kp32(c0), k, ctypes.addressof(splits), ctypes.addressof(rps), ctypes.addressof(wsb))
        ws = torch.empty((wsb.value // 4,), device=dev, dtype=F32)
        dw_eff = torch.empty((4 * k, c0, 3, 3), device=dev, dtype=F32)
        db_eff = torch.empty((4 * k,), device=dev, dtype=F32)
        lib().call("hn_conv_gemm_tn_phase", ptr(x0), n, h, w, c0, ld(x0), ptr(dzs), ld(dzs), 4 * k, kp32(c0), k, ptr(ws), ptr(dw_eff), ptr(db_eff))
        dx0 = dx1 = None
        if not ctx.dgrad_phase:
            # full-resolution data gradient for both operands at once (padded (2h+2) x (2w+2) grid), folded back per operand
            dvp, _, _ = k_gemm_nt(dz, None, 3, (n, 2 * h + 2, 2 * w + 2), wt_full, c0 + c1, kp32(k), 9, c0=k, c1=0)
            if ctx.needs_input_grad[0]:
                dx0 = new_act(n, h, w, c0, dev)
                yp = x0 if ctx.x0_is_elu else None
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, 2 * h, 2 * w, c0, 1)
            if c1 and ctx.needs_input_grad[1]:
                dx1 = new_act(n, 2 * h, 2 * w, c1, dev)
                lib().call("hn_seg_fold", ptr(dvp), ld(dvp), c0, ptr(dx1), ld(dx1), None, 0, n, 2 * h, 2 * w, c1, 0)
        elif ctx.needs_input_grad[0] and dgrad_fold_ok(c0, h, w, n):
            dx0 = k_dgrad_fold(dzs, wt_eff, n, h, w, c0, kp32(4 * k), k, 1, x0 if ctx.x0_is_elu else None)
        elif ctx.needs_input_grad[0]:
            dvp = new_act(n, h + 2, w + 2, c0, dev)
            lib().call("hn_conv3x3_phase", ptr(dzs), 3, n, h + 2, w + 2, 4 * k, ld(dzs), ptr(wt_eff), c0, kp32(4 * k), None, ACT_NONE, ptr(dvp),
                       ld(dvp), k, None, 0)
            dx0 = new_act(n, h, w, c0, dev)
            yp = x0 if ctx.x0_is_elu else None
            lib().call("hn_seg_fold", ptr(dvp), ld(dvp), 0, ptr(dx0), ld(dx0), ptr(yp), ld(yp) if yp is not None else 0, n, h, w, c0, 2)
        dw1 = None
        if c1:
            dw1 = k_gemm_tn(x1, None, 2, (n, 2 * h, 2 * w), dz, k, kp32(c1), 9, c1, kh=3)
            if ctx.dgrad_phase and ctx.needs_input_grad[1]:
                dvp1, _, _ = k_gemm_nt(dz, None, 3, (n, 2 * h + 2, 2 * w + 2), wt1, c1, kp32(k), 9, c0=k, c1=0)
                dx1 = new_act(n, 2 * h, 2 * w, c1, dev)
                lib().call("hn_seg_fold", ptr(dvp1), ld(dvp1), 0, ptr(dx1), ld(dx1), None, 0, n, 2 * h, 2 * w, c1, 0)
        # effective-weight gradient mapped back to the 3x3 taps (the transpose of the phase map), joined with the skip operand's part
        dw = torch.empty((k, c0 + c1, 3, 3), device=dev, dtype=F32)
        dbias = torch.empty((k,), device=dev, dtype=F32)
        lib().call("hn_phase_fold", ptr(dw_eff), ptr(dw1), ptr(db_eff), ptr(dw), ptr(dbias), k, c0, c1)
        return dx0, dx1, dw, dbias, None, None


SEG_PHASE_UP = os.environ.get("HN_SEG_PHASE_UP", "1") != "0"
SEG_DGRAD_PHASE_MIN_TILES = int(os.environ.get("HN_SEG_DGRAD_PHASE_MIN_TILES", "448"))
SEG_FWD_PHASE = None        # None: per-layer heuristic; True / False force the forward form (tests)
SEG_DGRAD_PHASE = None      # the same for the data gradient w.r.t. the up-sampled operand


def seg_up_phase_ok(x0, x1, weight):
    """phase form needs 64-aligned output channels (a cout tile / K chunk must lie inside one phase)"""
    return SEG_PHASE_UP and x0.is_cuda and weight.shape[0] % 64 == 0 and x0.shape[3] % 8 == 0


# --------------------------------------------------------------------------------------------------------------
# Lane losses on the device (head_lane/lanedetect_loss.py:18-78): one workgroup does the OHEM classification loss (log-softmax, counts,
# radix select of the k-th smallest background log-prob instead of torch.sort/topk, both sums); the location loss is a row kernel + a
# one-block finalize.  ~5 launches instead of ~60 tiny torch ops, and no memcpy nodes in the captured step.
# --------------------------------------------------------------------------------------------------------------
class LaneClsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_targets, cls_preds, negative_ratio, alpha):
        tgt = cls_targets.reshape(-1, 2).float().contiguous()
        z = cls_preds.reshape(-1, 2).float().contiguous()
        m = z.shape[0]
        dev = z.device
        lsm = torch.empty((m, 2), device=dev, dtype=F32)
        pmask = torch.empty((m,), device=dev, dtype=torch.uint8)
        out = torch.empty((2,), device=dev, dtype=F32)
        aux = torch.empty((4,), device=dev, dtype=F32)
        lib().call("hn_lane_cls_loss_fwd", ptr(z), ptr(tgt), m, float(negative_ratio), float(alpha), ptr(lsm), ptr(pmask), ptr(out), ptr(aux))
        ctx.alpha, ctx.shape = float(alpha), cls_preds.shape
        ctx.save_for_backward(lsm, pmask, aux)
        ctx.mark_non_differentiable(pmask, aux)
        return out[0], out[1], pmask, aux

    @staticmethod
    def backward(ctx, gpos, gneg, _gm, _ga):
        lsm, pmask, aux = ctx.saved_tensors
        m = lsm.shape[0]
        dz = torch.empty((m, 2), device=lsm.device, dtype=F32)
        gp = gpos.reshape(1).to(F32) if gpos is not None else zeros((1,), lsm.device)
        gn = gneg.reshape(1).to(F32) if gneg is not None else zeros((1,), lsm.device)
        lib().call("hn_lane_cls_loss_bwd", ptr(lsm), ptr(pmask), ptr(aux), ptr(gp), ptr(gn), ctx.alpha, m, ptr(dz))
        return None, dz.view(ctx.shape), None, None


class LaneLocLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pmask, aux, loc_targets, loc_preds, wcol, alpha):
        L = loc_preds.shape[-1]
        if wcol + 1 >= L:
            raise IndexError(f"index {wcol + 1} is out of bounds for dimension 1 with size {L}")   # the reference's own failure mode
        p = loc_preds.reshape(-1, L).float().contiguous()
        t = loc_targets.reshape(-1, L).float().contiguous()
        m = p.shape[0]
        dev = p.device
        rowloss = torch.empty((m,), device=dev, dtype=F32)
        rownorm = torch.empty((m,), device=dev, dtype=F32)
        out = torch.empty((1,), device=dev, dtype=F32)
        lib().call("hn_lane_loc_loss_fwd", ptr(p), ptr(t), ptr(pmask), ptr(aux), m, L, int(wcol), float(alpha), ptr(rowloss), ptr(rownorm),
                   ptr(out))
        ctx.meta = (int(wcol), float(alpha), loc_preds.shape)
        ctx.save_for_backward(p, t, pmask, rownorm, aux)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        p, t, pmask, rownorm, aux = ctx.saved_tensors
        wcol, alpha, shape = ctx.meta
        m, L = p.shape
        dp = torch.empty((m, L), device=p.device, dtype=F32)
        g = gout.reshape(1).to(F32)
        lib().call("hn_lane_loc_loss_bwd", ptr(p), ptr(t), ptr(pmask), ptr(rownorm), ptr(aux), ptr(g), m, L, wcol, alpha, ptr(dp))
        return None, None, None, dp.view(shape), None, None


def lane_cls_loss_hip(cls_targets, cls_preds, negative_ratio=15, alpha=10.0):
    """cal_loss_cls (lanedetect_loss.py:18-54): returns (pos, neg, pmask, positive_num) like the reference; pmask / positive_num are
    device-side handles (byte mask, aux vector) consumed by lane_loc_loss_hip."""
    pos, neg, pmask, aux = LaneClsLoss.apply(cls_targets, cls_preds, negative_ratio, alpha)
    return pos, neg, pmask, aux


def lane_loc_loss_hip(pmask, positive_num, loc_targets, loc_preds, alpha=10.0, points_per_line=160):
    """cal_loss_regress (lanedetect_loss.py:57-78) incl. its hard-coded points_per_line = 160 default (x10 weights on columns 160/161)."""
    return LaneLocLoss.apply(pmask, positive_num, loc_targets, loc_preds, points_per_line, alpha)


# --------------------------------------------------------------------------------------------------------------
# HydraTrainer.cal_total_loss (model/train.py:192-203) as one launch forward and one backward instead of ~22 scalar torch kernels.
# --------------------------------------------------------------------------------------------------------------
class WeightedLossSum(torch.autograd.Function):
    """total = sum_g (sum_{i in g} x_i * w_i) * gw_g; meta = (w tuple, gw tuple, group-id tuple), xs = fp32 scalar device tensors"""

    @staticmethod
    def forward(ctx, meta, *xs):
        w, gw, grp = meta
        n = len(xs)
        ctx.shapes = [x.shape for x in xs]
        xs = [x.reshape(1) for x in xs]
        assert all(x.dtype == F32 and x.is_cuda for x in xs)
        out = torch.empty((1,), device=xs[0].device, dtype=F32)
        ctx.host = (_ptr_array(xs), (ctypes.c_float * n)(*w), (ctypes.c_float * len(gw))(*gw), (ctypes.c_int * n)(*grp), n)
        pa, wa, ga, ia, _ = ctx.host
        lib().call("hn_weighted_sum", ctypes.addressof(pa), ctypes.addressof(wa), ctypes.addressof(ga), ctypes.addressof(ia), n, None, ptr(out), None)
        ctx.keep = xs                                     # the pointer table refers to these
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        pa, wa, ga, ia, n = ctx.host
        grads = torch.empty((n,), device=gout.device, dtype=F32)
        g = gout.reshape(1)
        if g.dtype != F32:
            g = g.float()
        lib().call("hn_weighted_sum", ctypes.addressof(pa), ctypes.addressof(wa), ctypes.addressof(ga), ctypes.addressof(ia), n, ptr(g), None, ptr(grads))
        return (None, *[grads[i:i + 1].view(shape) for i, shape in enumerate(ctx.shapes)])


def weighted_loss_sum(groups):
    """groups = [(group weight, [(loss tensor, weight), ...]), ...] -> the reference's total loss, same association order"""
    xs, w, gw, grp = [], [], [], []
    for gi, (gweight, terms) in enumerate(groups):
        gw.append(float(gweight))
        for x, wi in terms:
            xs.append(x)
            w.append(float(wi))
            grp.append(gi)
    return WeightedLossSum.apply((tuple(w), tuple(gw), tuple(grp)), *xs)
