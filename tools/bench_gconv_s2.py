"""Stride-2 grouped 3x3 conv kernels (first block of every backbone stage) at the bench shapes: forward, data gradient, weight gradient.
Each timed as 20 back-to-back launches (HIP events) after a 256 MB cache flush; bytes = algorithmic operand bytes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_hydranet_amd._lib import lib
from multitask_hydranet_amd import ops as K

dev = torch.device("cuda:0")
SHAPES = [(16, 256, 512, 24), (16, 128, 256, 64), (16, 64, 128, 152), (16, 32, 64, 376), (16, 16, 32, 936)]


def timeit(fn, reps=20):
    flush = torch.empty(64 << 20, device=dev)
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for n, hi, wi, c in SHAPES:
    ho, wo = hi // 2, wi // 2
    x = torch.randn(n, hi, wi, c, device=dev).bfloat16()
    dz = torch.randn(n, ho, wo, c, device=dev).bfloat16()
    w = torch.randn(c, 8, 3, 3, device=dev) * 0.1
    wk, wd = K.pack_gconv_weight(w, 0)
    z = torch.empty(n, ho, wo, c, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(n, hi, wi, c, device=dev, dtype=torch.bfloat16)
    chunks = lib().query("hn_wgrad_chunks", n * ho * wo, (c // 8) * 9)
    part = torch.empty(chunks, c * 72, device=dev)
    p = K.ptr
    t_f = timeit(lambda: lib().call("hn_gconv_fwd", p(x), c, p(wd), p(z), c, n, hi, wi, c, 2))
    t_d = timeit(lambda: lib().call("hn_gconv_dgrad_s2", p(dz), c, p(wk), p(dx), c, n, hi, wi, c))
    t_w = timeit(lambda: lib().call("hn_gconv_wgrad", p(x), c, p(dz), c, p(part), n, hi, wi, c, 2))
    bx, bz = x.numel() * 2 / 1e6, dz.numel() * 2 / 1e6
    print(f"C={c:4d} {hi}x{wi}: fwd {t_f:7.1f} us ({(bx + bz) / t_f:6.2f} TB/s)  dgrad {t_d:7.1f} us ({(bx + bz) / t_d:6.2f})  "
          f"wgrad {t_w:7.1f} us ({(bx + bz) / t_w:6.2f}; {chunks} partial rows)", flush=True)
