"""hn_dwconv_bwd_levels on the step's two big shapes (level-packed det-tower tensor, one P3 map): microseconds per launch (HIP events over
replays) and effective HBM rate (dz + x read, dx written).  GPU only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multitask_hydranet_amd import ops as K

dev = torch.device("cuda:0")
c, n = 112, 16
for name, geom in (("towers (5 levels packed)", (n, [64, 32, 16, 8, 4], [128, 64, 32, 16, 8])), ("P3 map 64x128", None), ("P4 map 32x64", (n, [32], [64]))):
    wt = torch.randn(c, 1, 3, 3, device=dev) * 0.3
    _, wf = K.pack_dw_weight(wt)
    if geom is None:
        x = torch.randn(n, 64, 128, c, device=dev).bfloat16()
        dz = torch.randn_like(x)
        rows = n * 64 * 128
    else:
        rows = sum(K._pad_rows(n * h * w) for h, w in zip(geom[1], geom[2]))
        x = torch.randn(1, 1, rows, c, device=dev).bfloat16()
        dz = torch.randn_like(x)
        if len(geom[1]) == 1:
            geom = None if False else geom
    f = lambda: K.k_dwconv_bwd(dz, x, wf, geom)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            f()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:28s} {us:7.1f} us per call (incl. the ~5 us partial-row reduce)   {3 * rows * c * 2 / us / 1e6:6.2f} TB/s of dz + x + dx")
