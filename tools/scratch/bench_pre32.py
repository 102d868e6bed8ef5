"""direct 3x3 conv, 64-cout tiles: streaming double buffer (64-channel chunks) vs one round trip per 32-channel chunk (knob 18 = 2)"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
dev = torch.device("cuda:0")
N = 16


def timeit(fn, reps=5, iters=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


LAYERS = [("d6", 128, 0, 64, 128, 256, 0), ("d7", 64, 0, 64, 128, 256, 1), ("d4as64", 256, 0, 64, 64, 128, 0), ("d5", 128, 24, 64, 64, 128, 1)]
for name, c0, c1, k, h, w, up in LAYERS:
    H, W = (2 * h, 2 * w) if up else (h, w)
    x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, W, c1, device=dev).to(torch.bfloat16) if c1 else None
    wt = torch.randn(k, c0 + c1, 3, 3, device=dev) * 0.02
    bias = torch.zeros(k, device=dev)
    wp, wtt = K.pack_conv_weight(wt)
    outs = []
    line = f"{name} {c0}+{c1}->{k} @{H}x{W}:"
    for knob in (0, 2):
        lib().call("hn_debug_knob", 18, knob)
        out = torch.empty(N, H, W, k, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: K.k_gemm_nt(x0, x1, 2, (N, H, W), wp, k, K.kp32(c0 + c1), 9, bias=bias, act=3, out=out, up=up))
        outs.append(out.float().clone())
        line += f"  knob18={knob}: fwd {t:.0f} us"
    lib().call("hn_debug_knob", 18, 0)
    err = float((outs[0] - outs[1]).abs().max()), float(outs[0].abs().max())
    print(line, " | max diff", err, flush=True)
