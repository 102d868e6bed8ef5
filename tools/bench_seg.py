"""micro-benchmark of the seg-decoder convs (N = 16, 512x1024): direct 3x3 kernel double-buffer vs software-pipelined, full-res vs phase form"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
from tools.bench_fused import timeit

dev = torch.device("cuda:0")
N = 16
# (name, c0, c1, k, low-res h, w, up)
LAYERS = [("d1", 512, 112, 512, 16, 32, 1), ("d2", 512, 0, 256, 32, 64, 0), ("d3", 256, 112, 256, 32, 64, 1), ("d4", 256, 0, 128, 64, 128, 0),
          ("d5", 128, 24, 128, 64, 128, 1), ("d6", 128, 0, 64, 128, 256, 0), ("d7", 64, 0, 64, 128, 256, 1)]
for name, c0, c1, k, h, w, up in LAYERS:
    H, W = (2 * h, 2 * w) if up else (h, w)
    x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(N, H, W, c1, device=dev).to(torch.bfloat16) if c1 else None
    wt = torch.randn(k, c0 + c1, 3, 3, device=dev) * 0.02
    bias = torch.zeros(k, device=dev)
    wp, wtt = K.pack_conv_weight(wt)
    out = torch.empty(N, H, W, k, device=dev, dtype=torch.bfloat16)
    dz = torch.randn(N, H, W, k, device=dev).to(torch.bfloat16)
    flops = 2.0 * N * H * W * k * (c0 + c1) * 9
    line = f"{name} {c0}+{c1}->{k} @{H}x{W}: {flops/1e9:.0f} GFLOP |"
    for pipe in (0, 1):
        lib().query("hn_debug_direct_pipe", pipe)
        t = timeit(lambda: K.k_gemm_nt(x0, x1, 2, (N, H, W), wp, k, K.kp32(c0 + c1), 9, bias=bias, act=3, out=out, up=up), reps=5, iters=5)
        line += f" fwd pipe{pipe} {t:.0f} us ({flops/t/1e6:.0f} TF)"
        dvp = torch.empty(N, H + 2, W + 2, c0 + c1, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: K.k_gemm_nt(dz, None, 3, (N, H + 2, W + 2), wtt, c0 + c1, K.kp32(k), 9, c0=k, c1=0, out=dvp), reps=5, iters=5)
        line += f" dgrad {t:.0f}"
        if up and k % 64 == 0:
            T = K._phase_matrix(dev)
            w_eff = (wt[:, :c0].reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
            wpe, wte = K.pack_conv_weight(w_eff)
            be = bias.repeat(4)
            z1 = None
            t1 = 0.0
            if c1:
                wp1, wt1 = K.pack_conv_weight(wt[:, c0:].contiguous())
                z1 = torch.empty(N, H, W, k, device=dev, dtype=torch.bfloat16)
                t1 = timeit(lambda: K.k_gemm_nt(x1, None, 2, (N, H, W), wp1, k, K.kp32(c1), 9, out=z1), reps=5, iters=5)
            t2 = timeit(lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, N, h, w, c0, c0, wpe.data_ptr(), 4 * k, K.kp32(c0), be.data_ptr(), 3,
                                           out.data_ptr(), k, k, z1.data_ptr() if z1 is not None else None, k), reps=5, iters=5)
            dzs = torch.randn(N, h, w, 4 * k, device=dev).to(torch.bfloat16)
            dvpl = torch.empty(N, h + 2, w + 2, c0, device=dev, dtype=torch.bfloat16)
            t3 = timeit(lambda: lib().call("hn_conv3x3_phase", dzs.data_ptr(), 3, N, h + 2, w + 2, 4 * k, 4 * k, wte.data_ptr(), c0, K.kp32(4 * k), None, 0,
                                           dvpl.data_ptr(), c0, k, None, 0), reps=5, iters=5)
            line += f" | phase fwd x1 {t1:.0f} + x0 {t2:.0f} ({flops/(t1+t2)/1e6:.0f} TF alg) dgrad-x0 {t3:.0f}"
        line += " ||"
    print(line, flush=True)
lib().query("hn_debug_direct_pipe", 0)
