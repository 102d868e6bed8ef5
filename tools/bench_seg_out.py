"""Seg output conv (64 -> 4 x 5 phase couts on the 256 x 512 low-resolution grid, N = 16): the persistent weights-in-registers launch (round 6)
against the direct kernel (hn_debug_knob(11, 3): tuning build), logits bit for bit, microseconds per launch; and the arg-max form at the
inference shape (N = 32, 576 x 960 low-res)."""
import os, sys, torch
os.environ["HN_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd.ops import seg as S
from multitask_hydranet_amd._lib import lib
from tools.bench_fused import timeit

dev = torch.device("cuda:0")
for N, h, w in ((16, 256, 512), (32, 576, 960)):
    x = torch.randn(N, h, w, 64, device=dev).to(torch.bfloat16)
    wo = torch.randn(5, 64, 3, 3, device=dev) * 0.05
    bo = torch.randn(5, device=dev) * 0.1
    wp, wt, be = S.pack_phase_weight(wo, 64, bo)
    res = {}
    for knob, name in ((3, "direct kernel"), (0, "persistent")):
        lib().query("hn_debug_knob", 11, knob)
        if N == 16:
            out = torch.empty((N, 2 * h, 2 * w, 5), device=dev, dtype=torch.float32)
            fn = lambda: K.k_gemm_nt(x, None, 4, (N, h, w), wp, 20, K.kp32(64), 9, bias=be, out=out, out_f32=True, ldc=20, img_stride=-5)
        else:
            out = torch.empty((N, 2 * h, 2 * w), device=dev, dtype=torch.int64)
            fn = lambda: lib().call("hn_conv3x3_out_argmax", x.data_ptr(), N, h, w, 64, 64, wp.data_ptr(), 5, K.kp32(64), be.data_ptr(), out.data_ptr())
        fn()
        torch.cuda.synchronize()
        res[name] = (out.clone(), timeit(fn, reps=5, iters=5))
    lib().query("hn_debug_knob", 11, 0)
    same = torch.equal(res["direct kernel"][0], res["persistent"][0])
    print(f"N={N} {h}x{w} ({'logits' if N == 16 else 'arg-max'}): direct {res['direct kernel'][1]:.1f} us, persistent {res['persistent'][1]:.1f} us, identical: {same}", flush=True)
