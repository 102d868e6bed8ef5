"""Seg output conv (64 -> 4 x 5 phase couts on the 256 x 512 low-resolution grid, N = 16): the persistent weights-in-registers launch (round 6)
against the direct kernel (hn_debug_knob(11, 3): tuning build), logits bit for bit, microseconds per launch; the arg-max form at the
inference shape (N = 32, 576 x 960 low-res); and the last decoder block's phase-form conv (64 -> 4 x 64, N = 16, 128 x 256 low-res) in its
persistent one-phase-per-workgroup form against the direct kernel."""
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

N, h, w, c0, k = 16, 128, 256, 64, 64
x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
wt = torch.randn(k, c0, 3, 3, device=dev) * 0.02
bias = torch.randn(k, device=dev) * 0.1
T = K._phase_matrix(dev)
w_eff = (wt.reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
wpe, wte = K.pack_conv_weight(w_eff)
be = bias.repeat(4)
res = {}
for knob, name in ((3, "direct kernel"), (0, "persistent")):
    lib().query("hn_debug_knob", 11, knob)
    out = torch.empty(N, 2 * h, 2 * w, k, device=dev, dtype=torch.bfloat16)
    fn = lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, N, h, w, c0, c0, wpe.data_ptr(), 4 * k, K.kp32(c0), be.data_ptr(), 3,
                            out.data_ptr(), k, k, None, 0)
    fn()
    torch.cuda.synchronize()
    res[name] = (out.clone(), timeit(fn, reps=5, iters=5))
lib().query("hn_debug_knob", 11, 0)
print(f"decoder.7 phase forward N={N} {h}x{w}: direct {res['direct kernel'][1]:.1f} us, persistent {res['persistent'][1]:.1f} us, "
      f"identical: {torch.equal(res['direct kernel'][0], res['persistent'][0])}", flush=True)
