"""micro-benchmark of the fused BatchNorm passes / XF GEMM at the backbone's stage shapes (N = 16, 512x1024), timed as hipGraph replays"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib

dev = torch.device("cuda:0")
SHAPES = {"s0": (16, 128, 256, 24), "s1": (16, 64, 128, 64), "s2": (16, 32, 64, 152), "s3": (16, 16, 32, 376), "s4": (16, 8, 16, 936),
          "p3": (16, 64, 128, 112), "p5": (16, 16, 32, 112)}


def timeit(fn, reps=20, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
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


def main():
    which = sys.argv[1:] or ["apply", "bwd", "gemm"]
    for name, (n, h, w, c) in SHAPES.items():
        m = n * h * w
        z = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
        res = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
        dout = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        out = torch.empty_like(z)
        if "apply" in which:
            # old pair
            ps, pq, _ = K.k_col_stats(z)
            def old():
                coef = K.k_bn_finalize(ps, pq, m, gamma, beta, 1e-5, 0.1, rm, rv)
                K.k_bn_act(z, coef, 1, res=res, out=out)
            t_old = timeit(old)
            coef = torch.empty(4, c, device=dev)
            line = f"{name} M={m} C={c}: finalize+bn_act {t_old:6.1f} us | fused apply (P, RB -> us):"
            for P in (16, 64, 256):
                psum, psq = torch.randn(P, c, device=dev), torch.rand(P, c, device=dev) * m
                for rb in sorted({32, 64, 128, 256, 512, 1024, 2048, max(16, m // 512)}):
                    if rb > m or (m + rb - 1) // rb > 4096:
                        continue
                    def f():
                        lib().call("hn_bn_apply_fused", z.data_ptr(), c, m, c, psum.data_ptr(), psq.data_ptr(), P, m, gamma.data_ptr(), beta.data_ptr(),
                                   1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), coef.data_ptr(), res.data_ptr(), c, 1, out.data_ptr(), c, None, None, 0, rb)
                    line += f" ({P},{rb})={timeit(f):.1f}"
            print(line, flush=True)
        if "bwd" in which:
            coef = torch.rand(4, c, device=dev)
            def oldb():
                K.bn_backward(dout, z, out, coef, 1, m, want_g=True)
            t_old = timeit(oldb)
            line = f"{name} bwd old (3-4 launches) {t_old:6.1f} us | fused reduce+apply (RB -> us):"
            t_new = timeit(lambda: K.bn_backward_fused(dout, z, out, coef, 1, m, want_g=True))
            line += f" policy pair {t_new:.1f} (rb_r {lib().query('hn_fused_row_block', m, c, 0, 0, 1)}, rb_a {lib().query('hn_fused_row_block', m, c, 0, 64, 0)})"
            print(line, flush=True)
        if "gemm" in which and name in ("s2", "s3", "s4"):
            wgt = torch.randn(c, c, 1, 1, device=dev) * c ** -0.5
            wp, wt = K.pack_conv_weight(wgt)
            sc, sh = torch.rand(c, device=dev), torch.rand(c, device=dev)
            gate = torch.rand(n, c, device=dev)
            t0 = timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, stats=True, out=out))
            t1 = timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, stats=True, out=out, xform=(sc, sh, gate, h * w, 1)))
            t2 = timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, stats=True, out=out, xform=(sc, sh, None, h * w, 0)))
            t3 = timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, out=out))
            line = f"{name} gemm 1x1 {c}->{c}: DMA+stats {t0:.1f} us, DMA no stats {t3:.1f}, XF relu+gate {t1:.1f} us, XF scale only {t2:.1f} us | ring depth"
            for r in (3, 4):
                lib().query("hn_debug_nt_config", 0, r)
                line += f" R={r}: {timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, out=out)):.1f}"
            lib().query("hn_debug_nt_config", 0, 0)
            for bc in (128,):
                lib().query("hn_debug_nt_config", bc, 0)
                line += f" | 128x128 tile: {timeit(lambda: K.k_gemm_nt(z, None, 0, (n, h, w), wp, c, K.kp32(c), 1, out=out)):.1f}"
            lib().query("hn_debug_nt_config", 0, 0)
            # weight gradient (TN GEMM + reduce) and the grouped conv trio
            dz = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
            line += f" | wgrad tn+reduce {timeit(lambda: K.k_gemm_tn(z, None, 0, (n, h, w), dz, c, K.kp32(c), 1, c)):.1f}"
            for cfg in ((128, 128, 4), (128, 128, 2), (64, 64, 1), (64, 64, 2), (64, 64, 4), (128, 64, 2), (128, 64, 4), (64, 128, 2)):
                lib().query("hn_debug_tn_config", *cfg)
                try:
                    line += f" tn{cfg}={timeit(lambda: K.k_gemm_tn(z, None, 0, (n, h, w), dz, c, K.kp32(c), 1, c)):.1f}"
                except Exception as e:
                    line += f" tn{cfg}=ERR"
            lib().query("hn_debug_tn_config", 0, 0, 0)
            w2 = torch.randn(c, 8, 3, 3, device=dev) * 0.1
            wk2, wd2 = K.pack_gconv_diag(w2)
            line += f" | gconv fwd+stats {timeit(lambda: K.k_gemm_nt(z, None, 5, (n, h, w), wk2, c, 64, 9, stats=True, out=out)):.1f}"
            line += f" fwd {timeit(lambda: K.k_gemm_nt(z, None, 5, (n, h, w), wk2, c, 64, 9, out=out)):.1f}"
            line += f" wgrad {timeit(lambda: K.k_gemm_tn(z, None, 5, (n, h, w), dz, c, 64, 9, 8, kh=3)):.1f}"
            # SE MLP
            cs = c // 4
            sw1, sb1, sw2, sb2 = torch.randn(cs, c, device=dev) * 0.05, torch.zeros(cs, device=dev), torch.randn(c, cs, device=dev) * 0.05, torch.zeros(c, device=dev)
            pooled, hid, gt = torch.rand(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)
            line += f" | se_mlp fwd {timeit(lambda: lib().call('hn_se_mlp_fwd', pooled.data_ptr(), sw1.data_ptr(), sb1.data_ptr(), sw2.data_ptr(), sb2.data_ptr(), hid.data_ptr(), gt.data_ptr(), n, c, cs)):.1f}"
            d2, d1, dp = torch.empty(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)
            g1, gb1, g2, gb2 = torch.empty_like(sw1), torch.empty_like(sb1), torch.empty_like(sw2), torch.empty_like(sb2)
            line += f" bwd {timeit(lambda: lib().call('hn_se_mlp_bwd', pooled.data_ptr(), gt.data_ptr(), hid.data_ptr(), pooled.data_ptr(), sw1.data_ptr(), sw2.data_ptr(), d2.data_ptr(), d1.data_ptr(), dp.data_ptr(), g1.data_ptr(), gb1.data_ptr(), g2.data_ptr(), gb2.data_ptr(), n, c, cs)):.1f}"
            print(line, flush=True)


main()
