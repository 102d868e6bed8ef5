"""micro-benchmark of the SE excite + gated apply at the backbone's stage shapes (N = 16, 512x1024), timed as hipGraph replays of the chain
first layer -> (second layer -> apply | hn_se_gate_apply): workgroup shapes of the one-launch form (knobs 16 / 17) against the two launches"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib

dev = torch.device("cuda:0")


def timeit(fn, reps=20, iters=10):
    """us per call of fn inside a replayed hipGraph of `reps` dependent calls"""
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


SHAPES = {"s0": (16, 128, 256, 24, 6), "s1": (16, 64, 128, 64, 16), "s2": (16, 32, 64, 152, 38), "s3": (16, 16, 32, 376, 94), "s4": (16, 8, 16, 936, 234),
          "s2@640": (16, 40, 40, 152, 38), "s3@640": (16, 20, 20, 376, 94), "s4@640": (16, 10, 10, 936, 234),
          "s2@infer": (32, 72, 120, 152, 38), "s3@infer": (32, 36, 60, 376, 94), "s4@infer": (32, 18, 30, 936, 234)}
P = lambda t: t.data_ptr()


def main():
    for name, (n, h, w, c, cs) in SHAPES.items():
        hw, m = h * w, n * h * w
        z = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
        out = torch.empty_like(z)
        coef = torch.rand(4, c, device=dev)
        w1, b1 = torch.randn(cs, c, device=dev) / c ** 0.5, torch.zeros(cs, device=dev)
        w2, b2 = torch.randn(c, cs, device=dev) / cs ** 0.5, torch.zeros(c, device=dev)
        rb = lib().query("hn_fused_row_block", m, c, hw, 0, 0)
        S = hw // rb
        pool = torch.rand(n * S, c, device=dev)
        pooled, hid, gate = torch.empty(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)

        def two():
            lib().call("hn_se_mlp_fwd_parts", P(pool), S, 1.0 / hw, P(w1), P(b1), P(w2), P(b2), P(pooled), P(hid), P(gate), n, c, cs)
            lib().call("hn_bn_apply_fused", P(z), c, m, c, None, None, 0, m, None, None, 0.0, 0.0, None, None, P(coef), None, 0, 1, P(out), c, None,
                       P(gate), hw, rb)

        def one():
            lib().call("hn_se_mlp_fwd_parts", P(pool), S, 1.0 / hw, P(w1), P(b1), None, None, P(pooled), P(hid), None, n, c, cs)
            lib().call("hn_se_gate_apply", P(z), c, P(coef), 1, P(hid), P(w2), P(b2), P(gate), P(out), c, n, hw, c, cs)

        def fc1():
            lib().call("hn_se_mlp_fwd_parts", P(pool), S, 1.0 / hw, P(w1), P(b1), None, None, P(pooled), P(hid), None, n, c, cs)

        t1 = timeit(fc1)
        # backward MLP (two launches), fed by the partial rows of the gate gradient a GEMM epilogue leaves: one per 64-row tile
        sb = max(1, min(16, hw // 64))
        pdot = torch.randn(n * sb, c, device=dev)
        dpre2, dpre1, dpool = torch.empty(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)
        dw1, db1, dw2, db2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)
        gate.uniform_(0.2, 0.8); hid.uniform_(-1, 1); pooled.uniform_(0, 1)
        tb = timeit(lambda: lib().call("hn_se_mlp_bwd_parts", P(pdot), sb, P(gate), P(hid), P(pooled), P(w1), P(w2), P(dpre2), P(dpre1), P(dpool),
                                       None, None, None, None, n, c, cs))
        line = f"{name} M={m} C={c} Cs={cs}: first layer (S {S}) {t1:5.1f} | backward MLP (S {sb}) {tb:5.1f} | + second layer + apply (rb {rb}) {timeit(two) - t1:5.1f} | + gate_apply (cw, RB):"
        for cw in (32,):
            if cw == 160 and c > 160:
                continue
            for rbk in (0,):
                lib().call("hn_debug_knob", 16, cw)
                lib().call("hn_debug_knob", 17, rbk)
                line += f" ({cw},{rbk})={timeit(one) - t1:.1f}"
        lib().call("hn_debug_knob", 16, 0)
        lib().call("hn_debug_knob", 17, 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
