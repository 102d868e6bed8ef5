"""micro-benchmark of the grouped deferred weight-gradient launch (hn_wgrad_group) on a backbone-stage-like job list; HN_DBG = bits of
hn_debug_knob(9): 1 skip stores, 2 skip MFMAs, 4 skip loads"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
dev = "cuda:0"
def stage(name):
    if name == "stage4":
        return [((16, 8, 16), 376, 936, 1), ((16, 16, 32), 376, 936, 0)] + [((16, 8, 16), 936, 936, 0)] * 27
    if name == "stage3":
        return [((16, 16, 32), 152, 376, 1), ((16, 32, 64), 152, 376, 0)] + [((16, 16, 32), 376, 376, 0)] * 19
    if name == "stage2":
        return [((16, 32, 64), 64, 152, 1), ((16, 64, 128), 64, 152, 0)] + [((16, 32, 64), 152, 152, 0)] * 7
    raise KeyError(name)
for name in sys.argv[1:] or ["stage4", "stage3", "stage2"]:
    jobs = stage(name)
    group = K.WgradGroup()
    ws = []
    flop = 0
    for (n, h, w), cin, cout, mode in jobs:
        hi, wi = (2 * h, 2 * w) if mode == 1 else (h, w)
        x = torch.randn(n, hi, wi, cin, device=dev).bfloat16()
        dz = torch.randn(n, h, w, cout, device=dev).bfloat16()
        wgt = torch.empty(cout, cin, 1, 1, device=dev)
        ws.append((wgt, x, dz, mode, (n, h, w), cin, cout))
        flop += 2.0 * n * h * w * cin * cout
    def run():
        group.weights = tuple(w[0] for w in ws)
        for w in ws:
            group.add(*w)
        return group.flush()
    if os.environ.get("HN_ONCE"):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        continue
    for spec in os.environ.get("HN_DBG", "0").split(","):
        variant, dbg = (int(v) for v in spec.split(":")) if ":" in spec else (0, int(spec))
        lib().query("hn_debug_knob", 9, dbg)
        lib().query("hn_debug_knob", 10, variant)
        run(); torch.cuda.synchronize()
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            run()
        torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                out = run()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        print("%s variant=%d dbg=%d: %.1f us per flush, %.0f TFLOP/s (%d jobs, %.1f GFLOP)" % (name, variant, dbg, us, flop / us / 1e6, len(jobs), flop / 1e9), flush=True)
    lib().query("hn_debug_knob", 9, 0)
    lib().query("hn_debug_knob", 10, 0)
