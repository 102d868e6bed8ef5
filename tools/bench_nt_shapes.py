"""1x1-conv GEMM launches (NT forward/dgrad form and TN wgrad form) at the shapes the network uses; optional forced tile / ring depth.
usage: python tools/bench_nt_shapes.py [bc r]"""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import sys, torch
sys.path.insert(0, '.')
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfgs = [(0, 0)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
if len(sys.argv) == 2 and sys.argv[1] == 'sweep':
    cfgs = [(0, 0), (0, 2), (0, 3), (0, 4), (64, 2), (64, 3), (128, 2), (128, 3), (32, 2), (32, 3)]
shapes = [(2097152, 32, 24), (524288, 24, 24), (524288, 24, 64), (131072, 64, 64), (131072, 64, 152), (32768, 152, 152), (8192, 376, 376),
          (2048, 936, 936), (131072, 112, 112), (32768, 112, 112), (8192, 112, 112), (2048, 112, 112), (512, 112, 112), (8192, 448, 448)]
def timeit(f, iters=20):
    """device time per launch: the launches are captured in one hipGraph (no host launch overhead between them)"""
    for _ in range(2): f()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        f()
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): f()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3
for bc, r in cfgs:
    lib().query("hn_debug_nt_config", bc, r)
    print(f"== forced bc={bc} r={r}")
    for (m, k, n) in shapes:
        x = torch.randn(1, 1, m, k, device=dev).bfloat16()
        w = torch.randn(n, k, 1, 1, device=dev) * 0.05
        wp, wt = K.pack_conv_weight(w)
        try:
            out = torch.empty(1, 1, m, n, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: K.k_gemm_nt(x, None, 0, (1, 1, m), wp, n, K.kp32(k), 1, out=out, stats=True))
            ref = (x.view(m, k).float() @ w.view(n, k).bfloat16().float().t()).view(1, 1, m, n)
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            gb = (m * k + m * n) * 2 / 1e9
            print(f"M={m:8d} K={k:4d} N={n:4d}  {t:7.1f} us  {gb / t * 1e6 / 1e3:6.2f} TB/s  {2.0 * m * k * n / t / 1e6:7.1f} TF/s  err {err:.1e}")
        except Exception as e:
            print(f"M={m} K={k} N={n} failed: {e}")
lib().query("hn_debug_nt_config", 0, 0)
