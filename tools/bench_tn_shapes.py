"""1x1 weight-gradient launches (gemm_tn + wgrad_reduce) at the shapes of the small layers; sweep tile / split count."""
import os as _os; _os.environ.setdefault("HN_TUNING", "1")   # hn_debug_* hooks: tuning build of the library
import sys, torch
sys.path.insert(0, '.')
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
def timeit(f, iters=20):
    for _ in range(2): f()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): f()
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
shapes = [(2048, 936, 936), (8192, 376, 376), (8192, 368, 368), (32768, 152, 152), (2048, 936, 232), (2048, 368, 936)]
cfgs = [(0, 0, 0), (64, 64, 3), (64, 64, 4), (64, 64, 6), (64, 64, 8), (64, 64, 12), (64, 64, 16), (64, 64, 32), (128, 64, 4), (128, 64, 8), (128, 64, 16), (64, 128, 8)]
print("shape".ljust(24), " ".join(f"{a}x{b}/{c}".rjust(10) for a, b, c in cfgs))
for (m, k, n) in shapes:
    x = torch.randn(1, 1, m, k, device=dev).bfloat16()
    dz = torch.randn(1, 1, m, n, device=dev).bfloat16()
    ref = dz.view(m, n).float().t() @ x.view(m, k).float()
    line = []
    for bc, bn, sp in cfgs:
        lib().query("hn_debug_tn_config", bc, bn, sp)
        try:
            dw = K.k_gemm_tn(x, None, 0, (1, 1, m), dz, n, K.kp32(k), 1, k)
            err = (dw.view(n, k) - ref).abs().max().item() / ref.abs().max().item()
            t = timeit(lambda: K.k_gemm_tn(x, None, 0, (1, 1, m), dz, n, K.kp32(k), 1, k))
            line.append(f"{t:7.1f}{'!' if err > 2e-2 else ' '}")
        except Exception as e:
            line.append("   fail ")
    print(str((m, k, n)).ljust(24), " ".join(v.rjust(10) for v in line), flush=True)
lib().query("hn_debug_tn_config", 0, 0, 0)
