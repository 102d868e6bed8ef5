"""Micro-benchmark of single implicit-GEMM launches (seg decoder.4: 256->128 3x3 reflect @64x128, N=16) for counter collection."""
import sys, torch
sys.path.insert(0, '.')
from multitask_hydranet_amd import ops as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
n, h, w, cin, cout = 16, 64, 128, 256, 128
x = torch.randn(n, h, w, cin, device=dev).bfloat16()
wgt = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
wp, wt = K.pack_conv_weight(wgt)
dz = torch.randn(n, h, w, cout, device=dev).bfloat16()
def timeit(f, iters=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
fl = 2.0 * n * h * w * cout * cin * 9
if which in ('all', 'nt'):
    t = timeit(lambda: K.k_gemm_nt(x, None, 2, (n, h, w), wp, cout, K.kp32(cin), 9, act=K.ACT_ELU))
    print('NT fwd   %.1f us  %.1f TF/s' % (t * 1e3, fl / t / 1e9))
if which in ('all', 'dg'):
    t = timeit(lambda: K.k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), wt, cin, K.kp32(cout), 9))
    print('NT dgrad %.1f us  %.1f TF/s' % (t * 1e3, fl / t / 1e9))
if which in ('all', 'tn'):
    t = timeit(lambda: K.k_gemm_tn(x, None, 2, (n, h, w), dz, cout, K.kp32(cin), 9, cin, kh=3))
    print('TN wgrad %.1f us  %.1f TF/s' % (t * 1e3, fl / t / 1e9))
if which in ('all', 'tn1'):
    x1 = torch.randn(n, h, w, 128, device=dev).bfloat16()
    t = timeit(lambda: K.k_gemm_tn(x1, None, 0, (n, h, w), dz, cout, 128, 1, 128))
    print('TN 1x1 128x128 M=131072 %.1f us  %.1f TF/s' % (t * 1e3, 2.0 * n * h * w * 128 * 128 / t / 1e9))
