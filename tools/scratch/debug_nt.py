import torch, sys
sys.path.insert(0, '.')
from multitask_hydranet_amd import ops as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
def run(M, cin, nout, f32, tag):
    x = torch.randn(1, 1, M, cin, device=dev).bfloat16()
    w = torch.randn(nout, cin, 1, 1, device=dev) * 0.2
    wp, wt = K.pack_conv_weight(w)
    for rep in range(3):
        out, _, _ = K.k_gemm_nt(x, None, 0, (1, 1, M), wp, nout, K.kp32(cin), 1, out_f32=f32)
        torch.cuda.synchronize()
        ref = x[0, 0].float() @ w.view(nout, cin).bfloat16().float().t()
        err = (out.view(M, nout).float() - ref).abs()
        bad = (err > 0.05).nonzero()
        print(tag, M, cin, nout, f32, 'rep', rep, 'maxerr', float(err.max()), 'nbad', bad.shape[0], 'first bad', bad[:5].tolist())
run(48, 8, 16, False, 'dgrad-like')
run(48, 16, 2, True, 'head')
run(300, 64, 5, True, 'seg-like-1x1')
run(300, 32, 8, False, 'c8')
run(1000, 64, 16, False, 'c16')
run(1000, 96, 24, False, 'c24')
