import torch, sys
sys.path.insert(0, '.')
from multitask_hydranet_amd import ops as K
dev = torch.device('cuda:0')
torch.manual_seed(0)
def run(M, nout, cin, label):
    x = torch.zeros(1, 1, M, cin, device=dev)
    for m in range(M):
        for c in range(cin):
            x[0, 0, m, c] = (m % 64) + c / 64.0
    x = x.bfloat16()
    res = {}
    for (m0, c0) in [(0, 0), (1, 0), (5, 3), (17, 2), (33, 1), (40, 7)]:
        if m0 >= M or c0 >= nout: continue
        dz = torch.zeros(1, 1, M, nout, device=dev, dtype=torch.bfloat16)
        dz[0, 0, m0, c0] = 1.0
        dw = K.k_gemm_tn(x, None, 0, (1, 1, M), dz, nout, K.kp32(cin), 1, cin)
        torch.cuda.synchronize()
        d = dw.view(nout, cin)
        nz = d.abs().sum(1).nonzero().flatten().tolist()
        print(label, "m0,c0", (m0, c0), "nonzero out rows", nz, "row vals[:4]", d[nz[0]][:4].tolist() if nz else None, "expect", x[0,0,m0,:4].float().tolist())
run(64, 16, 32, "A")
run(64, 64, 64, "B")
run(128, 128, 128, "C")
# random check
for (M, nout, cin) in [(64, 16, 32), (200, 40, 24), (512, 128, 128)]:
    x = torch.randn(1, 1, M, cin, device=dev).bfloat16()
    dz = torch.randn(1, 1, M, K.pad8(nout), device=dev).bfloat16()
    dz[..., nout:] = 0
    dw = K.k_gemm_tn(x, None, 0, (1, 1, M), dz, nout, K.kp32(cin), 1, cin).view(nout, cin)
    ref = dz[0, 0, :, :nout].float().t() @ x[0, 0].float()
    print("rand", M, nout, cin, "err", float((dw - ref).abs().max()), "ref", float(ref.abs().max()))
