"""a dependent chain shaped like the 1x1 convs of a backbone stage: GEMM (rows x C x C, statistics epilogue) -> fused BatchNorm apply, repeated
(each GEMM reads what the previous kernel wrote, as in the step); eager launches for rocprofv3 --pmc passes.
usage: python tools/gemm_chain.py [stage=4] [pairs=10]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multitask_hydranet_amd import ops as K
dev = "cuda:0"
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m, c = {4: (2048, 936), 3: (8192, 376), 2: (32768, 152)}[stage]
torch.manual_seed(0)
x = torch.randn(1, 1, m, c, device=dev).bfloat16()
wgt = torch.randn(c, c, 1, 1, device=dev) * c ** -0.5
wp, _ = K.pack_conv_weight(wgt)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
t = x
for _ in range(pairs):
    z, ps, pq = K.k_gemm_nt(t, None, 0, (1, 1, m), wp, c, K.kp32(c), 1, stats=True)
    t, _, _, _ = K.k_bn_apply_fused(z, ps, pq, m, gam, bet, 1e-5, 0.1, rm, rv, K.ACT_RELU, training=True)
torch.cuda.synchronize()
print("stage", stage, "rows", m, "channels", c, "pairs", pairs, "checksum", float(t.float().abs().mean()))
