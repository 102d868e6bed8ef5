"""kernel-trace probe of the SE MLP launches (run under rocprofv3 --kernel-trace --stats)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd._lib import lib
dev = torch.device("cuda:0")
n = 16
for c in (152, 368, 936):
    cs = c // 4
    sw1, sb1, sw2, sb2 = torch.randn(cs, c, device=dev) * 0.05, torch.zeros(cs, device=dev), torch.randn(c, cs, device=dev) * 0.05, torch.zeros(c, device=dev)
    pooled, hid, gt = torch.rand(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)
    d2, d1, dp = torch.empty(n, c, device=dev), torch.empty(n, cs, device=dev), torch.empty(n, c, device=dev)
    g1, gb1, g2, gb2 = torch.empty_like(sw1), torch.empty_like(sb1), torch.empty_like(sw2), torch.empty_like(sb2)
    for _ in range(20):
        lib().call('hn_se_mlp_fwd', pooled.data_ptr(), sw1.data_ptr(), sb1.data_ptr(), sw2.data_ptr(), sb2.data_ptr(), hid.data_ptr(), gt.data_ptr(), n, c, cs)
        lib().call('hn_se_mlp_bwd', pooled.data_ptr(), gt.data_ptr(), hid.data_ptr(), pooled.data_ptr(), sw1.data_ptr(), sw2.data_ptr(), d2.data_ptr(), d1.data_ptr(), dp.data_ptr(), g1.data_ptr(), gb1.data_ptr(), g2.data_ptr(), gb2.data_ptr(), n, c, cs)
    torch.cuda.synchronize()
