"""per-parameter gradient difference of the big cfg's backbone: BN3_PARTS_FROM_DGRAD on vs off, next to the noise floor of another
summation-order change (EPILOGUE_STATS on vs off)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multitask_hydranet_amd import HydraNet, ops as K
from tests.helpers import load_cfg

cfgs = load_cfg("hydranet_big.yml")
torch.manual_seed(6)
net = HydraNet(cfgs).cuda().train()
x = torch.randn(2, 3, 512, 1024, device="cuda")


def run(**sw):
    for k, v in sw.items():
        setattr(K, k, v)
    net.zero_grad(set_to_none=True)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    feats = net._backbone(x)
    net._flush_nbt()
    sum((f.float() ** 2).mean() for f in feats).backward()
    g = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    net.load_state_dict(sd)
    return g


base = run(BN3_PARTS_FROM_DGRAD=False, EPILOGUE_STATS=True)
a = run(BN3_PARTS_FROM_DGRAD=True, EPILOGUE_STATS=True)
b = run(BN3_PARTS_FROM_DGRAD=False, EPILOGUE_STATS=False)
c = run(BN3_PARTS_FROM_DGRAD=False, EPILOGUE_STATS=True)
for k in base:
    if "conv_block_1.0.weight" in k or "stem" in k:
        r = float(base[k].abs().max())
        print("%-60s bn3 %.2e  epi %.2e  rerun %.2e" % (k, float((a[k] - base[k]).abs().max()) / r, float((b[k] - base[k]).abs().max()) / r,
                                                       float((c[k] - base[k]).abs().max()) / r))
