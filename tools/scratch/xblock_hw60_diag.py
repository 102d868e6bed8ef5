"""fused XBlock node vs the unfused composition vs an fp32 torch restatement at hw = 60 (a test shape that fails on sw1 only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from multitask_hydranet_amd import ops as K
dev = "cuda:0"
c, n, h, w = 376, 8, 6, 10
if len(sys.argv) > 4: c, n, h, w = [int(v) for v in sys.argv[1:5]]
gen = torch.Generator(device=dev).manual_seed(c)
rn = lambda *s, scale=1.0: torch.randn(*s, device=dev, generator=gen) * scale
cs = c // 4
prm = dict(w1=rn(c, c, 1, 1, scale=c ** -0.5), w2=rn(c, 8, 3, 3, scale=72 ** -0.5), w3=rn(c, c, 1, 1, scale=c ** -0.5),
           sw1=rn(cs, c, 1, 1, scale=c ** -0.5), sb1=rn(cs, scale=0.1), sw2=rn(c, cs, 1, 1, scale=cs ** -0.5), sb2=rn(c, scale=0.1))
bn0 = [(torch.rand(c, device=dev, generator=gen) + 0.5, rn(c, scale=0.1), rn(c, scale=0.1), torch.rand(c, device=dev, generator=gen) + 0.5) for _ in range(3)]
x0 = torch.relu(rn(n, h, w, c)).to(torch.bfloat16)
up = rn(n, h, w, c).to(torch.bfloat16)
res = {}
for mode in ("unfused", "fused", "torch"):
    p = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    bn = [[t.clone() for t in b] for b in bn0]
    for b in bn:
        b[0].requires_grad_(True); b[1].requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    K.clear_pack_cache()
    if mode == "fused":
        out = K.XBlockFn.apply(x, p["w1"], *bn[0], p["w2"], *bn[1], p["sw1"], p["sb1"], p["sw2"], p["sb2"], p["w3"], *bn[2], 1e-5, 0.1, True)
        out.backward(up)
    elif mode == "unfused":
        a = K.conv_bn_act(x, p["w1"], None, (*bn[0], None), act=K.ACT_RELU)
        b_ = K.conv_bn_act(a, p["w2"], None, (*bn[1], None), kind="g3x3", stride=1, act=K.ACT_RELU)
        b_ = K.SEGate.apply(b_, p["sw1"], p["sb1"], p["sw2"], p["sb2"])
        out = K.conv_bn_act(b_, p["w3"], None, (*bn[2], None), res=x, act=K.ACT_RELU)
        out.backward(up)
    else:
        xf = x0.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        bnf = lambda z, i: F.batch_norm(z, None, None, bn[i][0], bn[i][1], True, 0.1, 1e-5)
        a = F.relu(bnf(F.conv2d(xf, p["w1"].to(torch.bfloat16).float()), 0))
        b_ = F.relu(bnf(F.conv2d(a, p["w2"].to(torch.bfloat16).float(), padding=1, groups=c // 8), 1))
        g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(b_.mean((2, 3), keepdim=True), p["sw1"], p["sb1"])), p["sw2"], p["sb2"]))
        out = F.relu(bnf(F.conv2d(b_ * g, p["w3"].to(torch.bfloat16).float()), 2) + xf)
        out.backward(up.float().permute(0, 3, 1, 2))
        out = out.permute(0, 2, 3, 1)
    grads = {k: v.grad.clone().float() for k, v in p.items()}
    grads.update({f"bn{i}_{j}": bn[i][j].grad.clone().float() for i in range(3) for j in range(2)})
    res[mode] = dict(out=out.detach().float(), grads=grads)
cos = lambda u, v: float(F.cosine_similarity(u.flatten(), v.flatten(), dim=0))
for a, b in (("fused", "unfused"), ("fused", "torch"), ("unfused", "torch")):
    print(a, "vs", b, "out rel", float((res[a]["out"] - res[b]["out"]).abs().max() / res[b]["out"].abs().max()))
    for k in res[a]["grads"]:
        u, v = res[a]["grads"][k], res[b]["grads"][k]
        print("   %-6s cos %.5f  rel %.4f  |u| %.4g |v| %.4g" % (k, cos(u, v), float((u - v).abs().max() / v.abs().max().clamp(min=1e-20)), float(u.norm()), float(v.norm())))
