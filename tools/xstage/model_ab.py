"""backbone forward + backward with and without the persistent stage launches: per-parameter gradient agreement (tools/, GPU)"""
import os, sys
import torch, yaml
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multitask_hydranet_amd import HydraNet, ops as K
dev = torch.device("cuda:0")
cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = 512, 1024
res = {}
modes = [("persistent", True, True), ("fwd_only", True, False), ("chain", False, False)]
for name, on, bwd in modes:
    K.XSTAGE = on
    K.XSTAGE_BWD = bwd
    K.clear_pack_cache()
    torch.manual_seed(0)
    net = HydraNet(cfgs).to(dev).train()
    with torch.no_grad():                               # the zero-init-residual conditioning of tests/helpers.conditioned_state
        for k, p in net._idx.items():
            if k.endswith("conv_block_3.1.weight"):
                p.mul_(0.1)
    gen = torch.Generator(device="cpu").manual_seed(1)
    img = torch.randn(16, 3, 512, 1024, generator=gen).to(dev)
    feats = net._backbone(img)
    gen2 = torch.Generator(device="cpu").manual_seed(2)
    loss = sum((f.float() * torch.randn(f.shape, generator=gen2).to(dev)).mean() for f in feats)      # a well-conditioned upstream gradient
    loss.backward()
    res[name] = (float(loss), {k: p.grad.detach().float().clone() for k, p in net._idx.items() if p.grad is not None and k.startswith("backbone.")})
    print(name, "loss", float(loss), flush=True)
cos = lambda a, b: float(F.cosine_similarity(a.flatten(), b.flatten(), dim=0))
ref = res["chain"][1]
for name in ("fwd_only", "persistent"):
    g = res[name][1]
    keys = [k for k in ref if ref[k].numel() >= 64]
    cs = sorted((cos(g[k], ref[k]), k) for k in keys)
    print(f"== {name} vs chain: min {cs[0][0]:.3f} p10 {cs[len(cs)//10][0]:.3f} median {cs[len(cs)//2][0]:.3f}")
    for k in ("backbone.net.stem.conv.weight", "backbone.net.stage_2.blocks.block_1.conv_block_1.0.weight", "backbone.net.stage_3.blocks.block_0.conv_block_1.0.weight",
              "backbone.net.stage_3.blocks.block_5.conv_block_1.0.weight", "backbone.net.stage_3.blocks.block_5.conv_block_2.0.weight",
              "backbone.net.stage_3.blocks.block_5.conv_block_3.0.weight", "backbone.net.stage_3.blocks.block_5.conv_block_1.1.weight",
              "backbone.net.stage_3.blocks.block_5.se.1.weight", "backbone.net.stage_3.blocks.block_5.se.3.weight",
              "backbone.net.stage_4.blocks.block_0.conv_block_1.0.weight", "backbone.net.stage_4.blocks.block_12.conv_block_1.0.weight",
              "backbone.net.stage_4.blocks.block_13.conv_block_3.0.weight", "backbone.net.stage_4.blocks.block_13.conv_block_1.1.weight",
              "backbone.net.stage_4.blocks.block_13.conv_block_3.1.bias", "backbone.net.stage_4.blocks.block_13.se.3.weight",
              "backbone.net.stage_4.blocks.block_1.conv_block_1.0.weight"):
        if k in ref:
            print(f"   {cos(g[k], ref[k]):.4f}  {float(g[k].norm()):.3e} vs {float(ref[k].norm()):.3e}  {k}")
