"""Eager-mode section timing (HIP events) of one training step: where do the milliseconds go?"""
import sys, torch, yaml
sys.path.insert(0, '.')
import bench
from multitask_hydranet_amd import HydraNet
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
h, w, n = 512, 1024, 16
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
dev = torch.device('cuda:0')
net = HydraNet(cfgs).to(dev).train(); net.check_finite = False; net.lane_points_per_line = h // 8
batch = bench.synthetic_batch(cfgs, n, h, w, 1, dev)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
def run(verbose):
    net.zero_grad(set_to_none=True)
    t = [ev()]
    feats = net._backbone(batch["image"]); t.append(ev())
    fused = net._neck(feats); t.append(ev())
    seg = net._seg([feats[0], fused[0], fused[1], fused[2]]); t.append(ev())
    anchors, reg, cls = net._det(batch["image"], fused); t.append(ev())
    lane = net._lane(fused); t.append(ev())
    out = {"seg": seg, "detection": {"anchors": anchors, "regression": reg, "classification": cls}, "lane": lane}
    ld = net.cal_loss(out, batch); tot = net.total_loss(ld); t.append(ev())
    # backward in pieces: heads first (grads w.r.t. feats/fused), then neck, then backbone
    heads_in = [feats[0]] + list(fused)
    g = torch.autograd.grad(tot, heads_in, retain_graph=True, allow_unused=True); t.append(ev())
    gf = torch.autograd.grad(list(fused), feats, [x for x in g[1:]], retain_graph=True, allow_unused=True); t.append(ev())
    gin = [gf[0] + g[0]] + list(gf[1:])
    params = [p for k, p in net.named_parameters() if k.startswith('backbone')]
    torch.autograd.grad(feats, params, gin, allow_unused=True); t.append(ev())
    torch.cuda.synchronize()
    names = ['fwd backbone', 'fwd neck', 'fwd seg', 'fwd det', 'fwd lane', 'loss', 'bwd loss+heads (data grads only)', 'bwd neck (data)', 'bwd backbone (params)']
    if verbose:
        for nm, a, b in zip(names, t[:-1], t[1:]):
            print(f"{nm:38s} {a.elapsed_time(b):8.2f} ms")
        print('total', t[0].elapsed_time(t[-1]))
for i in range(3): run(i == 2)
