"""HIP (bf16) vs fp32 CPU oracle on the tiny cfg at a better-conditioned geometry; prints per-tensor errors."""
import sys, json, os
sys.path.insert(0, '.')
import numpy as np, torch, yaml
from multitask_hydranet_amd import HydraNet
from oracle import hydranet_oracle as O
N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = yaml.safe_load(open('cfgs/hydranet_tiny.yml'))
z = np.load('tests/golden/tiny_hydranet.npz')
sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith('sd/')}
ppl = H // 8
# lane head output widths depend on H: re-init those convs
torch.manual_seed(3)
net = HydraNet({**cfgs, 'dataloader': {**cfgs['dataloader'], 'network_input_height': H, 'network_input_width': W}})
own = net.state_dict()
for k, v in sd.items():
    if own[k].shape == v.shape:
        own[k].copy_(v)
sd = {k: v.clone() for k, v in net.state_dict().items()}
batch = O.synthetic_batch(cfgs, N, H, W, seed=11)
dev = torch.device('cuda:0')
net = net.to(dev).train()
net.lane_points_per_line = ppl
gb = {k: v.to(dev) for k, v in batch.items()}
feats = net._backbone(gb['image']); fused = net._neck(feats)
net.load_state_dict(sd)
out = net(gb['image']); ld = net.cal_loss(out, gb); tot = net.total_loss(ld); tot.backward(); torch.cuda.synchronize()
osd = {k: v.clone() for k, v in sd.items()}
for k, v in osd.items():
    if v.is_floating_point() and 'running' not in k: v.requires_grad_(True)
import contextlib
ctx = O.bf16_mirror() if (len(sys.argv) > 4 and sys.argv[4] == 'mirror') else contextlib.nullcontext()
with ctx:
    oo = O.hydranet_forward(osd, cfgs, batch['image'], training=True, want_features=True)
old = O.hydranet_losses(cfgs, oo, batch, lane_points_per_line=ppl); otot = O.total_loss(cfgs, old); otot.backward()
def se(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return round(float((a - b).abs().max() / b.abs().max().clamp(min=1e-20)), 4)
rep = {}
for i, f in enumerate(feats): rep[f'feat{i}'] = se(f.permute(0, 3, 1, 2), oo['_feats'][i])
for i, f in enumerate(fused): rep[f'fused{i}'] = se(f.permute(0, 3, 1, 2), oo['_fused'][i])
rep['seg'] = se(out['seg'], oo['seg']); rep['reg'] = se(out['detection']['regression'], oo['detection']['regression'])
rep['cls'] = se(out['detection']['classification'], oo['detection']['classification'])
rep['lane_cls'] = se(out['lane']['predict_cls'], oo['lane']['predict_cls']); rep['lane_loc'] = se(out['lane']['predict_loc'], oo['lane']['predict_loc'])
print('ACT', rep)
print('LOSS', {k: (round(float(v), 5), round(float(old[k]), 5)) for k, v in ld.items()}, float(tot), float(otot))
g = {}
for name, p in net.named_parameters():
    r = osd[name].grad
    if r is None or p.grad is None: continue
    a = p.grad.float().cpu()
    if float(r.abs().max()) < 1e-6: continue
    g[name] = (round(float(torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0)), 4), se(a, r))
bad = sorted(g.items(), key=lambda kv: kv[1][0])
print('GRAD worst', bad[:25])
print('GRAD n', len(g), 'n cos<0.98', sum(1 for v in g.values() if v[0] < 0.98), 'median cos', float(np.median([v[0] for v in g.values()])))
