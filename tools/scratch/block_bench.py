"""True GPU time per network section under hipGraph replay (no profiler, no Python launch overhead)."""
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
bf = torch.bfloat16
def act(c, s): return torch.randn(n, h // s, w // s, c, device=dev).to(bf).requires_grad_(True)

def timed(name, fn, inputs, iters=20):
    def step():
        net.zero_grad(set_to_none=True)
        for t in inputs:
            if t.grad is not None: t.grad = None
        outs = fn(*inputs)
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        loss = sum(o.float().mean() for o in outs if o.is_floating_point())
        loss.backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): step()
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / iters:8.3f} ms", flush=True)

p = "backbone.net."
img = batch["image"]
timed("stem (fwd+bwd)", lambda x: net._cba(x, p + "stem.conv", p + "stem.bn", dict(eps=1e-5, momentum=0.1), kind="stem", act=1), [img])
chans = [32] + net.widths
for k, d in enumerate(net.depths):
    s_in = 2 * 2 ** k
    x = act(chans[k], s_in)
    timed(f"stage_{k} block_0 (stride 2, {chans[k]}->{chans[k+1]})", lambda t, k=k: net._xblock(f"{p}stage_{k}.blocks.block_0.", t, 2), [x])
    if d > 1:
        x = act(chans[k + 1], s_in * 2)
        timed(f"stage_{k} block_1 (x{d-1} such blocks, C={chans[k+1]})", lambda t, k=k: net._xblock(f"{p}stage_{k}.blocks.block_1.", t, 1), [x])
feats = [act(c, 4 * 2 ** i) for i, c in enumerate(net.widths)]
timed("neck (3 BiFPN cells)", lambda *f: net._neck(list(f)), feats)
fused = [act(112, 8 * 2 ** i) for i in range(5)]
timed("seg head", lambda a, b, c, d: net._seg([a, b, c, d]), [feats[0], fused[0], fused[1], fused[2]])
timed("det head", lambda *f: net._det(img, list(f))[1:], fused)
timed("lane head", lambda *f: list(net._lane(list(f)).values()), fused)
def full():
    out = net(img); ld = net.cal_loss(out, batch); return net.total_loss(ld)
timed("full step", lambda: full(), [])
with torch.no_grad():
    out = net(img)
seg = out["seg"].detach().requires_grad_(True)
cls = out["detection"]["classification"].detach().requires_grad_(True); reg = out["detection"]["regression"].detach().requires_grad_(True)
lc = out["lane"]["predict_cls"].detach().requires_grad_(True); ll = out["lane"]["predict_loc"].detach().requires_grad_(True)
def lossonly(seg, cls, reg, lc, ll):
    o = {"seg": seg, "detection": {"anchors": out["detection"]["anchors"], "regression": reg, "classification": cls}, "lane": {"predict_cls": lc, "predict_loc": ll}}
    return net.total_loss(net.cal_loss(o, batch))
timed("losses only (fwd+bwd)", lossonly, [seg, cls, reg, lc, ll])
