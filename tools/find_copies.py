"""Where do aten::copy_ / clone / fill / add calls of one fwd+bwd come from? (torch.profiler with stacks, eager step)"""
import sys, collections, torch, yaml
sys.path.insert(0, '.')
import bench
from multitask_hydranet_amd import HydraNet
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
h, w, n = 512, 1024, 16
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
dev = torch.device('cuda:0')
net = HydraNet(cfgs).to(dev).train(); net.check_finite = False; net.lane_points_per_line = h // 8
batch = bench.synthetic_batch(cfgs, n, h, w, 1, dev)
def step():
    net.zero_grad(set_to_none=True)
    out = net(batch["image"]); ld = net.cal_loss(out, batch); net.total_loss(ld).backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::contiguous", "aten::zeros", "aten::zeros_like"):
        st = [s for s in (e.stack or []) if "multitask_hydranet_amd" in s or "autograd" in s or "bench" in s][:2]
        cnt[(e.name, " <- ".join(s.split("/")[-1][:70] for s in st))] += 1
for (k, v) in cnt.most_common(60):
    print(v, k)
