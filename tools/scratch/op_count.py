import sys, os, torch, yaml
sys.path.insert(0, '.')
import bench
from multitask_hydranet_amd import HydraNet
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
h, w = 512, 1024
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
dev = torch.device('cuda:0')
net = HydraNet(cfgs).to(dev).train(); net.check_finite = False; net.lane_points_per_line = h // 8
batch = bench.synthetic_batch(cfgs, 4, h, w, 1, dev)
def step():
    net.zero_grad(set_to_none=True)
    out = net(batch["image"]); ld = net.cal_loss(out, batch); net.total_loss(ld).backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    step()
torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.count:6d}  {e.key[:70]}")
