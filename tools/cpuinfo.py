import os, time, torch, yaml, sys
sys.path.insert(0, '.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; nproc")
import bench
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = 512, 1024
from oracle import hydranet_oracle as O
import multitask_hydranet_amd as pkg
net = pkg.HydraNet(cfgs)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
batch = O.synthetic_batch(cfgs, 1, 512, 1024, seed=1)
def step():
    for v in sd.values(): v.grad = None
    out = O.hydranet_forward(sd, cfgs, batch["image"], training=True)
    ld = O.hydranet_losses(cfgs, out, batch, lane_points_per_line=64)
    O.total_loss(cfgs, ld).backward()
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time(); step(); t1 = time.time(); step(); t2 = time.time()
    print("threads", nt, "first %.2f s second %.2f s" % (t1 - t0, t2 - t1), flush=True)
