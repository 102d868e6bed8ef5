"""hipGraph replays of the full step while another process shares the GPU: where does the loss turn non-finite?"""
import sys, time, os, torch, yaml
sys.path.insert(0, '.')
import bench
from multitask_hydranet_amd import HydraNet
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
h, w, n = 512, 1024, int(os.environ.get("PROBE_BATCH", "16"))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = HydraNet(cfgs).to(dev).train(); net.check_finite = False; net.lane_points_per_line = h // 8
batch = bench.synthetic_batch(cfgs, n, h, w, 1, dev)
def fwd_bwd():
    out = net(batch["image"]); ld = net.cal_loss(out, batch); loss = net.total_loss(ld); loss.backward(); return loss
if os.environ.get("PROBE_NULLSTREAM"):
    for _ in range(2):
        net.zero_grad(set_to_none=True); fwd_bwd()
    keep = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        net.zero_grad(set_to_none=True); fwd_bwd()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
if os.environ.get("PROBE_RELOAD"):
    state = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(state)
net.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    static_loss = fwd_bwd()
t0 = time.time()
def finite_report(tag):
    bad = []
    for k, v in net.state_dict().items():
        if v.is_floating_point() and not torch.isfinite(v).all(): bad.append(k)
    print(tag, "non-finite state entries:", bad[:5], len(bad), flush=True)
grads = [p.grad for p in net.parameters() if p.grad is not None]
def checksum():
    return float(torch.stack(torch._foreach_norm(grads)).double().sum())
for i in range(10):
    if mode == "sleep" and i == 3: time.sleep(2.0)
    if i % 2 == 1: torch.cuda.synchronize()           # the trigger of the memset-node failure: a host sync between replays
    g.replay()
    v = float(static_loss.detach())
    print(f"replay {i} t={time.time() - t0:.2f}s loss {v} grad-norm checksum {checksum():.10e}", flush=True)
    if v != v:
        finite_report("after NaN:")
        break
