"""Run ONE network section eagerly (fwd+bwd) a few times so that `rocprofv3 --kernel-trace` yields its per-dispatch timeline.
usage: rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 tools/trace_section.py s4b1
       python tools/trace_section.py --parse gpurun_out/tr 6      (prints the last of 6 iterations)"""
import sys, glob, csv, re
if sys.argv[1] == "--parse":
    iters = int(sys.argv[3])
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // iters
    rows = rows[-per:]
    t0 = int(rows[0]["Start_Timestamp"]); tot = 0
    prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = re.sub(r"\(.*", "", r["Kernel_Name"]); nm = nm.replace("void ", "")[:60]
        g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        print(f"{(s - t0) / 1e3:9.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f} us  wg {g:6d}  {nm}")
        tot += e - s; prev_end = e
    print(f"launches {per}  kernel time {tot / 1e3:.1f} us  span {(prev_end - t0) / 1e3:.1f} us")
    sys.exit(0)
import torch, yaml
sys.path.insert(0, '.')
import bench
from multitask_hydranet_amd import HydraNet
cfgs = yaml.safe_load(open('cfgs/hydranet_big.yml'))
h, w, n = 512, 1024, 16
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
dev = torch.device('cuda:0')
net = HydraNet(cfgs).to(dev).train(); net.check_finite = False; net.lane_points_per_line = h // 8
batch = bench.synthetic_batch(cfgs, n, h, w, 1, dev)
def act(c, s): return torch.randn(n, h // s, w // s, c, device=dev).to(torch.bfloat16).requires_grad_(True)
p = "backbone.net."
chans = [32] + net.widths
which = sys.argv[1]
img = batch["image"]
if which[0] == "s" and which[1].isdigit():
    k, b = int(which[1]), int(which[3])
    x = act(chans[k + (1 if b else 0)], (2 if b else 1) * 2 * 2 ** k)
    fn, inputs = (lambda t: net._xblock(f"{p}stage_{k}.blocks.block_{b}.", t, 1 if b else 2)), [x]
else:
    feats = [act(c, 4 * 2 ** i) for i, c in enumerate(net.widths)]
    fused = [act(112, 8 * 2 ** i) for i in range(5)]
    if which == "neck": fn, inputs = (lambda *f: net._neck(list(f))), feats
    elif which == "det": fn, inputs = (lambda *f: net._det(img, list(f))[1:]), fused
    elif which == "seg": fn, inputs = (lambda a, b, c, d: net._seg([a, b, c, d])), [feats[0], fused[0], fused[1], fused[2]]
    elif which == "lane": fn, inputs = (lambda *f: list(net._lane(list(f)).values())), fused
for it in range(6):
    net.zero_grad(set_to_none=True)
    for t in inputs: t.grad = None
    outs = fn(*inputs); outs = outs if isinstance(outs, (list, tuple)) else [outs]
    sum(o.float().mean() for o in outs if o.is_floating_point()).backward()
    torch.cuda.synchronize()
