"""Which Python lines of the package still launch ATen / library kernels inside one training step?  Runs two eager steps of the bench
workload under torch.profiler (with_stack) and prints top-level aten ops that own device kernels, grouped by the innermost frame inside
multitask_hydranet_amd/ (forward ops and the Python bodies of the custom backward nodes; gradient accumulation by the autograd engine
shows up as 'autograd engine')."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
from bench import synthetic_batch  # noqa: E402
from multitask_hydranet_amd import HydraNet  # noqa: E402

dev = torch.device("cuda:0")
h, w, n = 512, 1024, int(os.environ.get("N", "16"))
cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
torch.manual_seed(0)
net = HydraNet(cfgs).to(dev).train()
net.check_finite = False
net.lane_points_per_line = h // cfgs["lane"]["interval"]
batch = synthetic_batch(cfgs, n, h, w, seed=1, device=dev)


def step():
    for p in net.parameters():
        p.grad = None
    out = net(batch["image"])
    loss = net.total_loss(net.cal_loss(out, batch))
    loss.backward()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_OPS = ("view", "reshape", "slice", "select", "permute", "transpose", "expand", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "t.",
            "empty", "_unsafe_view", "split", "unbind", "narrow", "sym_", "stride", "size", "is_", "_local_scalar", "lift_fresh", "chunk")
groups = collections.Counter()
shapes = collections.defaultdict(collections.Counter)


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_OPS):
            where = "autograd engine (accumulation) / outside the package"
            for fr in reversed(traceback.extract_stack()):
                if "multitask_hydranet_amd/" in fr.filename:
                    where = "%s:%d %s" % (fr.filename.split("multitask_hydranet_amd/")[-1], fr.lineno, fr.name)
                    break
            groups[(name, where)] += 1
            sh = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            shapes[(name, where)][str(sh[:2])] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
with Log():
    step()
torch.cuda.synchronize()
tot = 0
for (name, where), c in sorted(groups.items(), key=lambda kv: -kv[1]):
    top = ", ".join("%s x%d" % kv for kv in shapes[(name, where)].most_common(3))
    print("%-28s x%-4d %s   | %s" % (name, c, where, top))
    tot += c
print("total aten ops with kernels (approx): %d" % tot)
