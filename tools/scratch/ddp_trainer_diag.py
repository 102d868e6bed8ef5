"""where do HydraTrainer(capture_step=True, force_distribute=True) and the plain eager trainer diverge? per step: losses, gradients, parameters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.helpers import load_cfg, load_npz, tiny_state
from multitask_hydranet_amd.train import HydraTrainer

z = load_npz("tiny_hydranet.npz")
cfgs = load_cfg("hydranet_tiny.yml")
cfgs["train"].update(dict(continue_train=False, weight_file="", epoch=1, lr=1e-4, weight_decay=0.0))
batch = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("in/")}
g = torch.Generator().manual_seed(5)
loader = []
for i in range(5):
    b = dict(batch)
    b["image"] = batch["image"] + 0.05 * torch.randn(batch["image"].shape, generator=g)
    loader.append(b)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
mode = sys.argv[1] if len(sys.argv) > 1 else "cap"
runs = []
for ddp in (False, True):
    tr = HydraTrainer(cfgs, trainloader=loader, validloader=None, iters_per_epoch=len(loader), capture_step=(ddp and mode == "cap"), force_distribute=ddp)
    tr.hydranet.load_state_dict(tiny_state(z))
    tr.hydranet.lane_points_per_line = int(z["meta/lane_points_per_line"])
    rec = []
    for b in loader:
        ld = tr.train_step({k: v.clone() for k, v in b.items()})
        torch.cuda.synchronize()
        rec.append(({k: float(v) for k, v in ld.items()},
                    {n: p.grad.detach().clone() for n, p in tr.hydranet.named_parameters() if p.grad is not None},
                    {n: p.detach().clone() for n, p in tr.hydranet.named_parameters()}))
    runs.append(rec)
for step, ((l0, g0, p0), (l1, g1, p1)) in enumerate(zip(*runs)):
    bad_g = [(n, float((g0[n] - g1[n]).abs().max()), float(g0[n].abs().max())) for n in g0 if n in g1 and not torch.equal(g0[n], g1[n])]
    bad_p = [(n, float((p0[n] - p1[n]).abs().max())) for n in p0 if not torch.equal(p0[n], p1[n])]
    print("step", step, "loss equal", l0 == l1, "grads differing", len(bad_g), "of", len(g0), "params differing", len(bad_p), "of", len(p0),
          "grad key sets equal", set(g0) == set(g1))
    print("   ", bad_g[:4], bad_p[:4])
