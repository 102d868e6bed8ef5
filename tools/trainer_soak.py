"""Soak run of the captured training step (round 6): `STEPS` iterations (default 600) of HydraTrainer(capture_step=True) on the bench workload with
fresh synthetic batches of four seeds in rotation; prints the loss every 100 iterations, checks that it stays finite and that the persistent
stage launches never raised their status word (a bounded wait that expired would show up here, not as a hang)."""
import os, sys, time, yaml, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from bench import synthetic_batch
from multitask_hydranet_amd.train import HydraTrainer
import multitask_hydranet_amd.ops.xstage as XS

h, w, n = 512, 1024, 16
steps = int(os.environ.get("STEPS", "600"))
cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
cfgs["train"].update(dict(continue_train=False, weight_file="", epoch=1, lr=1e-4, weight_decay=0.0))
dev = torch.device("cuda:0")
batches = [synthetic_batch(cfgs, n, h, w, seed=s, device=dev) for s in (1, 2, 3, 4)]
torch.manual_seed(0)
tr = HydraTrainer(cfgs, trainloader=None, validloader=None, iters_per_epoch=steps + 10, capture_step=True)
tr.hydranet.lane_points_per_line = h // cfgs["lane"]["interval"]
t0 = time.perf_counter()
for i in range(steps):
    ld = tr.train_step(dict(batches[i & 3]))
    if i % 100 == 99 or i == steps - 1:
        torch.cuda.synchronize()
        loss = float(ld["total_loss"])
        st = XS.xstage_status(dev)
        print(f"iteration {i + 1}: loss {loss:.4f}  persistent-launch status 0x{st:x}  {(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms per iteration", flush=True)
        assert loss == loss and abs(loss) != float("inf"), "loss is not finite"
        assert st == 0, "a persistent stage launch raised its status word"
print("soak ok")
