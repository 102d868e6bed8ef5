"""HydraTrainer end to end (forward + loss + backward + Adam + LR step) on the bench workload, eager launches vs capture_step=True"""
import os, sys, time, yaml, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
from bench import synthetic_batch
from multitask_hydranet_amd.train import HydraTrainer

h, w, n = 512, 1024, 16
cfgs = yaml.safe_load(open(os.path.join(ROOT, "cfgs", "hydranet_big.yml")))
cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
cfgs["train"].update(dict(continue_train=False, weight_file="", epoch=1, lr=1e-4, weight_decay=0.0))
batch = synthetic_batch(cfgs, n, h, w, seed=1, device=torch.device("cuda:0"))
for capture in (False, True):
    torch.manual_seed(0)
    tr = HydraTrainer(cfgs, trainloader=None, validloader=None, iters_per_epoch=1000, capture_step=capture)
    tr.hydranet.check_finite = False
    tr.hydranet.lane_points_per_line = h // cfgs["lane"]["interval"]
    for _ in range(5):
        tr.train_step(dict(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 20
    for _ in range(k):
        ld = tr.train_step(dict(batch))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print("capture_step=%s: %.2f ms per iteration (incl. Adam + LR step) = %.1f img/s, loss %.4f" % (capture, dt * 1e3, n / dt, float(ld["total_loss"])), flush=True)
    del tr
    torch.cuda.empty_cache()
