"""One rank of tests/test_train_gpu.py::test_two_rank_gradients_are_the_mean_of_the_single_rank_gradients (started twice by the test, both
ranks on cuda:0, gloo -- RCCL refuses two ranks on one device).  DDP semantics of model/train.py:130-137 with the REAL HIP model (tiny cfg),
a different batch per rank:

  A. single-rank gradients g_r (no reducer); their mean over the two ranks (a plain gloo all-reduce of copies) is the expected result;
  B. GradReducer in eager hook mode, two steps: step 0 starts from .grad = None (the deferred weight-gradient kernels write straight into
     their bucket slots: `_hn_grad_slot` / ops.grad_out), step 1 accumulates into the bucket views -- after finish() every parameter's
     .grad must equal the mean (1e-6 of the tensor's magnitude: the exchange adds two fp32 numbers and halves the sum);
  C. HydraTrainer(capture_step=True) at world size 2: two eager iterations, then captured ones (gloo collectives cannot be captured: the
     graph holds forward + loss + backward, the buckets are exchanged after every replay) -- parameters after five iterations must equal,
     bit for bit, those of the eager-hook trainer, and be identical on both ranks.

Exit code 0 = all assertions held on this rank.  Not a test module (no test_ prefix): it needs RANK / WORLD_SIZE / MASTER_* in the env."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.helpers import load_cfg, load_npz, tiny_state  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multitask_hydranet_amd import HydraNet
    from multitask_hydranet_amd.ddp import GradReducer, unused_parameters
    from multitask_hydranet_amd.train import HydraTrainer

    z = load_npz("tiny_hydranet.npz")
    cfgs = load_cfg("hydranet_tiny.yml")
    cfgs["train"].update(dict(continue_train=False, weight_file="", epoch=1, lr=1e-4, weight_decay=0.0))
    batch = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("in/")}
    g = torch.Generator().manual_seed(100 + rank)
    batch["image"] = batch["image"] + 0.1 * torch.randn(batch["image"].shape, generator=g)        # a different batch per rank
    gb = {k: v.cuda() for k, v in batch.items()}
    ppl = int(z["meta/lane_points_per_line"])

    def make():
        net = HydraNet(cfgs).cuda().train()
        net.load_state_dict(tiny_state(z))
        net.lane_points_per_line = ppl
        return net

    def fwd_bwd(net):
        net.total_loss(net.cal_loss(net(gb["image"]), gb)).backward()

    # ---- A: single-rank gradients and their mean over the ranks ------------------------------------------------------------------
    net = make()
    fwd_bwd(net)
    torch.cuda.synchronize()
    local = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    mean = {}
    for n, t in local.items():
        m = t.clone()
        dist.all_reduce(m)
        mean[n] = m / world
    other = {n: 2 * mean[n] - local[n] for n in local}
    assert any(float((other[n] - local[n]).abs().max()) > 1e-3 * float(local[n].abs().max() + 1e-30) for n in local), "the ranks must differ"

    def check(net_, what):
        got = {n: p.grad for n, p in net_.named_parameters() if p.grad is not None}
        assert got.keys() == mean.keys(), (what, set(got) ^ set(mean))
        for n in mean:
            err = float((got[n].double() - mean[n].double()).abs().max())
            bound = 1e-6 * float(mean[n].abs().max()) + 1e-12
            assert err <= bound, (what, n, err, bound)

    # ---- B: eager hook mode, in-place bucket slots (step 0) and accumulation into the bucket views (step 1) ----------------------
    net = make()
    red = GradReducer(list(net.named_parameters()), world_size=world, skip=unused_parameters(net), bucket_bytes=64 << 10)
    assert red.active and len(red.buckets) >= 4 and all(hasattr(p, "_hn_grad_slot") for b in red.buckets for _, p in b["params"])
    for step in range(2):
        net.zero_grad(set_to_none=(step == 0))
        fwd_bwd(net)
        red.finish()
        torch.cuda.synchronize()
        if step == 0:
            assert red.direct_fraction() > 0.0, "no gradient was produced in its bucket slot"
        check(net, "eager hook mode, step %d" % step)
    red.remove()

    # ---- C: the trainer's captured step (exchange after every replay under gloo) == its eager hook form ---------------------------
    os.environ["LOCAL_RANK"] = "0"
    loader = []
    for i in range(5):
        b = dict(batch)
        b["image"] = batch["image"] + 0.05 * torch.randn(batch["image"].shape, generator=g)
        loader.append(b)
    finals = []
    for capture in (False, True):
        tr = HydraTrainer(copy.deepcopy(cfgs), trainloader=loader, validloader=None, iters_per_epoch=len(loader), capture_step=capture)
        assert tr.use_distribute and tr.reducer is not None and tr.reducer.active
        tr.hydranet.load_state_dict(tiny_state(z))
        tr.hydranet.lane_points_per_line = ppl
        losses = []
        for b in loader:
            ld = tr.train_step({k: v.clone() for k, v in b.items()})
            losses.append({k: float(v.detach()) for k, v in ld.items()})
        torch.cuda.synchronize()
        assert (tr._cap is not None) == capture
        finals.append((losses, {n: p.detach().clone() for n, p in tr.hydranet.named_parameters()}))
        tr.reducer.remove()
    (l0, p0), (l1, p1) = finals
    assert l0 == l1, (l0, l1)
    for n in p0:
        assert torch.equal(p0[n], p1[n]), ("captured != eager", n)
    # both ranks hold the same parameters after five data-parallel iterations (and they moved)
    for n, t in p1.items():
        both = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        assert torch.equal(both[0], both[1]), ("ranks diverged", n)
    init = tiny_state(z)
    assert sum(int(not torch.equal(p1[n].cpu(), init[n])) for n in p1) > 0.8 * len(p1)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
