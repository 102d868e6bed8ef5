"""CPU, world_size 2, gloo: the bucketed gradient exchange averages gradients exactly like a single process over the concatenated batch,
skips named never-used parameters identically on both ranks, and re-arms for the next step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    m.unused = torch.nn.Parameter(torch.ones(3))
    return m


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multitask_hydranet_amd.ddp import GradReducer, broadcast_state
    m = _net()
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)                              # diverge on purpose; broadcast_state must undo it
    broadcast_state(m)
    red = GradReducer(list(m.named_parameters()), bucket_bytes=600, skip=("unused",))
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 8, generator=g)
    y = torch.randn(2, 6, 4, generator=g)
    out = {}
    for step in range(2):
        m.zero_grad(set_to_none=False) if step else None
        loss = ((m(x[rank]) - y[rank]) ** 2).mean()
        loss.backward()
        red.finish()
        out[step] = [p.grad.clone().numpy() for n, p in m.named_parameters() if n != "unused"]   # numpy: pickled by value (no fd passing race)
    assert m.unused.grad is None
    q.put((rank, out))
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = _net()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 8, generator=g)
    y = torch.randn(2, 6, 4, generator=g)
    loss = 0.5 * (((m(x[0]) - y[0]) ** 2).mean() + ((m(x[1]) - y[1]) ** 2).mean())
    loss.backward()
    ref = [p.grad for n, p in m.named_parameters() if n != "unused"]
    for step in (0, 1):
        for a, b, r in zip(res[0][step], res[1][step], ref):
            a, b = torch.from_numpy(a), torch.from_numpy(b)
            assert torch.equal(a, b)                                    # both ranks hold identical averaged gradients
            assert torch.allclose(a, r, rtol=1e-5, atol=1e-7)


def _worker_static(rank, world, port, q):
    """graph-replay mode: the 'captured step' rewrites the same gradient tensors; reduce_now() must pick the fresh values up every time"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multitask_hydranet_amd.ddp import GradReducer
    m = _net()
    static = {n: torch.zeros_like(p) for n, p in m.named_parameters() if n != "unused"}
    for n, p in m.named_parameters():
        if n != "unused":
            p.grad = static[n]
    red = GradReducer(list(m.named_parameters()), bucket_bytes=600, skip=("unused",))
    red.remove()
    red.bind_static_grads()
    out = {}
    for step in range(3):
        for i, (n, t) in enumerate(static.items()):          # "replay": new contents in the same tensors
            t.fill_(float((rank + 1) * (step + 1) + i))
        red.reduce_now()
        out[step] = [p.grad.clone().numpy() for n, p in m.named_parameters() if n != "unused"]
    q.put((rank, out))
    dist.destroy_process_group()


def test_reduce_now_tracks_static_graph_grads():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_static, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(3):
        for i, (a, b) in enumerate(zip(res[0][step], res[1][step])):
            expect = 0.5 * ((1 * (step + 1) + i) + (2 * (step + 1) + i))
            assert (a == b).all() and abs(float(a.reshape(-1)[0]) - expect) < 1e-6 and (a == a.reshape(-1)[0]).all()


def _worker_real(rank, world, port, q):
    """GradReducer over HydraNet's real 697-tensor parameter list (names / shapes recorded from the reference: tests/golden/big_keys.npz)"""
    import hashlib
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multitask_hydranet_amd.ddp import GradReducer, UNUSED_5STAGE
    from tests.helpers import load_npz
    z = load_npz("big_keys.npz")
    shapes = dict(zip(z["keys"].tolist(), z["shapes"].tolist()))
    named = []
    for i, k in enumerate(z["param_keys"].tolist()):
        shp = tuple(int(v) for v in shapes[k].split(",")) if shapes[k] else ()
        named.append((k, torch.nn.Parameter(torch.zeros(shp))))
    red = GradReducer(list(named), skip=UNUSED_5STAGE)           # DDP's 25 MiB default
    comp = [[n for n, _ in b["params"]] for b in red.buckets]
    sizes = [int(sum(v.numel() for v in b["views"])) for b in red.buckets]          # payload elements (slots start on 16-byte boundaries)
    assert all(o % 4 == 0 for b in red.buckets for o in b["offs"]) and all(b["flat"].numel() % 4 == 0 for b in red.buckets)
    # after-replay mode on the real composition: static gradients, one exchange
    for j, (n, p) in enumerate(named):
        if n not in UNUSED_5STAGE:
            p.grad = torch.full_like(p, float(rank + 1) * (1 + (j % 7)))
    red.remove()
    red.bind_static_grads()
    red.reduce_now()
    ok = True
    for j, (n, p) in enumerate(named):
        if n in UNUSED_5STAGE:
            ok &= p.grad is None
        else:
            ok &= bool((p.grad == 1.5 * (1 + (j % 7))).all())
    digest = hashlib.sha256("|".join(",".join(c) for c in comp).encode()).hexdigest()
    q.put((rank, dict(n_buckets=len(comp), sizes=sizes, digest=digest, first=comp[0][:3], last=comp[-1][-3:], ok=bool(ok),
                      total=sum(len(c) for c in comp))))
    dist.destroy_process_group()


def test_real_parameter_list_buckets_identically_on_both_ranks():
    """bucket composition / ordering for HydraNet's real parameter list (693 exchanged tensors of 697; 170.9 MB fp32 -> 7 buckets of
    ~25 MiB in reverse execution order: lane head first, stem last) is identical on both ranks, the four never-used p5_to_p6 tensors are
    excluded by name, and one exchange over that composition averages every gradient"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["digest"] == b["digest"] and a["sizes"] == b["sizes"] and a["n_buckets"] == b["n_buckets"]
    assert a["ok"] and b["ok"]
    assert a["total"] == 693 and sum(a["sizes"]) == 42715747 - (376 * 112 + 112 + 112 + 112)
    assert 6 <= a["n_buckets"] <= 8
    assert a["first"][0].startswith("laneheader.") and a["last"][-1] == "backbone.net.stem.conv.weight"
