"""Does a synchronous torch.distributed all_reduce issued on the capture stream keep the captured hipGraph linear (fast path)?"""
import os
import time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
x = torch.zeros(4096, device=dev)
flat = torch.zeros(20 << 20, device=dev)
dist.all_reduce(flat)
torch.cuda.synchronize()


def build(n, mode):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for i in range(n):
            x.add_(1.0)
            if i == n // 2 and mode == "inline":
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            if i == n // 2 and mode == "async":
                w = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
                w.wait()
            if i == n // 2 and mode == "side":
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        if mode == "side":
            cur.wait_stream(side)
    return g


def timeit(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for mode in ["none", "inline", "async", "side", "none"]:
    try:
        print(mode, f"{timeit(build(1200, mode)):.3f} ms", flush=True)
    except Exception as e:      # noqa: BLE001
        print(mode, "FAILED", repr(e)[:300], flush=True)
dist.destroy_process_group()
