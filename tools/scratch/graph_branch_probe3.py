"""Does a small pinned H2D memcpy node inside a linear captured hipGraph slow the replay?"""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(4096, device=dev)
h = torch.zeros(4096, dtype=torch.int64).pin_memory()
d = torch.zeros(4096, dtype=torch.int64, device=dev)
d2 = torch.zeros(4096, dtype=torch.int64, device=dev)


def build(n, copies, kind):
    g = torch.cuda.CUDAGraph()
    where = set(int((i + 1) * n / (copies + 1)) for i in range(copies))
    with torch.cuda.graph(g):
        for i in range(n):
            x.add_(1.0)
            if i in where:
                if kind == "h2d":
                    d.copy_(h, non_blocking=True)
                elif kind == "d2d":
                    d2.copy_(d)
                elif kind == "memset":
                    d2.zero_()
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


for copies, kind in [(0, "none"), (1, "h2d"), (4, "h2d"), (4, "d2d"), (4, "memset"), (0, "none")]:
    print(copies, kind, f"{timeit(build(1200, copies, kind)):.3f} ms", flush=True)
