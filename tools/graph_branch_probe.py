"""How much does a fork/join cost inside a captured hipGraph on this runtime?  A chain of N tiny dependent kernels, captured
(a) linear, (b) with F fork/join pairs (each side branch = B tiny kernels on a second stream), replayed and timed."""
import sys
import time
import torch


def build(n, forks, branch_len, side_big=False):
    dev = torch.device("cuda:0")
    x = torch.zeros(4096, device=dev)
    y = torch.zeros(4096, device=dev)
    big = torch.zeros(64 << 20, device=dev) if side_big else None
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    where = set(int((i + 1) * n / (forks + 1)) for i in range(forks))
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        pending = False
        for i in range(n):
            x.add_(1.0)
            if i in where:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(branch_len):
                        y.add_(1.0)
                    if big is not None:
                        big.add_(1.0)
                pending = True
        if pending:
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


if __name__ == "__main__":
    n = 1200
    for forks, bl in [(0, 0), (1, 1), (1, 50), (2, 1), (8, 1), (8, 50), (32, 4), (0, 0)]:
        g = build(n, forks, bl)
        print(f"chain {n}, forks {forks}, branch kernels {bl}: {timeit(g):.3f} ms", flush=True)
    g = build(n, 1, 1, side_big=True)
    print(f"chain {n}, 1 fork with a 256 MB add on the side: {timeit(g):.3f} ms")
