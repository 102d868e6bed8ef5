"""do SEVERAL LINEAR hipGraphs replayed on two streams, ordered by events between whole graphs, keep the back-to-back fast
path that a single graph loses as soon as it contains one fork/join (tools/graph_branch_probe.py: 1.6 us -> 2.9 us per node)?

Shape of the step this models: a main chain of N tiny dependent kernels cut into K segments (the dgrad chain of the backward pass, cut at
the stage boundaries), and after segment k a side chain of M kernels (the deferred weight gradients of that segment) that only the END of
the step waits for.  Variants:
  linear      one graph, main and side kernels in one chain                                  (what the step is today)
  forked      one graph, the side chains as branches (fork after segment k, one join at the end)
  multi       2K linear graphs: main segment k on stream A, side chain k on stream B behind an event recorded after main segment k;
              stream A waits for B's last event at the end of the step
Side kernels are either tiny (pure launch-rate question) or a 64 MB add (~30 us each: is the work really overlapped?).
Prints ms per step, median of 20, steps issued back to back (one device sync at the end of all of them) and one by one."""
import sys
import time

import torch


def tiny(t):
    t.add_(1.0)


def build(n, k, m, heavy, form):
    dev = torch.device("cuda:0")
    x = torch.zeros(4096, device=dev)
    ys = [torch.zeros((16 << 20) if heavy else 4096, device=dev) for _ in range(k)]
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    seg = n // k
    if form in ("linear", "forked"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=A):
            for s in range(k):
                for _ in range(seg):
                    tiny(x)
                if form == "linear":
                    for _ in range(m):
                        tiny(ys[s])
                else:
                    B.wait_stream(A)
                    with torch.cuda.stream(B):
                        for _ in range(m):
                            tiny(ys[s])
            if form == "forked":
                A.wait_stream(B)

        def step():
            with torch.cuda.stream(A):
                g.replay()
        return step, A
    mains, sides = [], []
    pool = None
    for s in range(k):
        gm = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gm, stream=A, pool=pool):
            for _ in range(seg):
                tiny(x)
        pool = gm.pool()
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, stream=B, pool=pool):
            for _ in range(m):
                tiny(ys[s])
        mains.append(gm)
        sides.append(gs)
    evA = [torch.cuda.Event() for _ in range(k)]
    evB = torch.cuda.Event()

    def step():
        for s in range(k):
            with torch.cuda.stream(A):
                mains[s].replay()
                evA[s].record(A)
            with torch.cuda.stream(B):
                B.wait_event(evA[s])
                sides[s].replay()
        evB.record(B)
        A.wait_event(evB)
    return step, A


def timeit(step, reps=20):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    one = []
    for _ in range(reps):
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one.append((time.perf_counter() - t) * 1e3)
    one.sort()
    t = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t) * 1e3 / reps
    return one[len(one) // 2], b2b


if __name__ == "__main__":
    n = 1000
    for heavy in (False, True):
        for k, m in ((1, 200), (6, 40), (6, 8)):
            for form in ("linear", "forked", "multi"):
                step, _ = build(n, k, m, heavy, form)
                one, b2b = timeit(step)
                print(f"main {n} tiny kernels in {k} segments, side {k} x {m} {'64 MB adds' if heavy else 'tiny kernels'}: {form:7s} "
                      f"{one:7.3f} ms per step (synced), {b2b:7.3f} ms back to back", flush=True)
