"""do timing events recorded INSIDE a captured hipGraph work on this stack? (torch.cuda.Event(external=True) -> event record nodes)"""
import torch
x = torch.randn(4096, 4096, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for kw in (dict(enable_timing=True, external=True), dict(enable_timing=True)):
    try:
        ev = [torch.cuda.Event(**kw) for _ in range(4)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ev[0].record()
            y = x @ x
            ev[1].record()
            for _ in range(4):
                y = y @ x
            ev[2].record()
            z = y.sum()
            ev[3].record()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(kw, "elapsed:", [ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("whole replay:", e0.elapsed_time(e1))
    except Exception as e:
        print(kw, "FAILED:", repr(e)[:300])
