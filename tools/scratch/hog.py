"""occupies the GPU with matmuls for ~70 s (second process for tools/graph_share_probe.py: graph replays while the GPU is shared)"""
import torch, time
a = torch.randn(4096, 4096, device='cuda'); t0 = time.time()
while time.time() - t0 < 70:
    for _ in range(50): b = a @ a
    torch.cuda.synchronize()
