"""Where does the dominant seg-decoder launch (decoder.3, phase form, N = 16) spend its time?  Ablation builds of the same launch through
hn_debug_knob(14): 1 = no epilogue, 2 = no MFMAs, 4 = no operand DMA after the first tiles, 8 = no LDS fragment reads; both loop forms
(hn_debug_direct_pipe 0 = 64-channel chunks, two weight buffers, vmcnt(0) per tap step; 1 = 32-channel chunks, ring of four, counted waits)."""
import sys, os, torch
os.environ["HN_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
from tools.bench_fused import timeit

dev = torch.device("cuda:0")
N, c0, k, h, w = 16, 256, 256, 32, 64
x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
wt = torch.randn(k, c0, 3, 3, device=dev) * 0.02
bias = torch.zeros(4 * k, device=dev)
T = K._phase_matrix(dev)
w_eff = (wt.reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
wpe, wte = K.pack_conv_weight(w_eff)
out = torch.empty(N, 2 * h, 2 * w, k, device=dev, dtype=torch.bfloat16)
z1 = torch.randn(N, 2 * h, 2 * w, k, device=dev).to(torch.bfloat16)
run = lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, N, h, w, c0, c0, wpe.data_ptr(), 4 * k, K.kp32(c0), bias.data_ptr(), 3,
                         out.data_ptr(), k, k, z1.data_ptr(), k)
# full-resolution 9-tap form of decoder.2 (512 -> 256 at 32 x 64) as the second shape
x2 = torch.randn(N, h, w, 512, device=dev).to(torch.bfloat16)
w2 = torch.randn(256, 512, 3, 3, device=dev) * 0.02
wp2, _ = K.pack_conv_weight(w2)
out2 = torch.empty(N, h, w, 256, device=dev, dtype=torch.bfloat16)
b2 = torch.zeros(256, device=dev)
run2 = lambda: K.k_gemm_nt(x2, None, 2, (N, h, w), wp2, 256, K.kp32(512), 9, bias=b2, act=3, out=out2, up=0)
for name, fn in (("decoder.3 phase x0 (4 taps x 256 ch, 1024 workgroups)", run), ("decoder.2 full (9 taps x 512 ch, 512 workgroups)", run2)):
    for pipe in (0, 1):
        lib().query("hn_debug_direct_pipe", pipe)
        line = f"{name} pipe{pipe}:"
        for dbg in (0, 1, 2, 4, 8, 3, 6, 10, 12, 14, 15):
            lib().query("hn_debug_knob", 14, dbg)
            line += f"  dbg{dbg}={timeit(fn, reps=5, iters=5):.0f}"
        lib().query("hn_debug_knob", 14, 0)
        print(line, flush=True)
lib().query("hn_debug_direct_pipe", 0)
