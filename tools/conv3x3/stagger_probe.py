"""Experiment (round 6): are the two co-resident workgroups of conv3x3_direct_kernel<128> phase-locked?  Every launch starts 512 workgroups at
once (2 per CU), so both workgroups of a CU run prologue, tap loop and epilogue at the same moments and never cover each other.
hn_debug_knob(19, us) makes workgroups 256..511 (the second one of each CU in the first round) wait `us` microseconds before they start;
later rounds inherit the offset.  Tuning build.  Prints the launch time per layer and stagger."""
import sys, os, torch
os.environ["HN_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib
from tools.bench_fused import timeit

dev = torch.device("cuda:0")
N = 16
LAYERS = [("d1", 512, 112, 512, 16, 32, 1), ("d2", 512, 0, 256, 32, 64, 0), ("d3", 256, 112, 256, 32, 64, 1), ("d4", 256, 0, 128, 64, 128, 0),
          ("d5", 128, 24, 128, 64, 128, 1)]
STAG = [0, 4, 8, 12, 16, 24, 32]
for name, c0, c1, k, h, w, up in LAYERS:
    H, W = (2 * h, 2 * w) if up else (h, w)
    x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
    wt = torch.randn(k, c0 + c1, 3, 3, device=dev) * 0.02
    bias = torch.zeros(k, device=dev)
    wp, wtt = K.pack_conv_weight(wt)
    dz = torch.randn(N, H, W, k, device=dev).to(torch.bfloat16)
    runs = []
    if up:
        T = K._phase_matrix(dev)
        w_eff = (wt[:, :c0].reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
        wpe, wte = K.pack_conv_weight(w_eff)
        be = bias.repeat(4)
        out = torch.empty(N, H, W, k, device=dev, dtype=torch.bfloat16)
        z1 = torch.randn(N, H, W, k, device=dev).to(torch.bfloat16)
        runs.append(("phase fwd x0", lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, N, h, w, c0, c0, wpe.data_ptr(), 4 * k, K.kp32(c0), be.data_ptr(), 3,
                                                        out.data_ptr(), k, k, z1.data_ptr(), k)))
        dzs = torch.randn(N, h, w, 4 * k, device=dev).to(torch.bfloat16)
        dvpl = torch.empty(N, h + 2, w + 2, c0, device=dev, dtype=torch.bfloat16)
        runs.append(("phase dgrad x0", lambda: lib().call("hn_conv3x3_phase", dzs.data_ptr(), 3, N, h + 2, w + 2, 4 * k, 4 * k, wte.data_ptr(), c0, K.kp32(4 * k), None, 0,
                                                          dvpl.data_ptr(), c0, k, None, 0)))
    else:
        out = torch.empty(N, H, W, k, device=dev, dtype=torch.bfloat16)
        runs.append(("full fwd", lambda: K.k_gemm_nt(x0, None, 2, (N, H, W), wp, k, K.kp32(c0), 9, bias=bias, act=3, out=out, up=0)))
        dvp = torch.empty(N, H + 2, W + 2, c0, device=dev, dtype=torch.bfloat16)
        runs.append(("full dgrad", lambda: K.k_gemm_nt(dz, None, 3, (N, H + 2, W + 2), wtt, c0, K.kp32(k), 9, c0=k, c1=0, out=dvp)))
    for rn, fn in runs:
        line = f"{name} {rn:15s}:"
        for s in STAG:
            lib().query("hn_debug_knob", 19, s)
            line += f"  {s}us={timeit(fn, reps=5, iters=5):.1f}"
        lib().query("hn_debug_knob", 19, 0)
        print(line, flush=True)
