"""s_memtime stamps of the dominant seg-decoder launch (decoder.3 phase form): per-iteration cycle counts of every 64th workgroup's wave 0."""
import sys, os, torch
os.environ["HN_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
from multitask_hydranet_amd._lib import lib

dev = torch.device("cuda:0")
N, c0, k, h, w = 16, 256, 256, 32, 64
if os.environ.get("LAYER") == "d7":            # decoder.7: 64 -> 64 over the up-sampled 128 x 256 map, no skip operand (single chunk: wpre path)
    c0, k, h, w = 64, 64, 128, 256
x0 = torch.randn(N, h, w, c0, device=dev).to(torch.bfloat16)
wt = torch.randn(k, c0, 3, 3, device=dev) * 0.02
bias = torch.zeros(4 * k, device=dev)
T = K._phase_matrix(dev)
w_eff = (wt.reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
wpe, wte = K.pack_conv_weight(w_eff)
out = torch.empty(N, 2 * h, 2 * w, k, device=dev, dtype=torch.bfloat16)
z1 = torch.randn(N, 2 * h, 2 * w, k, device=dev).to(torch.bfloat16)
ADD = os.environ.get("NO_ADDEND") != "1" and os.environ.get("LAYER") != "d7"
run = lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, N, h, w, c0, c0, wpe.data_ptr(), 4 * k, K.kp32(c0), bias.data_ptr(), 3,
                         out.data_ptr(), k, k, z1.data_ptr() if ADD else None, k)
if os.environ.get("LAYER") in ("g3", "g4"):       # grouped 3x3 conv (group width 8) of a stage-3 / stage-4 XBlock, forward with BatchNorm statistics
    c, hh, ww = (376, 16, 32) if os.environ["LAYER"] == "g3" else (936, 8, 16)
    xa = torch.randn(N, hh, ww, c, device=dev).to(torch.bfloat16)
    wg = torch.randn(c, 8, 3, 3, device=dev) * 0.1
    wk2, wd2 = K.pack_gconv_diag(wg)
    run = lambda: K.k_gemm_nt(xa, None, 5, (N, hh, ww), wk2, c, 64, 9, stats=True)
if os.environ.get("LAYER") == "out":            # seg output conv (phase form, fp32 logits): 64 -> 4 x 5 on the 256 x 512 low-res grid
    from multitask_hydranet_amd.ops import seg as S
    xo = torch.randn(N, 256, 512, 64, device=dev).to(torch.bfloat16)
    wo = torch.randn(5, 64, 3, 3, device=dev) * 0.05
    bo = torch.zeros(5, device=dev)
    wpo, wto, beo = S.pack_phase_weight(wo, 64, bo)
    oo = torch.empty((N, 512, 1024, 5), device=dev, dtype=torch.float32)
    run = lambda: K.k_gemm_nt(xo, None, 4, (N, 256, 512), wpo, 20, K.kp32(64), 9, bias=beo, out=oo, out_f32=True, ldc=20, img_stride=-5)
if os.environ.get("LAYER") == "outdgrad":       # data gradient of the seg output conv: 4 x 5 (padded to 24) -> 64 on the 256 x 512 low-res grid, folding epilogue
    from multitask_hydranet_amd.ops import seg as S
    wo = torch.randn(5, 64, 3, 3, device=dev) * 0.05
    wpo, wto, beo = S.pack_phase_weight(wo, 64, torch.zeros(5, device=dev))
    dzo = torch.randn(N, 256, 512, 24, device=dev).to(torch.bfloat16)
    ypo = torch.randn(N, 256, 512, 64, device=dev).to(torch.bfloat16)
    run = lambda: S.k_dgrad_fold(dzo, wto, N, 256, 512, 64, K.kp32(20), 0, 1, ypo)
buf = torch.zeros(256 * 128, device=dev, dtype=torch.int64)
for pipe in (0,):
    lib().query("hn_debug_direct_pipe", pipe)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    for dbg in (16,):
        buf.zero_()
        lib().query("hn_debug_knob", 15, buf.data_ptr())
        lib().query("hn_debug_knob", 14, dbg)
        run()
        torch.cuda.synchronize()
        lib().query("hn_debug_knob", 14, 0)
        b = buf.view(256, 128).cpu()
        print(f"--- pipe{pipe} dbg {dbg} (cycles at 100 MHz s_memtime? -> deltas)")
        t0 = int(b[:16, 0][b[:16, 0] > 0].min())
        for blk in (0, 1, 2, 3, 8, 15, 40, 100):
            row = [int(v) for v in b[blk] if int(v) > 0]
            if len(row) < 3:
                continue
            d = [row[i + 1] - row[i] for i in range(len(row) - 1)]
            print(f"blk {blk * 64}: start+{row[0] - t0} total {row[-1] - row[0]} | prologue {d[0]} | iters {d[1:-5]} | tail {d[-5:]}")
lib().query("hn_debug_direct_pipe", 0)
