"""Best case for VERDICT r4 item 1 (weight-gradient tiles co-resident with the latency-bound dgrad chain): the two run as SEPARATE kernels (each
with its own register / LDS footprint -- a role-split launch could only be worse) on two HIP streams at the same time.
  chain: a dependent chain shaped like a stage-4 XBlock backward -- 1x1 GEMM 2048 x 936 x 936 with the statistics epilogue, fused BatchNorm
         apply pass, repeated; one linear hipGraph on stream A;
  side : the stage-4 grouped weight-gradient flush (29 jobs, 104 GFLOP; hn_wgrad_group), a linear hipGraph on stream B, repeated to last
         about as long as the chain.
Reports: chain alone, side alone, both at once (wall, and the chain's own elapsed time), and what the overlap recovered.
usage: python tools/overlap_probe.py [chain_pairs=40]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from multitask_hydranet_amd import ops as K
dev = "cuda:0"
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
m, c = 2048, 936
x = torch.randn(1, 1, m, c, device=dev).bfloat16()
wgt = torch.randn(c, c, 1, 1, device=dev) * c ** -0.5
wp, _ = K.pack_conv_weight(wgt)
gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)

def chain():
    t = x
    for _ in range(pairs):
        z, ps, pq = K.k_gemm_nt(t, None, 0, (1, 1, m), wp, c, K.kp32(c), 1, stats=True)
        t, _, _, _ = K.k_bn_apply_fused(z, ps, pq, m, gam, bet, 1e-5, 0.1, rm, rv, K.ACT_RELU, training=True)
    return t

jobs = [((16, 8, 16), 376, 936, 1), ((16, 16, 32), 376, 936, 0)] + [((16, 8, 16), 936, 936, 0)] * 27
group = K.WgradGroup()
ws = []
for (n, h, w), cin, cout, mode in jobs:
    hi, wi = (2 * h, 2 * w) if mode == 1 else (h, w)
    ws.append((torch.empty(cout, cin, 1, 1, device=dev), torch.randn(n, hi, wi, cin, device=dev).bfloat16(), torch.randn(n, h, w, cout, device=dev).bfloat16(),
               mode, (n, h, w), cin, cout))

def side_once():
    group.weights = tuple(w[0] for w in ws)
    for w in ws:
        group.add(*w)
    return group.flush()

def capture(fn, stream):
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        fn(); fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=stream):
        keep = fn()
    return gr, keep

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
g_chain, k1 = capture(chain, sa)

def timed(fa, fb, reps=10):
    """median wall (events on the default stream around both) and the chain's own elapsed time"""
    walls, chains = [], []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        e0, e1, c0, c1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e0.record()
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        if fa is not None:
            with torch.cuda.stream(sa):
                c0.record(); fa(); c1.record()
        if fb is not None:
            with torch.cuda.stream(sb):
                fb()
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
        e1.record()
        torch.cuda.synchronize()
        walls.append(e0.elapsed_time(e1) * 1e3)
        chains.append(c0.elapsed_time(c1) * 1e3 if fa is not None else 0.0)
    walls, chains = sorted(walls[2:]), sorted(chains[2:])
    return walls[len(walls) // 2], chains[len(chains) // 2]

t_chain, _ = timed(g_chain.replay, None)
g1, k2 = capture(side_once, sb)
t1, _ = timed(None, g1.replay)
nside = max(1, round(t_chain / t1))
def side_n():
    out = None
    for _ in range(nside):
        out = side_once()
    return out
g_side, k3 = capture(side_n, sb)
t_side, _ = timed(None, g_side.replay)
t_both, t_chain_in_both = timed(g_chain.replay, g_side.replay)
print("chain: %d x (GEMM 2048x936x936 + statistics, fused BatchNorm apply) = %d dependent launches" % (pairs, 2 * pairs))
print("chain alone            %8.1f us  (%.1f us per launch)" % (t_chain, t_chain / (2 * pairs)))
print("side alone (%d flushes) %8.1f us  (%.1f us per stage-4 weight-gradient flush)" % (nside, t_side, t_side / nside))
print("serial sum             %8.1f us" % (t_chain + t_side))
print("both at once           %8.1f us wall; the chain itself took %.1f us (%.2fx its solo time)" % (t_both, t_chain_in_both, t_chain_in_both / t_chain))
print("recovered by overlap   %8.1f us = %.0f %% of the shorter of the two" % (t_chain + t_side - t_both, 100.0 * (t_chain + t_side - t_both) / min(t_chain, t_side)))
