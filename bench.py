#!/usr/bin/env python3
"""bench.py -- images/sec for HydraNet forward+loss+backward on MI355X (BASELINE.json metric), HIP path only.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one forward + multitask loss + backward over one synthetic batch that is already resident in HBM (the optimizer step is
not part of the reference's fwd+bwd metric; its time is reported separately as ms_optimizer_step).  At N=1 the workload is
BASELINE.json configs[2]: full HydraNet (big cfg), batch 16, 3x512x1024, bf16 compute.  For N>1 every rank runs the same per-GPU
batch (weak scaling) and the gradients are averaged with the bucketed RCCL all-reduce of multitask_hydranet_amd.ddp.
Rank 0 prints ONE JSON line:
  * value / ms_per_step: the K timed steps bracketed by barrier + device sync on both sides (wall clock, max over ranks);
    ms_per_step_median: median of the per-step HIP-event durations recorded on the launch stream inside the same timed loop;
  * roofline: the dominant launch (largest seg-decoder conv), timed live with HIP events;
  * segments: {backbone, neck, seg, det, lane, losses}: {ms, floor_ms, frac} -- each segment's forward + backward as its own captured
    hipGraph (ablation: backbone, backbone + neck, forward-only, head-only backward phases), floors from accounting.segment_floors_ms;
  * extra_configs (N=1 only): the other BASELINE configurations on the same device, ~0.5 s each: backbone-only N=8, the repo-default
    640x640, inference 1152x1920 N=32, the gradient exchange at world size 1, the three head-only fine-tuning phases;
  * cpu_baseline: the fp32 oracle (a port, oracle/hydranet_oracle.py) timed on this box's host cores on a bounded sample;
  * env_overrides: every HN_* environment variable that was set (tools/ hooks change what runs; none of them skips work in the timed region).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import yaml  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_IMG_512x1024 = 81.13   # BASELINE.md section 3 (conv MACs x 2); fwd+bwd = 3x
METRIC = "images/sec (fwd+bwd) HydraNet @ default res, 1/2/4/8 MI355X; CPU-ref same run"


def emit(res):
    """the JSON line must be the LAST line on stdout: RCCL leaves a version banner in the C stdio buffer (printed at exit when stdout is a
    pipe), so the C streams are flushed first"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass
    print(json.dumps(res), flush=True)


def synthetic_batch(cfgs, n, h, w, seed, device):
    """SURVEY.md section 8(d) synthetic inputs (same recipe as oracle.synthetic_batch, restated so the product path never imports oracle/)."""
    g = torch.Generator().manual_seed(seed)
    stride, interval = cfgs["lane"]["anchor_stride"], cfgs["lane"]["interval"]
    hw, ppl = (h // stride) * (w // stride), h // interval
    image = torch.randn(n, 3, h, w, generator=g)
    gt_seg = torch.randint(0, len(cfgs["segment"]["class_list"]), (n, h, w), generator=g).float()
    m = 16
    cx, cy = torch.rand(n, m, generator=g) * w, torch.rand(n, m, generator=g) * h
    bw, bh = 16 + torch.rand(n, m, generator=g) * 240, 16 + torch.rand(n, m, generator=g) * 240
    cls = torch.randint(0, cfgs["detection"]["num_classes"], (n, m), generator=g).float()
    gt_det = torch.stack([(cx - bw / 2).clamp(0, w - 1), (cy - bh / 2).clamp(0, h - 1), (cx + bw / 2).clamp(0, w - 1),
                          (cy + bh / 2).clamp(0, h - 1), cls], dim=2)
    for i in range(n):
        if i % 8 == 7:
            gt_det[i] = -1.0
    gt_cls = torch.zeros(n, hw, 2)
    gt_cls[..., 0] = 1.0
    gt_loc = torch.zeros(n, hw, 2 * ppl + 2)
    for i in range(n):
        idx = torch.randperm(hw, generator=g)[:8]
        gt_cls[i, idx, 0], gt_cls[i, idx, 1] = 0.0, 1.0
        gt_loc[i, idx] = torch.randn(8, 2 * ppl + 2, generator=g)
    return {k: v.to(device) for k, v in dict(image=image, gt_seg=gt_seg, gt_det=gt_det, gt_cls=gt_cls, gt_loc=gt_loc).items()}


# ------------------------------------------------------------------------------------------------------------------------------------
# dominant launch
# ------------------------------------------------------------------------------------------------------------------------------------
def dominant_launch_roofline(net, n, h, w, iters=100, graph_timing=False):
    """Time the largest seg-decoder launch with HIP events on the stream it is launched on (the launches of the timed region are replayed
    from one captured hipGraph, as they run inside the training step), and price it against the dense bf16 MFMA peak.
    decoder.3 of the big cfg = Conv3x3(reflect-pad(cat[up2(x 256 ch), P3 112 ch])) -> 256 @ (h/8) x (w/8).  Its up-sampled operand runs in
    PHASE form on the low-resolution grid (hn_conv3x3_phase: 4 of 9 taps per output phase, the skip operand's partial sum arrives as a
    pre-activation addend): that launch is timed here.  `achieved` prices the ALGORITHMIC flops of the convolution it replaces
    (2*N*H*W*Cout*C0*9, SURVEY 8(d) conv-MAC figure); `executed_tflops` is what the MFMA pipe actually ran (16/36 of it)."""
    from multitask_hydranet_amd import ops as K
    from multitask_hydranet_amd._lib import lib
    P = net._idx
    wgt = P["segheader.decoder.3.conv.conv.weight"]
    cout, cin = wgt.shape[0], wgt.shape[1]
    c1 = net.fpn_num_filters
    c0 = cin - c1
    hh, ww = h // 8, w // 8                                         # P3 resolution (stride 8) = output resolution
    dev = wgt.device
    x0 = torch.randn(n, hh // 2, ww // 2, c0, device=dev).to(torch.bfloat16)
    x1 = torch.randn(n, hh, ww, c1, device=dev).to(torch.bfloat16)
    bias = P["segheader.decoder.3.conv.conv.bias"]
    out = torch.empty(n, hh, ww, cout, device=dev, dtype=torch.bfloat16)
    phase = K.seg_up_phase_ok(x0, x1, wgt)
    if phase:
        with torch.no_grad():
            wpe, _, be = K.pack_phase_weight(wgt, c0, bias)
            wp1, _ = K.pack_conv_weight_slice(wgt, c0, c1)
            z1, _, _ = K.k_gemm_nt(x1, None, 2, (n, hh, ww), wp1, cout, K.kp32(c1), 9)
        run = lambda: lib().call("hn_conv3x3_phase", x0.data_ptr(), 4, n, hh // 2, ww // 2, c0, c0, wpe.data_ptr(), 4 * cout, K.kp32(c0),
                                 be.data_ptr(), K.ACT_ELU, out.data_ptr(), cout, cout, z1.data_ptr(), cout)
        flops = 2.0 * n * hh * ww * cout * c0 * 9
        executed = flops * 16.0 / 36.0
        alg_bytes = 2.0 * (n * (hh // 2) * (ww // 2) * c0 + 2 * n * hh * ww * cout + 4 * cout * c0 * 9)
        form = "phase"
        kname = "conv3x3_direct_kernel<128,false,false> phase form: seg decoder.3 up-sampled operand (256 ch @ %dx%d -> 4 phases x 256 @ %dx%d, " \
                "+ skip addend, ELU), N=%d, fwd" % (hh // 2, ww // 2, hh, ww, n)
    else:
        wp, _ = K.pack_conv_weight(wgt)
        run = lambda: K.k_gemm_nt(x0, x1, 2, (n, hh, ww), wp, cout, K.kp32(cin), 9, bias=bias, act=K.ACT_ELU, out=out, up=1)
        flops = executed = 2.0 * n * hh * ww * cout * cin * 9
        alg_bytes = 2.0 * (n * (hh // 2) * (ww // 2) * c0 + n * hh * ww * c1 + n * hh * ww * cout + cout * cin * 9)
        form = "full"
        kname = "conv3x3_direct_kernel<128,false,false> seg decoder.3 (reflect-pad 3x3 over cat[up2(x), skip], 368->256 @ %dx%d, N=%d) fwd" % (hh, ww, n)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    g = None
    if graph_timing:           # the `iters` launches as one captured hipGraph: back-to-back dispatches as inside the training step's graph
        try:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                run()
            torch.cuda.current_stream().wait_stream(s_)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    run()
            g.replay()
            torch.cuda.synchronize()
        except Exception:       # noqa: BLE001
            g = None
            torch.cuda.synchronize()
    e0.record()
    if g is not None:
        g.replay()
    else:
        for _ in range(iters):
            run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ach = flops / (ms * 1e-3) / 1e12
    traffic, source = measured_traffic(n, h, w, form)
    if _LIVE_TRAFFIC.get("bytes") and form == "phase":
        traffic, source = _LIVE_TRAFFIC["bytes"], _LIVE_TRAFFIC["source"]
    elif _LIVE_TRAFFIC.get("error") and source:
        source += " (live sampling: %s)" % _LIVE_TRAFFIC["error"]
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_source": source, "kernel": kname, "launch_ms": round(ms, 4), "flop_per_launch": flops,
            "executed_flop_per_launch": executed, "executed_tflops": round(executed / (ms * 1e-3) / 1e12, 2),
            "frac_executed": round(executed / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "algorithmic_bytes_per_launch": alg_bytes}


def time_dominant_gemm_family(net, n, h, w, iters=20):
    """The launches that take the most TIME in the step since round 6: the persistent stage launches (csrc/hn_xstage.hip: the stride-1
    identity XBlocks of backbone stages 3 and 4, forward and backward, one launch each; profiles/r06_roofline_table.md: 2.49 ms of the
    step, the next family is the 64 x 64 1x1 GEMM at 1.07 ms).  Each launch is built on the net's own blocks (random input of the stage's
    shape), captured alone in a hipGraph and timed with HIP events on the launch stream.  Algorithmic bytes per launch = SURVEY 8(d)'s conv
    operand figure of the convs it contains: forward, per block, (X + Y) of the two 1x1 convs and of the grouped 3x3 conv = 6 M C 2 bytes
    + the packed weights once; backward (data gradients only -- the weight gradients stay separate launches) the same 6 M C 2 + weights."""
    from multitask_hydranet_amd import ops as K
    dev = net._idx["backbone.net.stem.conv.weight"].device
    per, tot_t, tot_b = [], 0.0, 0.0
    p = "backbone.net."
    for k in (3, 4):
        if k >= len(net.widths):
            continue
        c, d = net.widths[k], net.depths[k]
        hh, ww = h >> (k + 2), w >> (k + 2)
        x = torch.randn(n, hh, ww, c, device=dev).to(torch.bfloat16).relu_()
        q0 = "%sstage_%d.blocks.block_1." % (p, k)
        if d < 2 or not K.xstage_ok(x, net._idx[q0 + "conv_block_1.0.weight"], net._idx[q0 + "se.1.weight"].shape[0]):
            continue
        params = []
        for j in range(1, d):
            params += [t.detach().clone() if i % 19 in (3, 4, 8, 9, 17, 18) else t.detach()
                       for i, t in enumerate(net._xblock_params("%sstage_%d.blocks.block_%d." % (p, k, j)))]
        nb = d - 1
        sws = [(params[b * 19 + 10], params[b * 19 + 12]) for b in range(nb)]
        dout = (torch.randn(n, hh, ww, c, device=dev) * 0.01).to(torch.bfloat16)
        with torch.no_grad():
            r = K.xstage_forward_raw(x, params, 1e-5, 0.1)
        m = n * hh * ww
        byts = nb * (6.0 * m * c * 2 + (2 * c * K.kp32(c) + c * 576) * 2.0)
        for what, fn in (("forward", lambda: K.xstage_forward_raw(x, params, 1e-5, 0.1)),
                         ("backward", lambda: K.xstage_backward_raw(dout, r, r["packs"], sws))):
            with torch.no_grad():
                for _ in range(2):
                    fn()
                s_ = torch.cuda.Stream()
                s_.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s_):
                    fn()
                torch.cuda.current_stream().wait_stream(s_)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(iters):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            per.append({"launch": "stage_%d %s, %d identity blocks (%d ch, %d x %d maps)" % (k, what, nb, c, hh, ww), "rows": m, "launch_us": round(us, 1),
                        "us_per_block": round(us / nb, 2), "algorithmic_bytes": byts, "GBps": round(byts / us / 1e3, 1),
                        "frac": round(byts / us / 1e3 / PEAK_HBM_GBS, 4), "TFLOPs": round(nb * 2.0 * m * (2 * c * c + 72 * c) / us / 1e6, 1)})
            tot_t += us
            tot_b += byts
            del g
        K.xstage_assert_ok(dev)
    if not per:
        return {"error": "no persistent stage launch at this shape"}
    ach = tot_b / tot_t / 1e3
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
            "kernel": "xstage_fwd_kernel / xstage_bwd_kernel (persistent stage launches), N=%d" % n,
            "launches": per,
            "note": "largest launches by TIME in the step.  Not bandwidth bound in the roofline sense: a block is two GEMM phases at the L2 -> LDS "
                    "feed rate plus six inter-workgroup exchanges (DESIGN.md section 4); the launch chain they replace took 1.4-1.8x as long "
                    "(profiles/r06_stage_persistent.md)"}


def measured_traffic(n, h, w, form):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction +
    WRITE_SIZE, separate passes: PMC counters cannot be read from inside this process).  The figure belongs to ONE workload: it is only
    emitted when batch, resolution and kernel form match what the profile recorded; otherwise null."""
    for name in ("r06_dominant_pmc.json", "r05_dominant_pmc.json", "r04_dominant_pmc.json", "r03_dominant_pmc.json", "r02_dominant_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:       # noqa: BLE001
            continue
        rec = d.get("workload", {"batch": 16, "res": "512x1024", "form": "phase"})        # (r02 file: recorded before the field existed)
        if rec.get("batch") == n and rec.get("res") == "%dx%d" % (h, w) and rec.get("form") == form:
            return d["hbm_bytes_per_launch"], "profiles/" + name
    return None, None


_LIVE_TRAFFIC = {}     # filled by sample_traffic_live() at the start of main(): {"bytes": ..., "source": ...} or {"error": ...}


def sample_traffic_live(args, timeout_s=150):
    """HBM bytes per launch of the dominant kernel sampled IN THIS RUN: two child processes `rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE>
    --kernel-trace -- python3 bench.py --dominant-only --steps 10` (separate passes, program directly behind `--`; started before this
    process touches the GPU), counters corrected as MI355X_MICROARCH.md prescribes (both in KB; gfx950 tallies 128-byte read requests at 64 B:
    FETCH_SIZE x 2).  Bounded: any failure or a pass over its time limit leaves the committed profile's figure in place (`traffic_source`
    says which one the line carries).  HN_BENCH_LIVE_TRAFFIC=0 / --no-live-traffic: skip."""
    import csv, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}
    vals = {}
    t_start = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = timeout_s - (time.perf_counter() - t_start)
        if left < 20:
            return {"error": "time limit"}
        with tempfile.TemporaryDirectory(prefix="hn_pmc_", dir="/tmp") as d:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--dominant-only", "--steps", "10", "--batch", str(args.batch), "--res", args.res, "--cfg", args.cfg]
            env = dict(os.environ, TMPDIR="/tmp", HN_BENCH_LIVE_TRAFFIC="0")
            try:
                pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    pr.wait(timeout=left)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)             # the process group we started (the profiler and its child), nothing else
                    pr.wait()
                    return {"error": counter + " pass over its time limit"}
                if pr.returncode != 0:
                    return {"error": "%s pass exited with %d" % (counter, pr.returncode)}
                rows = []
                for root, _, files in os.walk(d):
                    for f in files:
                        if f.endswith("counter_collection.csv"):
                            rows += [r for r in csv.DictReader(open(os.path.join(root, f)))
                                     if "conv3x3_direct" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
                v = [float(r["Counter_Value"]) for r in rows]
                v = v[4:] if len(v) > 8 else v                      # the skip-operand conv and the warm-up launches
                if not v:
                    return {"error": counter + ": no rows of the dominant kernel"}
                vals[counter] = sum(v) / len(v)
            except Exception as e:       # noqa: BLE001
                return {"error": "%s: %s" % (type(e).__name__, e)}
    return {"bytes": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --dominant-only --steps 10` inside this run "
                      "(%.0f s)" % (time.perf_counter() - t_start)}


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU box reports 256 logical
    CPUs but grants 16; running 256 threads against that quota is ~1000x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:       # noqa: BLE001
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:       # noqa: BLE001
        pass
    return "unknown"


def cpu_baseline(cfgs, h, w, budget_s=20.0):
    """fp32 CPU oracle (port of the reference path) on this box's host cores: N=1 fwd+loss+bwd, bounded sample (3 warm-up +
    up to 10 timed iterations, median; the budget caps the timed ones)."""
    from oracle import hydranet_oracle as O
    import multitask_hydranet_amd as pkg
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = pkg.HydraNet(cfgs)                                         # CPU parameter container only (initial weights)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    batch = O.synthetic_batch(cfgs, 1, h, w, seed=1)
    ppl = h // cfgs["lane"]["interval"]

    def step():
        for v in sd.values():
            v.grad = None
        out = O.hydranet_forward(sd, cfgs, batch["image"], training=True)
        ld = O.hydranet_losses(cfgs, out, batch, lane_points_per_line=ppl)
        O.total_loss(cfgs, ld).backward()
    t0 = time.time()
    nwarm = 0
    while nwarm < 3 and (nwarm == 0 or time.time() - t0 < 0.3 * budget_s):
        step()
        nwarm += 1
    warm = time.time() - t0
    times = []
    while len(times) < 10 and (sum(times) + warm) < budget_s:
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    if not times:
        times = [warm / nwarm]
    times.sort()
    med = times[len(times) // 2]
    # one-thread figure: ONE timed iteration after the warm multi-thread runs (a second would double the bounded sample)
    torch.set_num_threads(1)
    t1 = time.time()
    step()
    one = time.time() - t1
    torch.set_num_threads(cores)
    return {"value": round(1.0 / med, 4), "unit": "images/sec", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "value_1thread": round(1.0 / one, 4),
            "sample": "fp32 oracle, big cfg, N=1, 3x%dx%d, fwd+loss+bwd, %d warm-up + %d timed iterations (median) on %d threads; "
                      "1 iteration on 1 thread" % (h, w, nwarm, len(times), cores)}


# ------------------------------------------------------------------------------------------------------------------------------------
# timing helpers
# ------------------------------------------------------------------------------------------------------------------------------------
def capture(fn, eager_warmup=2):
    """two eager runs on a side stream (allocator warm-up, lazy packs), then `fn` captured as one hipGraph -> (graph, fn's return value)"""
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(eager_warmup):
            fn()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = fn()
    return g, out


def time_replays(step, steps, warmup, world=1):
    """`warmup` untimed + EXACTLY `steps` timed calls of step(), bracketed by barrier + device synchronize on both sides (wall clock), with
    one HIP event recorded on the launch stream before every call and after the last -> (seconds, [per-step ms from the events])"""
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i].record()
        step()
    ev[steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    return dt, [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def build_net(cfg_path, h, w, dev):
    from multitask_hydranet_amd import HydraNet
    cfgs = yaml.safe_load(open(cfg_path))
    cfgs["dataloader"]["network_input_height"], cfgs["dataloader"]["network_input_width"] = h, w
    torch.manual_seed(0)
    net = HydraNet(cfgs).to(dev).train()
    net.check_finite = False                       # the reference's exit()-on-NaN guard is a host sync; checked once after the run instead
    net.lane_points_per_line = h // cfgs["lane"]["interval"]     # the reference default (160) raises IndexError at H=512
    return net, cfgs


def quick_graph_ms(fn, reps=10, warm=2):
    """median per-replay milliseconds (HIP events) of `fn` captured as one hipGraph"""
    g, _ = capture(fn)
    _, per = time_replays(g.replay, reps, warm)
    del g
    return median(per)


# ------------------------------------------------------------------------------------------------------------------------------------
# inference (BASELINE config 5)
# ------------------------------------------------------------------------------------------------------------------------------------
def infer_measure(net, batch_n, h, w, dev, steps, warmup, use_graph=True, world=1, seed=1):
    net.eval()
    net.prepare_inference()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch_n, 3, h, w, generator=g).to(dev)
    with torch.no_grad():
        fn = lambda: net(x, "deploy")
        if use_graph:
            graph, out = capture(fn)
            step = graph.replay
        else:
            graph, out = None, fn()
            step = fn
        dt, per = time_replays(step, steps, warmup, world)
    assert bool(torch.isfinite(out[2]).all()) and out[0].dtype == torch.int64
    return dt, per, graph is not None


def infer_bench(args, net, cfgs, h, w, dev, rank, world):
    """BASELINE config 5: inference-only deploy forward (seg arg-max + detection / lane head outputs), batch per GPU, replicas only (no
    collective on the data path).  One captured hipGraph per step; BatchNorm folded into the packed weights (HydraNet.prepare_inference)."""
    dt, per, graphed = infer_measure(net, args.batch, h, w, dev, args.steps, args.warmup, not args.no_graph, world, seed=1 + rank)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    dt = float(tmax)
    if rank != 0:
        return
    value = args.batch * world * args.steps / dt
    scale = (h * w) / (512.0 * 1024.0)
    res = {"metric": "images/sec (inference fwd) HydraNet, deploy mode, BASELINE config 5", "value": round(value, 2), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "ms_per_step_median": round(median(per), 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "full HydraNet eval-mode deploy forward (seg arg-max, det + lane head outputs), big cfg, BatchNorm folded",
                      "batch_per_gpu": args.batch, "global_batch": args.batch * world, "resolution": "3x%dx%d" % (h, w),
                      "parallelism": "replicas x%d (no collective)" % world, "hipgraph": graphed,
                      "padding": "1080-row frames are zero-padded (post-normalisation) by 36 rows top and bottom to 1152 = 9 x 128"},
           "model_tflops": round(value * FWD_GFLOP_PER_IMG_512x1024 * scale / 1e3, 2), "env_overrides": env_overrides()}
    try:
        res["roofline"] = dominant_launch_roofline(net, args.batch, h, w)
    except Exception as e:      # noqa: BLE001
        res["roofline"] = {"error": repr(e)}
    emit(res)


def other_exchange_form(run, net, cfgs, batch, dev, rank, world, payload, args, dt_primary, headline=None):
    """Time the captured step with the gradient exchange in its other form (the headline used `run.reducer.graph_overlap`) -> dict with both
    forms' ms per step (max over ranks).  Only with --both-exchange-forms.  Every rank runs a watchdog: if the second capture / replay does
    not come back within HN_BENCH_FORM_TIMEOUT seconds (default 120), rank 0 prints the finished headline line with exchange_forms.error
    set and EVERY rank ends with exit code 4 -- a wedged collective costs the driver neither its measurement nor a false success."""
    import threading
    primary = "graph_overlap" if run.reducer.graph_overlap else "in_line"
    other = "in_line" if primary == "graph_overlap" else "graph_overlap"
    res = {primary: {"ms_per_step": round(dt_primary / args.steps * 1e3, 3), "headline": True}}
    done = threading.Event()
    lock = threading.Lock()                           # emit-or-exit is decided once, under the lock
    state = {"armed": True, "line": headline}         # rank 0: the finished headline line (printed by the watchdog if this wedges)

    def watchdog():
        if done.wait(float(os.environ.get("HN_BENCH_FORM_TIMEOUT", "120"))):
            return
        with lock:
            if not state["armed"]:
                return
            state["armed"] = False
            if rank == 0 and state.get("line") is not None:
                line = dict(state["line"])
                line["exchange_forms"] = dict(res, **{other: {"error": "timed out (watchdog): exit code 4"}})
                emit(line)
            os._exit(4)
    th = threading.Thread(target=watchdog, daemon=True)
    th.start()
    try:
        net.zero_grad(set_to_none=True)
        run2 = TrainRun(net, cfgs, batch, dev, rank, world, "nccl", use_graph=True, exchange=True, force_world1=args.ddp_world1, payload=payload,
                        graph_overlap=(other == "graph_overlap"))
        if not run2.in_graph_exchange:
            raise RuntimeError("the %s form was not captured" % other)
        steps = max(3, min(args.steps, 10))
        dt2, _ = time_replays(run2.step, steps, 2, world)
        t = torch.tensor([dt2], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[other] = {"ms_per_step": round(float(t) / steps * 1e3, 3), "steps": steps, "grad_allreduce": run2.describe_exchange()}
        del run2
    except Exception as e:          # noqa: BLE001
        res[other] = {"error": repr(e)[:300]}
    finally:
        with lock:                  # (a watchdog that already took the lock ends the process before this returns)
            state["armed"] = False
        done.set()
    return res



_CAPTURE_STREAM = None


def capture_stream():
    global _CAPTURE_STREAM
    if _CAPTURE_STREAM is None:
        _CAPTURE_STREAM = torch.cuda.Stream()
    return _CAPTURE_STREAM


def env_overrides():
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("HN_") or k == "HIPCC"}


# ------------------------------------------------------------------------------------------------------------------------------------
# the training step under measurement
# ------------------------------------------------------------------------------------------------------------------------------------
class TrainRun:
    """one configuration of the fwd + loss + bwd step: builds the (optionally DDP-exchanging) captured step and times it"""

    def __init__(self, net, cfgs, batch, dev, rank=0, world=1, backend="nccl", backbone_only=False, use_graph=True, exchange=False,
                 force_world1=False, payload=torch.float32, exchange_after_replay=False, graph_overlap=None):
        self.net, self.cfgs, self.batch, self.dev = net, cfgs, batch, dev
        self._graph_overlap = graph_overlap         # None: GradReducer's default (in line; HN_DDP_GRAPH_OVERLAP=1: forked side stream)
        self.rank, self.world, self.backend = rank, world, backend
        self.backbone_only = backbone_only
        self.graph = self.static_loss = self.reducer = None
        self.in_graph_exchange = False
        self.one = torch.ones((), device=dev)       # d loss / d loss, allocated once (loss.backward() alone fills a fresh one every step)
        from multitask_hydranet_amd.ddp import GradReducer, unused_parameters
        self._skip = unused_parameters(net)
        self._GradReducer = GradReducer
        self._payload, self._force = payload, force_world1
        if exchange:
            # captured exchange: gather + all-reduce of every bucket in line on the capture stream (ddp.py); two ~86 MB buckets: few, large
            # collectives for the point-to-point xGMI links; eager hook mode keeps DDP's 25 MiB granularity
            self.reducer = self._make_reducer(bucket_bytes=96 << 20) if use_graph and not exchange_after_replay else self._make_reducer()
        capture_failed = False
        if use_graph:
            try:
                # (only RCCL collectives can be captured: the gloo test hook exchanges after the replay)
                if self.reducer is not None and not exchange_after_replay and (backend == "nccl" or os.environ.get("HN_BENCH_TRY_CAPTURE") == "1"):
                    try:
                        self.graph, self.static_loss = self._capture(with_hooks=True)    # (.grad adopted, hooks removed by the recipe)
                        self.in_graph_exchange = self.reducer.captured
                    except Exception as e:              # noqa: BLE001  (an RCCL build that cannot be captured)
                        # a capture that failed half-way leaves PyTorch's graph bookkeeping unusable for a second capture in this process
                        # ("Cannot register the state during capturing stage"): run eager launches with the overlapped hook exchange instead
                        if rank == 0:
                            print("capturing the all-reduce inside the hipGraph failed (%r): eager launches with the hook exchange instead" % (e,),
                                  file=sys.stderr)
                        try:
                            torch.cuda.synchronize()
                        except Exception:               # noqa: BLE001  (a capture that could not be ended keeps its stream in capture mode)
                            pass
                        self.graph = None
                        capture_failed = True
                        self.reducer.remove()
                        self.reducer = self._make_reducer()
                if self.graph is None and not capture_failed:
                    if self.reducer is not None:
                        self.reducer.remove()           # no hooks during this capture; gradients are exchanged right after each replay
                    self.graph, self.static_loss = self._capture(with_hooks=False)
                    if self.reducer is not None:
                        # after a replay nothing is left to overlap with: ONE flat bucket = one gather + one all-reduce for all 693 gradients
                        self.reducer = self._make_reducer(bucket_bytes=1 << 40)
                        self.reducer.remove()
                        self.reducer.bind_static_grads()    # every replay rewrites these tensors; reduce_now() gathers them into the bucket
            except Exception as e:                      # noqa: BLE001
                if rank == 0:
                    import traceback
                    traceback.print_exc()
                    print("hipGraph capture failed, falling back to eager launches: %r" % (e,), file=sys.stderr)
                self.graph = None
                torch.cuda.synchronize()
                if exchange:
                    self.reducer = self._make_reducer()

    def _make_reducer(self, **kw):
        if os.environ.get("HN_BUCKET_MB"):                                   # tools/ sweeps of the exchange granularity
            kw["bucket_bytes"] = int(float(os.environ["HN_BUCKET_MB"]) * (1 << 20))
        return self._GradReducer(list(self.net.named_parameters()), world_size=self.world, skip=self._skip, payload_dtype=self._payload,
                                 force_collectives=self._force, graph_overlap=self._graph_overlap, **kw)

    def fwd_bwd(self):
        net, batch = self.net, self.batch
        if self.backbone_only:
            feats = net._backbone(batch["image"])
            loss = sum(f.float().mean() for f in feats)
            loss.backward(self.one)
            return loss
        out = net(batch["image"])
        ld = net.cal_loss(out, batch)
        loss = net.total_loss(ld)
        loss.backward(self.one)
        return loss

    def _capture(self, with_hooks):
        """two eager warm-up steps on a side stream, then the capture on the same stream.  with_hooks: the reducer's autograd hooks stay
        armed, so every bucket's gather + all-reduce is captured where its last gradient appears (ddp.capture_exchange_step: the recipe
        HydraTrainer(capture_step=True) runs at world size > 1)."""
        net, reducer = self.net, self.reducer
        s_ = capture_stream()                   # one stream for every warm-up and capture of this process
        zero = lambda: net.zero_grad(set_to_none=True)

        def fwd_bwd():
            l0 = self.fwd_bwd()
            if os.environ.get("HN_BENCH_DEBUG") and not torch.cuda.is_current_stream_capturing():
                print("rank", self.rank, "eager loss", float(l0.detach()), file=sys.stderr, flush=True)
            return l0
        if reducer is not None and with_hooks:
            from multitask_hydranet_amd.ddp import capture_exchange_step
            g, sl = capture_exchange_step(reducer, fwd_bwd, zero, s_, warmup=2)
            return g, sl.detach()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(2):
                zero()
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        zero()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s_, capture_error_mode="thread_local"):
            sl = self.fwd_bwd()
        return g, sl.detach()                   # (the value lives in graph memory; its autograd graph would pin AccumulateGrad nodes)

    def step(self):
        if self.graph is not None:
            self.graph.replay()
            if self.reducer is not None and not self.in_graph_exchange:
                self.reducer.reduce_now()
            return self.static_loss
        self.net.zero_grad(set_to_none=False) if self.reducer is not None else self.net.zero_grad(set_to_none=True)
        loss = self.fwd_bwd()
        if self.reducer is not None:
            self.reducer.finish()
        return loss

    def describe_exchange(self):
        if self.reducer is None:
            return None
        return ("%s backend: " % self.backend) + self.reducer.describe(after_replay=self.graph is not None and not self.in_graph_exchange)


# ------------------------------------------------------------------------------------------------------------------------------------
# per-segment table (ablation graphs) and the other BASELINE configurations
# ------------------------------------------------------------------------------------------------------------------------------------
def segment_table(net, cfgs, batch, n, h, w, full_ms):
    """{backbone, neck, seg, det, lane, losses}: {ms, floor_ms, frac} for the step under measurement, plus `unattributed_ms`.  Timing events
    cannot be recorded inside a captured hipGraph on ROCm, so every segment's forward + backward is measured as its OWN captured graph
    (median of 10 replays, HIP events), by ablation on the same module and batch:
      backbone        = fwd+bwd of the backbone alone (loss = sum of feature means)
      neck            = fwd+bwd of backbone + BiFPN (loss = sum of fused-map means)  -  backbone
      <head> forward  = that head alone on fixed fused maps, under no_grad
      <head> backward = head-only fine-tuning step (grad_scope: whole forward + losses, backward through that head only)  -  forward-only step
      losses          = the three losses + weighted total, forward + backward, on fixed head outputs
    frac = floor_ms / ms with the floors of accounting.segment_floors_ms (seg on the MFMA peak, the rest on HBM bytes)."""
    from multitask_hydranet_amd.accounting import segment_floors_ms
    from multitask_hydranet_amd import ops as K
    one = torch.ones((), device=batch["image"].device)
    x = batch["image"]

    def bb():
        net.zero_grad(set_to_none=True)
        feats = net._backbone(x)
        net._flush_nbt()
        sum(f.float().mean() for f in feats).backward(one)

    def bbn():
        net.zero_grad(set_to_none=True)
        fused = net._neck(net._backbone(x))
        net._flush_nbt()
        sum(f.float().mean() for f in fused).backward(one)

    def fwd_all():
        with torch.no_grad():
            out = net(x)
            return net.total_loss(net.cal_loss(out, batch))

    def scoped(head):
        def f():
            net.zero_grad(set_to_none=True)
            net.grad_scope = head
            try:
                net.total_loss(net.cal_loss(net(x), batch)).backward(one)
            finally:
                net.grad_scope = None
        return f
    t = {"backbone": quick_graph_ms(bb), "bbn": quick_graph_ms(bbn), "fwd_all": quick_graph_ms(fwd_all)}
    for head in ("seg", "det", "lane"):
        t["scope_" + head] = quick_graph_ms(scoped(head))
    with torch.no_grad():
        feats = net._backbone(x)
        fused = [f.detach() for f in net._neck(feats)]
        feat0 = feats[0].detach()
        out = net(x)
    heads_fwd = {"seg": lambda: net._seg([feat0, fused[0], fused[1], fused[2]]), "det": lambda: net._det(x, fused), "lane": lambda: net._lane(fused)}
    for head, fn in heads_fwd.items():
        def f(fn=fn):
            with torch.no_grad():
                return fn()
        t["fwd_" + head] = quick_graph_ms(f)
    leaf = {"seg": out["seg"].detach().requires_grad_(True),
            "detection": {"anchors": out["detection"]["anchors"], "regression": out["detection"]["regression"].detach().requires_grad_(True),
                          "classification": out["detection"]["classification"].detach().requires_grad_(True)},
            "lane": {k: v.detach().requires_grad_(True) for k, v in out["lane"].items()}}

    def losses():
        net.total_loss(net.cal_loss(leaf, batch)).backward(one)
    t["losses"] = quick_graph_ms(losses)
    net.zero_grad(set_to_none=True)
    ms = {"backbone": t["backbone"], "neck": t["bbn"] - t["backbone"], "losses": t["losses"]}
    for head in ("seg", "det", "lane"):
        ms[head] = t["fwd_" + head] + (t["scope_" + head] - t["fwd_all"])
    floors = segment_floors_ms(cfgs, h, w, n)
    table = {k: {"ms": round(v, 3), "floor_ms": round(floors[k], 3), "frac": round(floors[k] / v, 4) if v > 0 else None} for k, v in ms.items()}
    table["unattributed_ms"] = round(full_ms - sum(ms.values()), 3)
    table["method"] = "per-segment captured hipGraphs (ablation), median of 10 replays; raw: " + json.dumps({k: round(v, 3) for k, v in t.items()})
    return table


def extra_configs(args, dev, headline_net, headline_cfgs):
    """the other BASELINE.json configurations on this device, each a short driver-timed run (10 timed steps after 3 warm-up replays)"""
    res = {}
    K_STEPS, K_WARM = 10, 3

    def rec(name, fn):
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res[name] = fn()
            res[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:       # noqa: BLE001
            res[name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()

    def train_cfg(batch_n, h, w, backbone_only=False, exchange=False, scope=None, net=None, cfgs=None, what=""):
        own = net is None
        if own:
            net, cfgs = build_net(args.cfg, h, w, dev)
        batch = synthetic_batch(cfgs, batch_n, h, w, seed=1, device=dev)
        net.grad_scope = scope
        try:
            run = TrainRun(net, cfgs, batch, dev, backbone_only=backbone_only, exchange=exchange, force_world1=exchange)
            dt, per = time_replays(run.step, K_STEPS, K_WARM)
            loss = float(run.static_loss if run.graph is not None else run.step())
        finally:
            net.grad_scope = None
            net.zero_grad(set_to_none=True)
        assert loss == loss and abs(loss) != float("inf")
        out = {"value": round(batch_n * K_STEPS / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / K_STEPS * 1e3, 3),
               "ms_per_step_median": round(median(per), 3), "steps": K_STEPS, "warmup": K_WARM, "batch": batch_n, "resolution": "3x%dx%d" % (h, w),
               "hipgraph": run.graph is not None, "workload": what}
        if not backbone_only and scope is None:
            # SURVEY 8(d) segment-wise step roofline (0.177 ms/img at 512x1024, linear in H*W): the same fraction the headline's step_roofline states
            sc = (h * w) / (512.0 * 1024.0)
            out["step_roofline"] = {"floor_ms_per_img": round(0.177 * sc, 4), "frac": round(out["value"] * 0.177e-3 * sc, 4)}
        if exchange:
            out["grad_allreduce"] = run.describe_exchange()
        return out
    rec("config2_backbone_only_n8", lambda: train_cfg(8, 512, 1024, backbone_only=True, net=headline_net, cfgs=headline_cfgs,
                                                       what="BASELINE config 2: RegNetY backbone fwd+bwd, loss = sum of the five feature means"))
    rec("repo_default_640x640_n16", lambda: train_cfg(16, 640, 640, what="full HydraNet fwd+loss+bwd at the reference's default 640x640"))
    for head in ("lane", "det", "seg"):
        rec("finetune_phase_%s_n16" % head, lambda head=head: train_cfg(
            16, 512, 1024, scope=head, net=headline_net, cfgs=headline_cfgs,
            what="head-only fine-tuning phase (train.py:441-515): whole forward + six losses, backward through the %s head only" % head))

    def ddp1():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        return train_cfg(16, 512, 1024, exchange=True, net=headline_net, cfgs=headline_cfgs,
                         what="BASELINE config 4's exchange path at world size 1: RCCL ncclAvg all-reduce of the gradient buckets inside the step")
    rec("config4_ddp_exchange_world1_n16", ddp1)

    def infer():
        net, cfgs = build_net(args.cfg, 1152, 1920, dev)
        dt, per, graphed = infer_measure(net, 32, 1152, 1920, dev, K_STEPS, K_WARM)
        return {"value": round(32 * K_STEPS / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / K_STEPS * 1e3, 3),
                "ms_per_step_median": round(median(per), 3), "steps": K_STEPS, "warmup": K_WARM, "batch": 32, "resolution": "3x1152x1920",
                "hipgraph": graphed, "workload": "BASELINE config 5: eval-mode deploy forward, BatchNorm folded, 1080-row frames zero-padded to 1152"}
    rec("config5_inference_1152x1920_n32", infer)
    if dist.is_initialized() and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        with contextlib.suppress(Exception):
            dist.destroy_process_group()
    return res



def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` (N > 1) started WITHOUT torch.distributed.run: start the N ranks as a CHILD process
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>), relay what it
    prints (rank 0's JSON line stays the last line) and return its exit code.  Called before this process makes any HIP call of its own
    (torch.cuda.device_count() below may initialise the runtime on builds without amdsmi -- harmless here: the ranks are started as a
    CHILD process, never by exec)."""
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("HN_BENCH_ONE_DEVICE") != "1":
        print("bench.py --gpus %d: this node shows %d GPU(s); refusing to report a smaller run under that label" % (n, have), file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    last_json = None
    for line in child.stdout:
        if line.startswith("{") and '"metric"' in line:
            if last_json is not None:
                sys.stdout.write(last_json)
            last_json = line
        else:
            sys.stdout.write(line)
    rc = child.wait()
    if last_json is not None:
        sys.stdout.write(last_json if last_json.endswith("\n") else last_json + "\n")
    sys.stdout.flush()
    if rc == 0 and last_json is None:
        print("bench.py: the %d-rank child printed no JSON line" % n, file=sys.stderr)
        return 3
    return rc


# ------------------------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--res", default="512x1024")
    ap.add_argument("--cfg", default=os.path.join(ROOT, "cfgs", "hydranet_big.yml"))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra_configs and segments measurements (profiling runs)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the dominant-launch timing loop (counter passes: only the step's own launches)")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the separate Adam-step timing (profiling runs: keeps the optimizer's "
                    "state initialisation and multi-tensor kernels out of the kernel statistics)")
    ap.add_argument("--backbone-only", action="store_true", help="BASELINE config[1]: backbone fwd+bwd, loss = sum of feature means")
    ap.add_argument("--dominant-only", action="store_true", help="launch only the dominant kernel (for rocprofv3 --pmc passes) and exit")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profile instead of two rocprofv3 --pmc "
                    "child passes inside this run")
    ap.add_argument("--infer", action="store_true", help="BASELINE config 5: eval-mode deploy forward with folded BatchNorm, hipGraph-captured "
                    "(use with --res 1152x1920 --batch 32: a 1080-row frame is zero-padded by 36 rows top and bottom after normalisation)")
    ap.add_argument("--ddp-world1", action="store_true", help="run the gradient exchange (RCCL init, ncclAvg, side stream, in-graph capture) "
                    "at world size 1 -- exercises the N > 1 code path on a single GPU")
    ap.add_argument("--grad-payload", default="fp32", choices=("fp32", "bf16"), help="gradient all-reduce payload type")
    ap.add_argument("--exchange-after-replay", action="store_true", help="do not capture the all-reduce inside the hipGraph")
    ap.add_argument("--both-exchange-forms", action="store_true", help="N > 1 / --ddp-world1: after the headline, also capture and time the step with "
                    "the exchange in its OTHER form (forked onto the side stream instead of in line); a second capture of RCCL collectives in "
                    "one process, guarded by a watchdog (exit code 4 on a wedge)")
    ap.add_argument("--phase", default=None, choices=("lane", "det", "seg"), help="head-only fine-tuning phase (train.py:441-515)")
    args = ap.parse_args()
    h, w = (int(v) for v in args.res.split("x"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))                # N ranks as a child torch.distributed.run; nothing below runs in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to measure a different number of GPUs than the line would report"
              % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # roofline.traffic of the headline line, sampled in this run (children under rocprofv3, started before this process uses the GPU)
    if (world == 1 and rank == 0 and not (args.dominant_only or args.no_roofline or args.no_live_traffic or args.infer or args.backbone_only
                                          or args.phase or args.ddp_world1)
            and args.batch == 16 and args.res == "512x1024" and os.environ.get("HN_BENCH_LIVE_TRAFFIC", "1") != "0"
            and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) and "rocprof" not in os.environ.get("LD_PRELOAD", "")):   # (not nested)
        import __graft_entry__ as ge0
        ge0.build()                                   # (the children must find the library built: they run under a time limit)
        _LIVE_TRAFFIC.update(sample_traffic_live(args))
    # test hooks for a 1-GPU box: HN_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and HN_BENCH_BACKEND=gloo replaces RCCL (which refuses
    # two ranks on one device), so the whole N>1 control flow can be exercised without a second GPU.  Never set by the driver.
    if os.environ.get("HN_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("HN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    exchange = world > 1 or args.ddp_world1
    if exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from multitask_hydranet_amd.ddp import broadcast_state

    net, cfgs = build_net(args.cfg, h, w, dev)
    broadcast_state(net)
    if os.environ.get("HN_TN"):                      # tools/ sweeps: "bc,bn,splits" -> hn_debug_tn_config
        from multitask_hydranet_amd._lib import lib
        lib().query("hn_debug_tn_config", *[int(v) for v in os.environ["HN_TN"].split(",")])
    if os.environ.get("HN_KNOBS"):                   # tools/ sweeps: "id=value,id=value" -> hn_debug_knob
        from multitask_hydranet_amd._lib import lib
        for kv in os.environ["HN_KNOBS"].split(","):
            k, v = kv.split("=")
            lib().query("hn_debug_knob", int(k), int(v))
    if os.environ.get("HN_DIRECT_PIPE"):             # tools/ A/B: 0 = two-buffer tap loop of the direct 3x3 conv, 1 = 32-channel ring form
        from multitask_hydranet_amd._lib import lib
        lib().query("hn_debug_direct_pipe", int(os.environ["HN_DIRECT_PIPE"]))
    if os.environ.get("HN_HEADS_SIDE") == "1":       # experiment hook: det + lane heads on a side stream (a hipGraph branch) next to the seg decoder
        net.heads_on_side_stream = True
    if args.dominant_only:
        print(json.dumps(dominant_launch_roofline(net, args.batch, h, w, iters=args.steps, graph_timing=False)))   # eager launches for the PMC passes
        return
    if args.infer:
        return infer_bench(args, net, cfgs, h, w, dev, rank, world)
    batch = synthetic_batch(cfgs, args.batch, h, w, seed=1 + rank, device=dev)
    net.grad_scope = args.phase
    payload = torch.bfloat16 if args.grad_payload == "bf16" else torch.float32
    run = TrainRun(net, cfgs, batch, dev, rank, world, backend, backbone_only=args.backbone_only, use_graph=not args.no_graph,
                   exchange=exchange, force_world1=args.ddp_world1, payload=payload, exchange_after_replay=args.exchange_after_replay)

    def step():
        loss = run.step()
        if os.environ.get("HN_BENCH_DEBUG"):
            print("rank", rank, "loss", float(loss.detach()), file=sys.stderr, flush=True)
        return loss
    dt, per = time_replays(step, args.steps, args.warmup, world)
    loss = run.static_loss if run.graph is not None else step()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    if world > 1:
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(t) / args.steps * 1e3, 3) for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(loss.detach())
    # the OTHER captured exchange form on the same ranks: in line on the capture stream (default) vs forked onto the side
    # stream (north_star's "overlapped with the next backward on a side HIP stream").  Timed after the headline so that it cannot disturb
    # it; a watchdog on every rank ends the process with the headline line if a second capture of RCCL collectives wedges.
    both_forms = exchange and backend == "nccl" and run.in_graph_exchange and args.both_exchange_forms
    if both_forms and rank != 0:
        other_exchange_form(run, net, cfgs, batch, dev, rank, world, payload, args, dt)
    # What the exchange costs on the step's critical path: the same step captured WITHOUT the exchange (no collective inside
    # this second capture, so it is safe in-process) and timed on every rank; exposed = step with - step without, max over ranks.  The
    # reducer's default form is in line on the capture stream; when the exposed time exceeds 3 % of the step the line says so
    # (`exchange_advice`): the forked side-stream form (HN_DDP_GRAPH_OVERLAP=1) and the bf16 payload (--grad-payload bf16) are the two knobs.
    exposed = None
    if exchange and run.graph is not None and run.in_graph_exchange and not os.environ.get("HN_BENCH_GRAD_NORM"):
        try:
            from multitask_hydranet_amd.ddp import settle_collectives
            settle_collectives(dev)
            run_plain = TrainRun(net, cfgs, batch, dev, rank, world, backend, backbone_only=args.backbone_only, use_graph=True, exchange=False)
            if run_plain.graph is not None:
                n_pl = max(5, args.steps // 2)
                dt_pl, _ = time_replays(run_plain.graph.replay, n_pl, 2, world)
                t_pl = torch.tensor([dt_pl / n_pl], device=dev, dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(t_pl, op=dist.ReduceOp.MAX)
                ms_plain = float(t_pl) * 1e3
                ms_with = dt / args.steps * 1e3
                exposed = {"ms_step_with_exchange": round(ms_with, 3), "ms_step_without_exchange": round(ms_plain, 3),
                           "exchange_exposed_ms": round(ms_with - ms_plain, 3), "exposed_frac_of_step": round((ms_with - ms_plain) / ms_with, 4),
                           "form": "graph_overlap (forked side stream)" if run.reducer.graph_overlap else "in line on the capture stream",
                           "payload": str(payload).replace("torch.", "")}
                if (ms_with - ms_plain) / ms_with > 0.03:
                    exposed["exchange_advice"] = ("exposed exchange > 3 % of the step: try HN_DDP_GRAPH_OVERLAP=1 (forked side stream) and "
                                                  "--grad-payload bf16 (half the bytes per link)")
            del run_plain
        except Exception as e:          # noqa: BLE001  (the headline stands without it)
            exposed = {"error": repr(e)[:300]}
    grad_norm = None
    if os.environ.get("HN_BENCH_GRAD_NORM"):        # tests: the gradients after the (possibly in-graph) exchange of the last step
        grad_norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None)))
    assert loss_val == loss_val and abs(loss_val) != float("inf"), "non-finite loss after the timed run"
    net.grad_scope = None

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        # the persistent stage launches never hang: an expired wait raises a status word and leaves invalid outputs -- a timed run with one
        # is not a measurement (checked here, after the timed region's synchronisation)
        from multitask_hydranet_amd.ops import xstage_assert_ok
        xstage_assert_ok(torch.device("cuda", torch.cuda.current_device()))
        # optimizer step, reported separately (Adam as in model/train.py:147)
        ms_opt = None
        if not args.no_optimizer:
            from multitask_hydranet_amd.optim import Adam       # torch.optim.Adam's rule, all tensors in one launch (HN_TORCH_ADAM=1: torch's)
            opt = (torch.optim.Adam if os.environ.get("HN_TORCH_ADAM") else Adam)([p for p in net.parameters() if p.grad is not None], 1e-5,
                                                                                  weight_decay=1e-8)
            for _ in range(2):
                opt.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                opt.step()
            torch.cuda.synchronize()
            ms_opt = (time.perf_counter() - t1) / 5 * 1e3
            del opt
        scale = (h * w) / (512.0 * 1024.0)
        gflop_img = 3 * FWD_GFLOP_PER_IMG_512x1024 * scale * (12.02 / 81.13 if args.backbone_only else 1.0)
        what = "RegNetY backbone only" if args.backbone_only else "full HydraNet (backbone + BiFPN + seg/det/lane heads + multitask loss)"
        if args.phase:
            what += ", head-only fine-tuning phase '%s' (backward through that head only)" % args.phase
        res = {
            "metric": METRIC, "value": round(value, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "ms_per_step_median": round(median(per), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": what + ", big cfg, fwd+loss+bwd", "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "resolution": "3x%dx%d" % (h, w), "parallelism": "dp%d" % world, "hipgraph": run.graph is not None,
                       "grad_allreduce": run.describe_exchange()},
            **({"dist_ranks": dist.get_world_size(), "dist_backend": backend, "per_rank_ms_per_step": per_rank_ms} if exchange else {}),
            **({"exchange_exposed": exposed} if exposed is not None else {}),
            "ms_optimizer_step": round(ms_opt, 3) if ms_opt is not None else None, "loss": round(loss_val, 4),
            **({"grad_norm": grad_norm} if grad_norm is not None else {}),
            "model_tflops": round(value * gflop_img / 1e3, 2),
            # SURVEY 8(d) segment-wise roofline of the whole step (seg decoder on MFMA, everything else on HBM): 0.177 ms/img at 512x1024
            "step_roofline": {"floor_ms_per_img": round(0.177 * scale, 4), "frac": round(value / world * 0.177e-3 * scale, 4)},
            "env_overrides": env_overrides(),
        }
        if not args.no_roofline:
            try:
                # top-level fields = the dominant kernel by FLOPs (the contract's single entry); the same object again as
                # `dominant_by_flops`, and the family that takes the most time in the step as `dominant_by_time`
                res["roofline"] = dominant_launch_roofline(net, args.batch, h, w)
                res["roofline"]["dominant_by_flops"] = {k: v for k, v in res["roofline"].items()}
                try:
                    res["roofline"]["dominant_by_time"] = time_dominant_gemm_family(net, args.batch, h, w)
                except Exception as e:  # noqa: BLE001
                    res["roofline"]["dominant_by_time"] = {"error": repr(e)[:300]}
            except Exception as e:      # noqa: BLE001
                res["roofline"] = {"error": repr(e)}
        plain = world == 1 and not (args.backbone_only or args.phase or args.ddp_world1 or args.no_graph or args.no_extras)
        if plain:
            try:
                res["segments"] = segment_table(net, cfgs, batch, args.batch, h, w, median(per))
            except Exception as e:  # noqa: BLE001
                res["segments"] = {"error": repr(e)[:300]}
            res["extra_configs"] = extra_configs(args, dev, net, cfgs)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(yaml.safe_load(open(args.cfg)) | {"dataloader": cfgs["dataloader"]}, h, w)
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"error": repr(e)}
        if both_forms:
            res["exchange_forms"] = other_exchange_form(run, net, cfgs, batch, dev, rank, world, payload, args, dt, headline=res)
        emit(res)
    if exchange and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
