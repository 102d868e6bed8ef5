"""Persistent stage kernel (hn_xstage_fwd) against the launch chain (XBlockFn.forward per block) on random identity blocks: every saved
tensor compared, both timed as replayed hipGraphs, in-kernel stamps printed.  GPU only (tools/; run through gpurun).
usage: python tools/xstage/xstage_check.py [case ...]   cases: s4 s3 s4_640 s3_640 small n8 (default: all)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multitask_hydranet_amd import ops as K            # noqa: E402
import multitask_hydranet_amd.ops.xstage as XS       # noqa: E402

CASES = {"small": (16, 8, 16, 128, 2), "s4": (16, 8, 16, 936, 13), "s3": (16, 16, 32, 376, 9), "s4_640": (16, 10, 10, 936, 13),
         "s3_640": (16, 20, 20, 376, 9), "n8": (8, 8, 16, 936, 13)}
EPS, MOM = 1e-5, 0.1


def make_params(nb, c, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    cs = c // 4
    ps = []
    for _ in range(nb):
        def r(*s, scale=1.0):
            return (torch.randn(*s, generator=g) * scale).to(dev)
        w1 = r(c, c, 1, 1, scale=(2.0 / c) ** 0.5)
        w2 = r(c, 8, 3, 3, scale=(2.0 / 72) ** 0.5)
        w3 = r(c, c, 1, 1, scale=(2.0 / c) ** 0.5)
        sw1, sb1 = r(cs, c, 1, 1, scale=(1.0 / c) ** 0.5), r(cs, scale=0.1)
        sw2, sb2 = r(c, cs, 1, 1, scale=(1.0 / cs) ** 0.5), r(c, scale=0.1)
        bn = lambda: [1.0 + 0.1 * r(c), 0.1 * r(c), 0.05 * r(c), 1.0 + 0.1 * torch.rand(c, generator=g).to(dev)]
        ps += [w1, *bn(), w2, *bn(), sw1, sb1, sw2, sb2, w3, *bn()]
    return ps


def clone_params(ps):
    return [p.clone() for p in ps]


def chain_forward(x, ps, nb):
    saved = []
    t = x
    for b in range(nb):
        p = ps[b * 19:(b + 1) * 19]
        t = K.XBlockFn.apply(t, *p, EPS, MOM, True, 1, None, None, None, None, None, None)
        saved.append(t.grad_fn.saved_tensors if t.grad_fn is not None else None)
    return t, saved


def cmp(name, a, b, res):
    a, b = a.float(), b.float()
    d = (a - b).abs()
    scale = max(float(b.abs().max()), 1e-30)
    nz = float((d > 0).float().mean())
    r = {"max_abs": float(d.max()), "rel_to_max": float(d.max()) / scale, "frac_differ": nz, "nan": bool(torch.isnan(a).any())}
    res[name] = r
    return r


def graph_time(fn, reps=30):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run_case(name, mode):
    n, h, w, c, nb = CASES[name]
    dev = torch.device("cuda:0")
    print(f"== case {name}: N={n} {h}x{w} C={c} blocks={nb} mode={mode} supported={K.lib().query('hn_xstage_supported', n, h, w, c, c // 4)}", flush=True)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(n, h, w, c, generator=g).to(dev).to(torch.bfloat16).relu_()
    ps_a = make_params(nb, c, dev, 11)
    ps_b = clone_params(ps_a)
    xa = x.clone().requires_grad_(True)
    out_a, saved = chain_forward(xa, ps_a, nb)
    torch.cuda.synchronize()
    print("   chain forward done", flush=True)
    stamps = torch.zeros((nb, 16), device=dev, dtype=torch.int64)
    try:
        with torch.no_grad():
            r = XS.xstage_forward_raw(x, ps_b, EPS, MOM, stamps=stamps, mode=mode)
    except RuntimeError as e:
        print("   FAILED:", e)
        ws = XS.xstage_ws(dev)[0]
        off = K.lib().query("hn_xstage_ws_bytes") - 256 * 16
        d = ws[off:off + 4096].view(torch.int32).view(256, 4).cpu().numpy()
        import collections
        print("   xcc census (blockIdx%8 -> set of xcc):", {k: sorted({int(d[i, 0]) >> 16 for i in range(256) if i % 8 == k}) for k in range(8)})
        print("   workgroups per xcc:", collections.Counter(int(v) >> 16 for v in d[:, 0]))
        bad = [(i, hex(int(d[i, 0])), hex(int(d[i, 1])), int(d[i, 2]), int(d[i, 3])) for i in range(256) if d[i, 1]]
        print("   failing workgroups (blockIdx, xcc<<16|ticket, code, block, aux):", bad[:40], "count", len(bad), flush=True)
        raise
    torch.cuda.synchronize()
    st = XS.xstage_status(dev)
    print(f"   persistent forward done, status=0x{st:x}", flush=True)
    res = {}
    worst = 0.0
    for b in range(nb):
        sx, z1, a, z2, z3, out, c1, c2, c3, pooled, hid, gate, sw1, sw2, bg = saved[b][:15]
        for nm, ta, tb in (("z1", r["z1"][b], z1), ("a", r["a"][b], a), ("z2", r["z2"][b], z2), ("bg", r["bg"][b], bg), ("z3", r["z3"][b], z3),
                           ("out", r["out"][b], out), ("coef1", r["coef"][b, 0], c1), ("coef2", r["coef"][b, 1], c2), ("coef3", r["coef"][b, 2], c3),
                           ("pooled", r["pooled"][b], pooled), ("hid", r["hid"][b], hid), ("gate", r["gate"][b], gate)):
            q = cmp(f"b{b}.{nm}", ta, tb, res)
            worst = max(worst, q["rel_to_max"])
        if b in (0, 1, nb - 1):
            print("   block", b, {k.split(".")[1]: (round(v["rel_to_max"], 5), round(v["frac_differ"], 5)) for k, v in res.items() if k.startswith(f"b{b}.")},
                  flush=True)
    for i in range(nb * 19):
        if i % 19 in (3, 4, 8, 9, 17, 18):      # running statistics
            q = cmp(f"run{i}", ps_b[i], ps_a[i], res)
            worst = max(worst, q["rel_to_max"])
    print(f"   worst rel-to-max over all tensors: {worst:.3e}   any nan: {any(v['nan'] for v in res.values())}", flush=True)
    ticks = stamps.cpu().numpy().astype("float64")
    if ticks[0, 0] > 0:
        names = ["start", "await_out", "gemm1", "bn1", "gconv", "bn2", "se+bg", "await_bg", "gemm3", "bn3", "out"]
        d = (ticks[:, 1:11] - ticks[:, 0:10]) / 100.0          # us (100 MHz counter)
        mid = d[1:].mean(axis=0) if nb > 1 else d[0]
        print("   stamps, mean us per phase over blocks 1..: " + "  ".join(f"{names[i + 1]}={mid[i]:.2f}" for i in range(10)), flush=True)
        print(f"   per block: {(ticks[1:, 10] - ticks[1:, 0]).mean() / 100.0 if nb > 1 else 0:.2f} us; whole launch (stamped wg): "
              f"{(ticks[nb - 1, 10] - ticks[0, 0]) / 100.0:.1f} us", flush=True)
    # timing: both as replayed graphs
    ps_c = clone_params(ps_a)

    def f_chain():
        with torch.no_grad():
            t = x
            for b in range(nb):
                t = K.XBlockFn.apply(t, *ps_c[b * 19:(b + 1) * 19], EPS, MOM, True, 1, None, None, None, None, None, None)
        return t

    def f_pers():
        with torch.no_grad():
            return XS.xstage_forward_raw(x, ps_b, EPS, MOM, mode=mode)["out"]
    t_chain = graph_time(f_chain)
    t_pers = graph_time(f_pers)
    st = XS.xstage_status(dev)
    print(f"   time per forward of the {nb} blocks: chain {t_chain:.1f} us, persistent {t_pers:.1f} us, ratio {t_pers / t_chain:.3f}, status 0x{st:x}", flush=True)
    return {"case": name, "mode": mode, "worst_rel": worst, "chain_us": t_chain, "persistent_us": t_pers, "status": st, "detail": res}


def run_bwd_case(name, mode):
    """backward: the persistent launch against XBlockFn.backward on the SAME forward tensors (the persistent forward's), block by block with
    the chain's own dout (teacher forced: one-block launches), then the whole run; both timed as replayed graphs (weight gradients deferred
    and not flushed in either)."""
    from types import SimpleNamespace
    n, h, w, c, nb = CASES[name]
    cs = c // 4
    dev = torch.device("cuda:0")
    print(f"== backward case {name}: N={n} {h}x{w} C={c} blocks={nb} mode={mode}", flush=True)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(n, h, w, c, generator=g).to(dev).to(torch.bfloat16).relu_()
    ps = make_params(nb, c, dev, 11)
    with torch.no_grad():
        r = XS.xstage_forward_raw(x, ps, EPS, MOM, mode=mode)
    dout = (torch.randn(n, h, w, c, generator=g) * 0.01).to(dev).to(torch.bfloat16)
    sws = [(ps[b * 19 + 10], ps[b * 19 + 12]) for b in range(nb)]
    grid = (n, h, w)

    def chain_block(b, d, group):
        p = ps[b * 19:(b + 1) * 19]
        fake = SimpleNamespace(saved_tensors=(x if b == 0 else r["out"][b - 1], r["z1"][b], r["a"][b], r["z2"][b], r["z3"][b], r["out"][b],
                                              r["coef"][b, 0], r["coef"][b, 1], r["coef"][b, 2], r["pooled"][b], r["hid"][b], r["gate"][b],
                                              p[10], p[12], r["bg"][b], None, None, None, None),
                               training=True, stride=1, packs=r["packs"][b], group=group,
                               wrefs=(p[0], p[14], None, p[5], p[10], p[11], p[12], p[13]), needs_input_grad=(True,) * 30)
        return K.XBlockFn.backward(fake, d)
    res = {}
    worst = 0.0
    d = dout
    douts = {}
    with torch.no_grad():
        for b in reversed(range(nb)):
            douts[b] = d
            ret = chain_block(b, d, None)
            sl = lambda t: t[b:b + 1]
            rb = XS.xstage_backward_raw(d.contiguous(), dict(z1=sl(r["z1"]), z2=sl(r["z2"]), z3=sl(r["z3"]), out=sl(r["out"]), coef=sl(r["coef"]),
                                                             hid=sl(r["hid"]), gate=sl(r["gate"])), [r["packs"][b]], [sws[b]], mode=mode)
            xb = x if b == 0 else r["out"][b - 1]
            dw1 = K.k_gemm_tn(xb, None, 0, grid, rb["dz1"][0], c, K.kp32(c), 1, c)
            dw3 = K.k_gemm_tn(r["bg"][b], None, 0, grid, rb["dz3"][0], c, K.kp32(c), 1, c)
            dw2 = K.k_gemm_tn(r["a"][b], None, 5, grid, rb["dz2"][0], c, 64, 9, 8, kh=3)
            dsw2 = rb["dpre2"][0].t() @ r["hid"][b]
            dsw1 = rb["dpre1"][0].t() @ r["pooled"][b]
            pairs = (("dx", rb["dx"], ret[0]), ("dw1", dw1, ret[1]), ("dg1", rb["dgb"][0, 0, 0], ret[2]), ("db1", rb["dgb"][0, 0, 1], ret[3]),
                     ("dw2", dw2, ret[6]), ("dg2", rb["dgb"][0, 1, 0], ret[7]), ("db2", rb["dgb"][0, 1, 1], ret[8]),
                     ("dsw1", dsw1, ret[11].view(cs, c)), ("dsb1", rb["dpre1"][0].sum(0), ret[12]),
                     ("dsw2", dsw2, ret[13].view(c, cs)), ("dsb2", rb["dpre2"][0].sum(0), ret[14]),
                     ("dw3", dw3, ret[15]), ("dg3", rb["dgb"][0, 2, 0], ret[16]), ("db3", rb["dgb"][0, 2, 1], ret[17]))
            for nm, ta, tb in pairs:
                qq = cmp(f"b{b}.{nm}", ta.reshape(tb.shape) if ta.numel() == tb.numel() else ta, tb, res)
                worst = max(worst, qq["rel_to_max"])
            if b in (nb - 1, nb - 2, 0):
                print("   block", b, {k.split(".")[1]: (round(v["rel_to_max"], 5), round(v["frac_differ"], 4)) for k, v in res.items()
                                      if k.startswith(f"b{b}.")}, flush=True)
            d = ret[0]
        st = XS.xstage_status(dev)
        print(f"   teacher-forced per block: worst rel-to-max {worst:.3e}, status 0x{st:x}, any nan {any(v['nan'] for v in res.values())}", flush=True)
        # the whole run in one launch
        stamps = torch.zeros((nb, 16), device=dev, dtype=torch.int64)
        full = XS.xstage_backward_raw(dout.contiguous(), r, r["packs"], sws, stamps=stamps, mode=mode)
        q0 = cmp("run.dx", full["dx"], d, res)
        q1 = cmp("run.dg1_first", full["dgb"][0, 0, 0], ret[2], res)
        print(f"   whole run: dx rel-to-max {q0['rel_to_max']:.3e} (differ {q0['frac_differ']:.3f}); block 0 dgamma1 {q1['rel_to_max']:.3e}", flush=True)
        ticks = stamps.cpu().numpy().astype("float64")
        if ticks[0, 0] > 0:
            names = ["top", "bn3", "dz3+await", "gemm_dbg", "se+bn2", "dz2+gconv", "bn1", "dz1+await", "gemm_dx", "tail"]
            dd = (ticks[:, 1:10] - ticks[:, 0:9]) / 100.0
            mid = dd[1:].mean(axis=0) if nb > 1 else dd[0]
            print("   stamps, mean us per phase over steps 1..: " + "  ".join(f"{names[i + 1]}={mid[i]:.2f}" for i in range(9)), flush=True)
            print(f"   per block: {(ticks[1:, 9] - ticks[1:, 0]).mean() / 100.0 if nb > 1 else 0:.2f} us; whole launch (stamped wg): "
                  f"{(ticks[nb - 1, 9] - ticks[0, 0]) / 100.0:.1f} us", flush=True)

    def f_chain():
        with torch.no_grad():
            grp = K.WgradGroup()
            dd = dout
            for b in reversed(range(nb)):
                dd = chain_block(b, dd, grp)[0]
            grp.jobs.clear(); grp.gconv.clear(); grp.tail.clear()
        return dd

    def f_pers():
        with torch.no_grad():
            return XS.xstage_backward_raw(dout, r, r["packs"], sws, mode=mode)["dx"]
    t_chain = graph_time(f_chain)
    t_pers = graph_time(f_pers)
    st = XS.xstage_status(dev)
    print(f"   time per backward of the {nb} blocks (weight gradients deferred): chain {t_chain:.1f} us, persistent {t_pers:.1f} us, ratio "
          f"{t_pers / t_chain:.3f}, status 0x{st:x}", flush=True)
    return {"case": name, "mode": mode, "dir": "bwd", "worst_rel": worst, "chain_us": t_chain, "persistent_us": t_pers, "status": st, "detail": res}


if __name__ == "__main__":
    mode = int(os.environ.get("XS_MODE", "0"))
    cases = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    dirs = [a for a in sys.argv[1:] if a in ("fwd", "bwd")] or ["fwd", "bwd"]
    out = []
    for cname in cases:
        if "fwd" in dirs:
            out.append(run_case(cname, mode))
        if "bwd" in dirs:
            out.append(run_bwd_case(cname, mode))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"xstage_check_mode{mode}.json"), "w"))
