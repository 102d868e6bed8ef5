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


if __name__ == "__main__":
    mode = int(os.environ.get("XS_MODE", "0"))
    cases = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    out = []
    for cname in cases:
        out.append(run_case(cname, mode))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"xstage_check_mode{mode}.json"), "w"))
