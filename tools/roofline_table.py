"""merge the passes of tools/roofline_table.sh into a markdown table (see there)"""
import collections
import csv
import json
import re
import sys

O = sys.argv[1]
PEAK_HBM = 8000.0          # GB/s
CUS, SIMDS, XCDS = 256, 4, 8


def fam(name):
    """kernel family = kernel name with its template arguments, without the parameter list"""
    name = name.replace("void ", "")
    return re.sub(r"\(.*$", "", name)[:70]


# ---- pass 1: time inside the captured step.  The stats cover 2 eager warm-up steps + the capture run + 25 replays + the dominant-launch
# loop; per-step figures come from the kernel trace: dispatches between the last two batched weight packs = one replayed step
rows = sorted(csv.DictReader(open(O + "/kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("pack_w_batched_kernel")]
a, b = packs[-2], packs[-1]
step = rows[a:b]
span_us = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
t = collections.OrderedDict()
for r in step:
    k = fam(r["Kernel_Name"])
    d = t.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


# ---- counter passes: eager launches; one step = the dispatches between the last two weight packs
def counters(path):
    rs = list(csv.DictReader(open(path)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rs if r["Kernel_Name"].startswith("pack_w_batched_kernel")})
    lo, hi = ids[-2], ids[-1]                                   # one whole eager step: between the last two batched weight packs
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in rs:
        if not (lo <= int(r["Dispatch_Id"]) < hi):
            continue
        k = fam(r["Kernel_Name"])
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return out, {k: len(v) for k, v in n.items()}


fetch, nf = counters(O + "/pmc_1.csv")
write, nw = counters(O + "/pmc_2.csv")
sq, ns = counters(O + "/pmc_3.csv")
tot_us = sum(v[1] for v in t.values())
print("# Counter-backed roofline table of one training step (big cfg, batch 16, 3x512x1024, bf16)\n")
print("Produced by `tools/roofline_table.sh` on one MI355X.  Time: kernel-trace durations of ONE replayed step of the captured hipGraph "
      "(%d launches, %.2f ms of kernel time, %.2f ms first start to last end).  Traffic: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
      "(separate passes, eager launches of the same step); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950 tallies the 128-byte "
      "read requests at 64 B: MI355X_MICROARCH.md).  MFMA: `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16` "
      "(third pass); busy %% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / %d XCDs x %d CUs x %d SIMDs); executed bf16 TFLOP/s = MOPS x 512 / time.\n"
      % (len(step), tot_us / 1e3, span_us / 1e3, XCDS, CUS, SIMDS))
print("| kernel family | launches/step | us/step | avg us | HBM MB/step | MB/launch | GB/s | of 8 TB/s | MFMA busy % | bf16 TFLOP/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot_mb = 0.0
for k, (n, us) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    mb = (2 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) + write.get(k, {}).get("WRITE_SIZE", 0.0)) / 1024.0
    tot_mb += mb
    if us < 0.004 * tot_us:
        continue
    gbs = mb / 1024.0 / (us * 1e-6) if us > 0 else 0.0
    s = sq.get(k, {})
    gui = s.get("GRBM_GUI_ACTIVE", 0.0)
    busy = 100.0 * s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / XCDS * CUS * SIMDS) if gui else 0.0
    tfl = s.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / (us * 1e-6) / 1e12 if us > 0 else 0.0
    print("| `%s` | %d | %.0f | %.1f | %.0f | %.2f | %.0f | %.3f | %.1f | %.0f |" % (k, n, us, us / n, mb, mb / max(n, 1), gbs, gbs / PEAK_HBM, busy, tfl))
alg_gb = 1.281 * 16
print("\nWhole step: %.2f GB of HBM traffic measured vs %.1f GB algorithmic (SURVEY 8(d): 1.281 GB per image x 16) = %.2fx; "
      "%.2f TB/s average over the %.2f ms of kernel time.  Families below 0.4 %% of the step are not listed (their traffic is in the total)."
      % (tot_mb / 1024.0, alg_gb, tot_mb / 1024.0 / alg_gb, tot_mb / 1024.0 / 1024.0 / (tot_us * 1e-6), tot_us / 1e3))
json.dump({"launches": len(step), "kernel_ms": tot_us / 1e3, "span_ms": span_us / 1e3, "hbm_gb": tot_mb / 1024.0}, open(O + "/summary.json", "w"))
