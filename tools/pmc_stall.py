"""per-kernel-family sums of the SQ wait / LDS counter passes of tools/pmc_stall.sh (one eager step)"""
import csv, sys, re, collections
O = sys.argv[1]
def fam(n):
    n = n.replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:60]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for i in (1, 2):
    rs = list(csv.DictReader(open("%s/pmc_%d.csv" % (O, i))))
    ids = sorted({int(r["Dispatch_Id"]) for r in rs if r["Kernel_Name"].startswith("pack_w_batched_kernel")})
    lo, hi = ids[-2], ids[-1]
    for r in rs:
        if lo <= int(r["Dispatch_Id"]) < hi:
            k = fam(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
keys = ["conv3x3_direct", "wgrad3x3_patch", "gemm_tn_group", "gconv_wgrad_group", "gemm_nt_kernel<64, 64, 2, 2, false, 2, false, 1>", "gemm_nt_kernel<64, 64, 2, 2, false, 2, false, 2>"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not any(s in k for s in keys):
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print("%-62s n=%3d wave_cycles %.3g  wait_any %.2f  wait_inst %.2f (lds %.2f)  active %.2f | mfma_busy/busy %.2f | lds conflict/active %.2f  insts: lds %.3g valu %.3g mfma %.3g vmem %.3g" % (
        k, len(cnt[k]), wc, v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_LDS", 0) / wc,
        v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1),
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VALU", 0),
        v.get("SQ_INSTS_MFMA", 0), v.get("SQ_INSTS_VMEM_RD", 0)))
