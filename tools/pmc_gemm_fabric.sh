#!/bin/bash
# What crosses the XCD fabric in the latency-bound 1x1 GEMMs of stages 3-4?  L2 hits / misses and the L2 <-> fabric (EA) requests of the
# GEMM launches of tools/gemm_chain.py, one rocprofv3 --pmc pass per counter group (no other tracing)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fabric
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA[0-9A-Z_]*\(RDREQ\|WRREQ\)[0-9A-Z_a-z]*\|TCC_HIT[_a-z]*\|TCC_MISS[_a-z]*\|TCC_REQ[_a-z]*" | sort -u > $O/avail.txt
for st in 4 3; do
  i=0
  for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p -- python3 tools/gemm_chain.py $st 10 > $O/run_${st}_$i.log 2>&1
    f=$(find $O/p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $O/pmc_${st}_$i.csv
    rm -rf $O/p
  done
done
python3 - <<'PY' | tee $O/summary.txt
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/fabric/"
for st, (m, c) in ((4, (2048, 936)), (3, (8192, 376))):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(O + "pmc_%d_*.csv" % st)):
        for r in csv.DictReader(open(f)):
            if "gemm_nt_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in agg.items()}          # skip the first two launches (cold input)
    alg = 2.0 * m * (c + c) + 2.0 * c * c
    print("stage %d GEMM %d x %d x %d (statistics epilogue), per launch, mean of %d launches; algorithmic operand bytes %.2f MB" % (st, m, c, c, len(next(iter(agg.values()), [])) - 2, alg / 1e6))
    for k in sorted(mean):
        print("   %-26s %14.0f" % (k, mean[k]))
    h, ms = mean.get("TCC_HIT_sum", 0), mean.get("TCC_MISS_sum", 0)
    if h + ms:
        print("   L2 requests %.3g, hit rate %.3f" % (h + ms, h / (h + ms)))
    rd, rd32 = mean.get("TCC_EA0_RDREQ_sum"), mean.get("TCC_EA0_RDREQ_32B_sum")
    if rd is not None and rd32 is not None:
        print("   L2 -> fabric reads: %.2f MB (32-B requests %.0f, 64-B requests %.0f)" % ((rd32 * 32 + (rd - rd32) * 64) / 1e6, rd32, rd - rd32))
    wr, wr64 = mean.get("TCC_EA0_WRREQ_sum"), mean.get("TCC_EA0_WRREQ_64B_sum")
    if wr is not None and wr64 is not None:
        print("   L2 -> fabric writes: %.2f MB" % ((wr64 * 64 + (wr - wr64) * 32) / 1e6))
    if "FETCH_SIZE" in mean:
        print("   FETCH_SIZE %.0f KB (x2 per the gfx950 note = %.2f MB), WRITE_SIZE %.0f KB" % (mean["FETCH_SIZE"], 2 * mean["FETCH_SIZE"] * 1024 / 1e6, mean.get("WRITE_SIZE", 0)))
PY
