#!/bin/bash
# L2 hit rate per kernel family of one eager training step (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum): where do the operand
# re-reads of the small GEMMs / weight-gradient GEMMs come from?
set -eu
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/l2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --no-graph --steps 2 --warmup 1 > $O/bench.log 2>&1
f=$(find $O/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/summary.txt
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:64]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_HIT_sum":
        n[k] += 1
rows = sorted(d.items(), key=lambda kv: -(kv[1]["TCC_HIT_sum"] + kv[1]["TCC_MISS_sum"]))
for k, v in rows[:30]:
    h, m = v["TCC_HIT_sum"], v["TCC_MISS_sum"]
    print("%-66s n=%4d  L2 requests %.3g  hit rate %.3f  (64-B requests per launch: %.3g)" % (k, n[k], h + m, h / max(h + m, 1), (h + m) / max(n[k], 1)))
PY
rm -rf $O/p
