#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the grouped weight-gradient launch (separate --pmc passes) -> gpurun_out/pmc_wg.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_wg
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" ; do
  tag=$(echo $c | tr ' ' '_')
  HN_ONCE=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$tag -- python3 tools/bench_wgrad_group.py ${STAGE:-stage4} > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_tn_group" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows:
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(sys.argv[2], k, "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
done 2>&1 | tee $R/gpurun_out/pmc_wg.txt
rm -rf $O
