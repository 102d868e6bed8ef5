#!/bin/bash
# kernel-trace durations of the grouped weight-gradient launch (gemm_tn_group + wgrad_reduce_group) per ablation setting of knob 9
set -eu
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wgtrace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for dbg in ${DBGS:-0 1 2 4 6 7}; do
  HN_TUNING=1 HN_DBG=$dbg rocprofv3 --kernel-trace --output-format csv -d $O/k$dbg -- python3 tools/bench_wgrad_group.py ${STAGES:-stage4 stage3} > $O/log$dbg.txt 2>&1
  f=$(find $O/k$dbg -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $dbg <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_tn_group" in n or "wgrad_reduce_group" in n:
        d[(n.split("(")[0][:60], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v)
    print("dbg", sys.argv[2], k, "n=%d median %.1f us" % (len(v), v[len(v) // 2]))
PY
  rm -rf $O/k$dbg
done
