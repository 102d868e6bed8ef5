#!/bin/bash
# per-dispatch durations (one captured step) of the kernels whose name matches $1 (regex), grouped by grid size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --steps ${STEPS:-4} --warmup 1 > $O/bench.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$1" > $O/summary.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
d = collections.OrderedDict()
for r in rows:
    if pat.search(r["Kernel_Name"]):
        k = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print("%-72s grid %8s x %-4s wg %-5s  n=%-3d median %8.1f us" % (k[0], k[1], k[2], k[3], len(v), v[len(v) // 2]))
PY
rm -rf $O/kt
cat $O/summary.txt
