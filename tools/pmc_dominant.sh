#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant launch (two separate --pmc passes), printed per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcq
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 bench.py --dominant-only --steps 10 > $O/$c.log 2>&1
  f=$(find $O/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "conv3x3_direct" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[2]]
v=[float(r["Counter_Value"]) for r in rows][4:]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows][4:]
print(sys.argv[2], "KB/launch %.0f" % (sum(v)/len(v)), "n", len(v), "us %.1f" % (sum(d)/len(d)))
PY
  rm -rf $O/$c
done
