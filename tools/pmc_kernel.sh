#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (raw KB, per launch) of the kernels matching $1 in a short bench run (two separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmck
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-graph --steps 2 --warmup 1 > $O/$c.log 2>&1
  f=$(find $O/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c "$1" <<'PY'
import csv, sys, re, collections
pat=re.compile(sys.argv[3])
d=collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2] and pat.search(r["Kernel_Name"]):
        k=(r["Kernel_Name"][:60], r["Grid_Size"])
        d.setdefault(k, []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
for k,v in d.items():
    print(sys.argv[2], k[0], "grid", k[1], "n", len(v), "KB/launch %.0f" % (sum(a for a,_ in v)/len(v)), "us %.1f" % (sum(b for _,b in v)/len(v)))
PY
  rm -rf $O/$c
done
