cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seprobe -- python3 tools/se_probe.py > gpurun_out/seprobe.log 2>&1; f=$(find gpurun_out/seprobe -name "*kernel_trace.csv" | head -1); python3 - "$f" <<PY
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    if "se_" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    v=sorted(v); print(k, len(v), "median %.1f us min %.1f" % (v[len(v)//2], v[0]))
PY
rm -rf gpurun_out/seprobe
