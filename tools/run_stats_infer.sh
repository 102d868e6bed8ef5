#!/bin/bash
# kernel-time table of the inference bench (BASELINE config 5): rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stats_infer
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --infer --res 1152x1920 --batch 32 --steps 8 --warmup 2 > $O/bench.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
rm -rf $O/kt
python3 - <<'PY'
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/stats_infer/"
rows=list(csv.DictReader(open(O+"kernel_stats.csv")))
n=[int(r["Calls"]) for r in rows if "argmax" in r["Name"]]
n=n[0] if n else 1
tot=sum(float(r["TotalDurationNs"]) for r in rows)
out=[f"executions {n}  kernel time per forward {tot/n/1e6:.2f} ms  launches per forward {sum(int(r['Calls']) for r in rows)/n:.0f}"]
for r in rows[:30]:
    out.append(f"{float(r['TotalDurationNs'])/n/1e3:8.0f} us/fwd  x{int(r['Calls'])/n:6.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:100]}")
open(O+"summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
