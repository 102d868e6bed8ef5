#!/bin/bash
# kernel-time table of one bench run (hipGraph replays): rocprofv3 --kernel-trace --stats, summarised per kernel and per step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stats
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --steps 8 --warmup 2 "$@" > $O/bench.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
rm -rf $O/kt
python3 - <<'PY'
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/stats/"
rows=list(csv.DictReader(open(O+"kernel_stats.csv")))
# steps = calls of a once-per-step kernel
steps=[int(r["Calls"]) for r in rows if "seg_loss_finalize" in r["Name"] or "det_loss_finalize" in r["Name"]]
n=steps[0] if steps else 1
tot=sum(float(r["TotalDurationNs"]) for r in rows)
out=[f"steps {n}  kernel time per step {tot/n/1e6:.2f} ms  launches per step {sum(int(r['Calls']) for r in rows)/n:.0f}"]
for r in rows[:45]:
    out.append(f"{float(r['TotalDurationNs'])/n/1e3:8.0f} us/step  x{int(r['Calls'])/n:6.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:90]}")
open(O+"summary.txt","w").write("\n".join(out)+"\n")
PY
