#!/bin/bash
# ordered dispatch list of ONE captured training step (start offset, duration, grid, kernel) -> gpurun_out/trace/step.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --steps ${STEPS:-3} --warmup 1 "$@" > $O/bench.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/step.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything after the second-to-last weighted_sum / stem dispatch
stems = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("stem_fwd_kernel")]
a = stems[-1]
# walk back to the batched weight pack that opens the step
while a > 0 and "pack_" not in rows[a - 1]["Kernel_Name"]:
    a -= 1
while a > 0 and "pack_" in rows[a - 1]["Kernel_Name"]:
    a -= 1
t0 = int(rows[a]["Start_Timestamp"])
print("idx,start_us,dur_us,grid_x,grid_y,wg,kernel")
for i, r in enumerate(rows[a:]):
    print("%d,%.1f,%.1f,%s,%s,%s,%s" % (i, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                      r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["Kernel_Name"].replace(",", ";")[:100]))
PY
rm -rf $O/kt
tail -1 $O/bench.log
wc -l $O/step.csv
