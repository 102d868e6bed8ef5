#!/bin/bash
export HN_TUNING=${HN_TUNING:-ab}    # product library; the package reads HN_LIB_AB / policy switches only under HN_TUNING=1|ab (_lib.policy)
# per-dispatch durations of the seg-decoder conv kernels in one captured step: working-tree library, then each named variant
# (multitask_hydranet_amd/libhydranet_hip_<name>.so):   tools/seg_trace_ab.sh B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/segtrace_ab
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for v in A "$@"; do
  if [ $v = A ]; then unset HN_LIB_AB; else export HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -- python3 bench.py --no-cpu-baseline --no-extras --no-roofline --no-optimizer --steps 3 --warmup 1 > $O/bench$v.log 2>&1
  f=$(find $O/kt$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee $O/trace_$v.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "seg_ce_fwd" in r["Kernel_Name"]]
sel = rows[idx[-2]:idx[-1]]
print("=== variant %s: conv3x3 dispatches of one step" % sys.argv[2])
tot = 0.0
for r in sel:
    n = r["Kernel_Name"]
    if "conv3x3_direct_kernel<128" in n or "conv3x3_wide" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        tot += d
        print("%8.1f us  wgs %7d  vgpr %s  %s" % (d, g, r.get("VGPR_Count", "?"), n[:60]))
print("total %.1f us" % tot)
PY
  rm -rf $O/kt$v
done
