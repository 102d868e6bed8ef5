#!/bin/bash
export HN_TUNING=${HN_TUNING:-ab}    # product library; the package reads HN_LIB_AB / policy switches only under HN_TUNING=1|ab (_lib.policy)
# per-dispatch durations of the seg-decoder kernels in one captured step (last replay), with and without the phase form
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/segtrace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for v in 1 0; do
  HN_SEG_PHASE_UP=$v rocprofv3 --kernel-trace --output-format csv -d $O/kt$v -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench$v.log 2>&1
  f=$(find $O/kt$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the once-per-step kernel marks the last replay
idx = [i for i, r in enumerate(rows) if "seg_ce_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
sel = rows[a:b]
keys = ("conv3x3_direct", "wgrad3x3_patch", "seg_fold", "space_to_depth", "gemm_tn_kernel<128, 32", "pack_w_kernel", "depth_to_space")
tot = 0.0
print("=== HN_SEG_PHASE_UP=%s: seg-decoder dispatches of one step" % sys.argv[2])
for r in sel:
    n = r["Kernel_Name"]
    if any(k in n for k in keys):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        if "conv3x3_direct_kernel<64" in n and g < 3000 and "true" not in n:
            continue      # grouped backbone convs
        tot += d
        print("%8.1f us  wgs %7d  %s" % (d, g, n[:70]))
print("total %.1f us; step kernel time %.1f us" % (tot, sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in sel) / 1e3))
PY
  rm -rf $O/kt$v
done
