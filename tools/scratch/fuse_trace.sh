#!/bin/bash
export HN_TUNING=${HN_TUNING:-ab}    # policy switches are read from the environment only under HN_TUNING=1|ab (_lib.policy)
# per-kernel averages of the BiFPN fusion kernels in the training step, for the working tree and (HN_LIB_AB) another build
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in A ${1:-}; do
  if [ "$v" != A ]; then export HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_$v.so; fi
  rm -rf $R/gpurun_out/fuseprof
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fuseprof -o s -- python3 $R/bench.py --no-extras --no-roofline --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
  python3 - $v $(find $R/gpurun_out/fuseprof -name "*kernel_trace.csv") <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    nm = r['Kernel_Name'].split('(')[0]
    if 'fuse_bwd' in nm:
        d[(nm, r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in sorted(d.items()):
    v = sorted(v)
    print(sys.argv[1], k, len(v), 'median %.1f' % v[len(v) // 2], 'top quartile %.1f' % v[3 * len(v) // 4])
PY
done
rm -rf $R/gpurun_out/fuseprof
