#!/bin/bash
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stemprof -o s -- python3 $R/bench.py --no-extras --no-roofline --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
grep -h "stem_fwd\|conv3x3_direct_kernel<32\|conv3x3_direct_kernel<64, false, false>\|conv3x3_direct_kernel<16" $(find $R/gpurun_out/stemprof -name "*kernel_stats.csv") | cut -c1-160
rm -rf $R/gpurun_out/stemprof
