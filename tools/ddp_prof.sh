#!/bin/bash
set -eu
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/ddpprof
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ddpprof/a -o ddp -- python3 $R/bench.py --ddp-world1 --no-extras --no-roofline --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/ddpprof/ddp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ddpprof/b -o plain -- python3 $R/bench.py --no-extras --no-roofline --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/ddpprof/plain.log 2>&1
tail -2 $R/gpurun_out/ddpprof/ddp.log | cut -c1-300
find $R/gpurun_out/ddpprof -name "*kernel_stats.csv" | head
for f in $(find $R/gpurun_out/ddpprof -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-150; done
find $R/gpurun_out/ddpprof -name "*kernel_trace.csv" -size +30M -delete
