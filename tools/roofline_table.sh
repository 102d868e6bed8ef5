#!/bin/bash
# Counter-backed roofline table of the training step: per kernel family of ONE step -- launches, time inside the
# captured hipGraph, HBM traffic (FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md), GB/s, fraction of 8 TB/s, MFMA busy share.
#   pass 1: rocprofv3 --kernel-trace --stats        on the captured step (timing; no counters)
#   pass 2-4: rocprofv3 --pmc <one group> --kernel-trace on eager launches of the same step (counters; separate passes, no other tracing)
# -> gpurun_out/roofline/{kernel_stats.csv, pmc_*.csv, roofline_table.md}; tools/copy_profiles.sh copies them to profiles/${ROUND}_*
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/roofline
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --steps 20 --warmup 5 > $O/bench_stats.log 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
cp "$(find $O/kt -name '*kernel_trace.csv' | head -1)" $O/kernel_trace.csv
rm -rf $O/kt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --no-graph --steps 2 --warmup 1 > $O/bench_pmc$i.log 2>&1
  cp "$(find $O/p$i -name '*counter_collection.csv' | head -1)" $O/pmc_$i.csv
  rm -rf $O/p$i
done
python3 tools/roofline_table.py $O > $O/roofline_table.md
tail -40 $O/roofline_table.md
