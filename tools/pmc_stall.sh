#!/bin/bash
# Where do the MFMA kernels wait?  One eager step under `rocprofv3 --pmc` (SQ wait / LDS counters, two passes), per-kernel-family sums.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stall
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python3 bench.py --no-cpu-baseline --no-optimizer --no-extras --no-roofline --no-graph --steps 2 --warmup 1 > $O/bench$i.log 2>&1
  cp "$(find $O/p$i -name '*counter_collection.csv' | head -1)" $O/pmc_$i.csv
  rm -rf $O/p$i
done
python3 tools/pmc_stall.py $O | tee $O/summary.txt
