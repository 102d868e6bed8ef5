#!/bin/bash
# round 5: software-pipelined direct conv with a ring of three weight tiles (74 KB: two workgroups per CU) vs four (82 KB: one)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5r; mkdir -p $O
for v in tring3 tring4; do
  echo "== $v" | tee -a $O/bench_seg.txt
  HN_TUNING=1 HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_$v.so timeout 900 python3 tools/bench_seg.py 2>&1 | grep -v "^$" | tail -12 | tee -a $O/bench_seg.txt
done
