#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ROUND=r05
bash tools/run_profiles.sh > gpurun_out/r5_profiles.log 2>&1
bash tools/pmc_stall.sh > gpurun_out/r5_pmc_stall.log 2>&1
tail -3 gpurun_out/prof/bench_n1.json | cut -c1-600; tail -5 gpurun_out/r5_pmc_stall.log
