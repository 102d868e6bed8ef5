#!/bin/bash
# round-6 profile set on the final build, one box: the main set (tools/run_profiles.sh: bench line, kernel stats, dominant-launch PMC,
# roofline table), the MFMA-kernel stall counters, per-kernel stats of the 640 x 640 step and of the inference configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ROUND=r06
bash tools/run_profiles.sh > gpurun_out/r6_profiles.log 2>&1
bash tools/pmc_stall.sh > gpurun_out/r6_pmc_stall.log 2>&1
bash tools/run_stats.sh --res 640x640 > gpurun_out/r6_stats640.log 2>&1; cp gpurun_out/stats/kernel_stats.csv gpurun_out/prof/kernel_stats_640.csv
bash tools/run_stats.sh --infer --batch 32 --res 1152x1920 > gpurun_out/r6_stats_infer.log 2>&1; cp gpurun_out/stats/kernel_stats.csv gpurun_out/prof/kernel_stats_infer.csv
tail -3 gpurun_out/prof/bench_n1.json | cut -c1-600; tail -5 gpurun_out/r6_pmc_stall.log; tail -n 2 gpurun_out/r6_stats640.log; tail -n 2 gpurun_out/r6_stats_infer.log
