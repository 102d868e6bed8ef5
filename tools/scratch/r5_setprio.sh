#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i
rm -rf $O; mkdir -p $O
cd $R
REPS=2 STEPS=40 bash tools/ab_run.sh CP WP CWP > $O/ab_train.log 2>&1
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=2 STEPS=20 bash tools/ab_run.sh CP > $O/ab_infer.log 2>&1
cat $O/ab_train.log $O/ab_infer.log
