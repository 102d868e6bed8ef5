#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "se_" > $O/tests_kernel.log 2>&1; tail -1 $O/tests_kernel.log
STEPS=60 REPS=4 bash tools/ab_run.sh B 2>&1 | tee $O/ab2.log
