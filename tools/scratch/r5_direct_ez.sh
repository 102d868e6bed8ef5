#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5p; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q > $O/tests.log 2>&1; tail -3 $O/tests.log
STEPS=60 REPS=4 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
