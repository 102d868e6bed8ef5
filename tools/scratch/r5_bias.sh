#!/bin/bash
# round 5: GEMM bias requested before the K loop -- parity, same-box A/B against HEAD's library (B)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5o; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q > $O/tests.log 2>&1; tail -3 $O/tests.log
STEPS=60 REPS=3 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=3 bash tools/ab_run.sh B 2>&1 | tee -a $O/ab.log
