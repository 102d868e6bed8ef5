#!/bin/bash
# round 5: SE MLP kernels with all loads of a launch in one round -- parity, microbench, same-box A/B against HEAD's library (B)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "se_" > $O/tests_kernel.log 2>&1; tail -3 $O/tests_kernel.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
(cd tools && timeout 600 python3 bench_gate_apply.py > $O/bench_new.txt 2>&1); cut -c1-200 $O/bench_new.txt | tail -12
STEPS=60 REPS=3 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
BENCH_ARGS="--res 640x640" REPS=2 bash tools/ab_run.sh B 2>&1 | tee -a $O/ab.log
