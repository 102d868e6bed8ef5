#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5s; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -k "seg or dgrad or end_to_end or segment or head_out" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step.csv
grep "conv3x3_direct_kernel<32" $O/step.csv | cut -c1-120
STEPS=60 REPS=3 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=2 bash tools/ab_run.sh B 2>&1 | tee -a $O/ab.log
