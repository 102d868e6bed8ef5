#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5m; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -q -k "loss or end_to_end or trainer or train" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step.csv; wc -l $O/step.csv; grep -c "at::native\|rocclr" $O/step.csv
