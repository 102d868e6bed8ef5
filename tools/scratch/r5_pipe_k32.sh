#!/bin/bash
# round 5: pipelined (32-channel chunk) direct conv for the output layer's data gradient -- parity, timeline of the launch, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5q; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -k "seg or dgrad or end_to_end or segment" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step.csv
grep "conv3x3_direct_kernel<64; false; true>\|conv3x3_direct_kernel<32; true" $O/step.csv | cut -c1-120
STEPS=60 REPS=3 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
