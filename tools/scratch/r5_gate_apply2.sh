#!/bin/bash
# ordered dispatch lists of one training step with and without hn_se_gate_apply
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "se_excite or se_gate" > $O/tests_kernel.log 2>&1; tail -3 $O/tests_kernel.log
bash tools/step_timeline.sh; cp gpurun_out/trace/step.csv $O/step_gate_apply.csv
export HN_TUNING=ab
HN_SE_GATE_APPLY=0 bash tools/step_timeline.sh; cp gpurun_out/trace/step.csv $O/step_two_launch.csv
