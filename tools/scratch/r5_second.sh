#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5b
rm -rf $O; mkdir -p $O
cd $R
python tools/scratch/ddp_trainer_diag.py cap > $O/diag_cap.log 2>&1
python tools/scratch/ddp_trainer_diag.py eager > $O/diag_eager.log 2>&1
timeout 1500 python -m pytest tests/test_train_gpu.py -q -m gpu > $O/tests_train.log 2>&1; echo "rc $?" >> $O/tests_train.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -k "xblock" > $O/tests_xblock.log 2>&1; echo "rc $?" >> $O/tests_xblock.log
timeout 1500 python -m pytest tests/test_fullsize2_gpu.py -q -m gpu -k "deep_stage or 640 or default" > $O/tests_fs2.log 2>&1; echo "rc $?" >> $O/tests_fs2.log
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step512.csv
bash tools/step_timeline.sh --res 640x640 > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step640.csv
tail -4 $O/diag_cap.log; tail -3 $O/tests_train.log $O/tests_xblock.log $O/tests_fs2.log
