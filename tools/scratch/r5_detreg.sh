#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5j; mkdir -p $O
show() { python3 -c "import json; d=json.load(open('gpurun_out/tiny_end_to_end.json'))['loss']; print('$1', {k: (round(v[0],5), round(v[1],5), round(abs(v[0]-v[1])/abs(v[1]),4)) for k, v in d.items()})"; }
timeout 600 python -m pytest tests/test_model_gpu.py -q -k "test_end_to_end_losses_features_and_statistics and tiny" > $O/t_new.log 2>&1; tail -1 $O/t_new.log; show new
HN_TUNING=ab HN_LIB_AB=$R/multitask_hydranet_amd/libhydranet_hip_B.so timeout 600 python -m pytest tests/test_model_gpu.py -q -k "test_end_to_end_losses_features_and_statistics and tiny" > $O/t_head.log 2>&1; tail -1 $O/t_head.log; show head
HN_TUNING=ab HN_SE_GATE_APPLY=0 timeout 600 python -m pytest tests/test_model_gpu.py -q -k "test_end_to_end_losses_features_and_statistics and tiny" > $O/t_two.log 2>&1; tail -1 $O/t_two.log; show new_two_launch
