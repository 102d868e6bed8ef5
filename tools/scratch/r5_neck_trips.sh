#!/bin/bash
# round 5: fuse_fwd / maxpool_bwd_arg with every load of a pixel in one round -- parity, same-box A/B against HEAD's library (B)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5n; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q > $O/tests.log 2>&1; tail -3 $O/tests.log
STEPS=60 REPS=4 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
