#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5t; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_post_gpu.py -q -k "seg or argmax or end_to_end or segment or infer or deploy" > $O/tests.log 2>&1; tail -3 $O/tests.log
LAYER=out timeout 600 python3 tools/stamp_seg.py 2>&1 | tail -5 | cut -c1-260
STEPS=60 REPS=3 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=3 bash tools/ab_run.sh B 2>&1 | tee -a $O/ab.log
