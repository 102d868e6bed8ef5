#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5u; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for l in g4 outdgrad; do LAYER=$l timeout 300 python3 tools/stamp_seg.py 2>&1 | tail -3 | cut -c1-200; done
STEPS=60 REPS=4 bash tools/ab_run.sh B 2>&1 | tee $O/ab.log
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=2 bash tools/ab_run.sh B 2>&1 | tee -a $O/ab.log
