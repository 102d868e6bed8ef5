#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5d
rm -rf $O; mkdir -p $O
cd $R
export HN_TUNING=ab
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "bifpn_node or epilogue_reductions or conv1x1 or batchnorm3 or tower or head_out" > $O/tests_node.log 2>&1; echo "rc $?" >> $O/tests_node.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize2_gpu.py -q -m gpu > $O/tests_model.log 2>&1; echo "rc $?" >> $O/tests_model.log
# A/B: GEMM K loop (A = register prefetch depth 2, DMA = LDS-DMA double buffer, R3 = depth 3), node kernel off for all
export HN_SEPNODE=0
REPS=2 STEPS=40 bash tools/ab_run.sh DMA R3 > $O/ab_gemm.log 2>&1
BENCH_ARGS="--res 640x640" REPS=1 STEPS=40 bash tools/ab_run.sh DMA > $O/ab_gemm640.log 2>&1
BENCH_ARGS="--infer --batch 32 --res 1152x1920" REPS=1 STEPS=20 bash tools/ab_run.sh DMA > $O/ab_gemm_infer.log 2>&1
# per-level durations of the node kernel
export HN_SEPNODE=1
bash tools/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/trace/step.csv $O/step512_node.csv
grep -n "passed\|failed\|^FAILED" $O/tests_node.log $O/tests_model.log | tail; cat $O/ab_gemm.log $O/ab_gemm640.log $O/ab_gemm_infer.log
